"""Data gradients of the convolutions as FORWARD convolutions with the adjoint geometry (the train
step's `updateGradInput`, THCUNN.h:664,794).  No dedicated kernel: the implicit-GEMM forward kernel
computes all of them.

    forward                                   data gradient
    ----------------------------------------  ----------------------------------------------------------
    conv k, stride 1, zero pad p              conv k, stride 1, zero pad k-1-p, weight flipped+transposed
    conv k, stride 1, reflect pad p           same with pad k-1 on dY -> [H+2p, W+2p], then the reflect-pad adjoint
    conv k3/k4, stride 2, pad p               ConvTranspose(k, s2, p, output_padding) with the SAME weight tensor
                                              ([Cout,Cin,k,k] read as [Cin_T,Cout_T,k,k])
    ConvTranspose k3 s2 p1 op1                conv k3 s2 p1 with the SAME weight tensor ([Cin,Cout,3,3] read as
                                              [Cout_c,Cin_c,3,3])
"""
import os


from . import ops


class ConvDataGrad:
    """dX for one forward conv layer.  Weight packings are cached per weight version by the caller
    (call `refresh(weight)` after every optimiser step)."""

    def __init__(self, fwd_desc):
        d = self.fwd = fwd_desc
        self.fold = 0
        if d.transposed:
            # forward: [H,W] -> [2H,2W]; gradient: conv k3 s2 p1 on dY
            self.desc = ops.conv_desc(2 * d.H, 2 * d.W, d.Cout, d.Cin, 3, 2, 1, ops.PAD_ZERO)
            self.kind = "same"
            self._polyphase()
        elif d.stride == 2:
            ho, wo = ops.conv_out_dims(d)
            # output_padding so that the transposed conv reproduces the forward input size
            op_h = d.H - ((ho - 1) * 2 - 2 * d.pad + d.kH)
            op_w = d.W - ((wo - 1) * 2 - 2 * d.pad + d.kW)
            assert op_h == op_w and op_h in (0, 1), "unsupported stride-2 geometry"
            self.desc = ops.conv_desc(ho, wo, d.Cout, d.Cin, d.kH, 2, d.pad, ops.PAD_ZERO, True, output_padding=op_h)
            self.kind = "same"
            self._polyphase()
        else:
            ho, wo = ops.conv_out_dims(d)
            if d.pad_mode == ops.PAD_REFLECT and d.pad > 0:
                self.fold = d.pad
                pad = d.kH - 1
            else:
                pad = d.kH - 1 - d.pad
            self.desc = ops.conv_desc(ho, wo, d.Cout, d.Cin, d.kH, 1, pad, ops.PAD_ZERO)
            self.kind = "flip"
            # the gradient of a 3x3 stride-1 conv is again one: Winograd where that is the smaller GEMM
            cap = int(os.environ.get("T2V_CONV_ALGO", "0"))
            if d.kH == 3 and d.Cin % 4 == 0 and cap != 1:
                algo = ops.best_conv_algo(self.desc, ops.round_up(self.desc.Cin, 4), cap)
                if algo != ops.ALGO_DIRECT:
                    self.desc = ops.with_algo(self.desc, algo)
        self.packed = None

    def _polyphase(self):
        """the gradient of a deep stride-2 3x3 layer is a transposed one (and vice versa) with the SAME weight tensor: both
        run as polyphase Winograd F(4,2) where the library selects that form (csrc/polyphase.hip)"""
        if int(os.environ.get("T2V_CONV_ALGO", "0")) == 0 and self.desc.kH == 3 and \
                ops.polyphase_pays(self.desc, ops.round_up(self.desc.Cin, 4)):
            self.desc = ops.with_algo(self.desc, ops.ALGO_POLYPHASE)

    def refresh(self, weight):
        """weight: the forward layer's torch-layout weight on the device."""
        # kind "flip": the gradient's filter is the forward one flipped and transposed -- gathered that way by the packer
        self.packed = ops.pack_conv_weight(weight.contiguous(), self.desc, ops.round_up(self.desc.Cin, 4),
                                           adjoint=self.kind == "flip")
        return self

    def batch(self, dy, out):
        """dy: [B, Hout, Wout, cs] -> out [B, H, W, round_up4(Cin)]: one launch for a direct-algorithm gradient conv"""
        if not self.fold:
            return ops.conv2d_auto_batch(dy, self.packed, None, self.desc, out=out)
        dxp = ops.conv2d_auto_batch(dy, self.packed, None, self.desc)
        for i in range(dy.shape[0]):
            ops.reflect_pad_backward(dxp[i], self.fold, out=out[i])
        return out

    def __call__(self, dy, out=None):
        """dy: [Hout, Wout, cs>=Cout] -> dX [H, W, round_up4(Cin)] (written into `out` if given)."""
        if not self.fold:
            return ops.conv2d_auto(dy, self.packed, None, self.desc, out=out)
        dxp = ops.conv2d_auto(dy, self.packed, None, self.desc)
        return ops.reflect_pad_backward(dxp, self.fold, out=out)
