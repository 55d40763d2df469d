#!/bin/bash
# A/B of T2V_XCD_SLICES (overlap-reading transform kernels give every XCD its own channel slices) on the ResnetBlock conv's data
# gradient (1024 -> 1024 at 64x64): per-kernel averages under rocprofv3 --kernel-trace --stats, both settings twice, alternating,
# and the bits of the result compared.  Output: gpurun_out/ab_xcd/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_xcd
mkdir -p $out
for rep in 1 2; do
  for v in 0 1; do
    T2V_XCD_SLICES=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out/p${v}_$rep -o t -- \
        python scripts/dgrad_fw_bench.py 100 > $out/log_${v}_$rep.txt 2>&1
    f=$(find $out/p${v}_$rep -name "*kernel_stats.csv" | head -1)
    echo "== T2V_XCD_SLICES=$v run $rep" | tee -a $out/summary.txt
    grep -E "dgrad_output|reflect_pad_backward|winograd4_input|winograd4_dy|wino_gemm_sk" "$f" | cut -d, -f1-5 | cut -c1-160 | tee -a $out/summary.txt
  done
done
python - <<'PY' 2>&1 | tee -a gpurun_out/ab_xcd/summary.txt
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, ".")
from text2video_amd import ops
H = W = 64; C = 1024
desc = ops.with_algo(ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT), ops.ALGO_WINOGRAD_F4)
g = torch.Generator().manual_seed(1)
xs = torch.randn(1, H, W, C, generator=g).cuda(); dys = torch.randn(1, H, W, C, generator=g).cuda()
wd = (torch.randn(C, C, 3, 3, generator=g) * 0.02).cuda()
ws = ops.backward_weight_winograd_workspace(desc, C, 1, "cuda:0")
ops.conv2d_backward_weight_winograd_stages(xs, dys, desc, ws, 1, 0, False)
u = ops.pack_conv_weight(wd, desc, C)
dx = ops.conv2d_backward_data_winograd(desc, 1, 0, ws, C, u, forward_weights=True)
torch.save(dx.cpu(), sys.argv[1])
'''
for v in ("0", "1"):
    subprocess.run([sys.executable, "-c", code, "/tmp/dx_%s.pt" % v], check=True, env=dict(os.environ, T2V_XCD_SLICES=v))
import torch
a, b = torch.load("/tmp/dx_0.pt"), torch.load("/tmp/dx_1.pt")
print("data gradient bit-equal across T2V_XCD_SLICES:", torch.equal(a, b), float((a - b).abs().max()))
PY
