import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_generator as T
for case in T.CASES:
    name, kw, scales, H, W = case
    for fg in (0.1,):
        for forced in (False, True):
            ref, hip = T._build(kw, scales, flow_gain=fg)
            poses = T._pose_seq(6, H, W)
            errs = []
            for t in range(2, 6):
                A = poses[t - 2:t + 1].unsqueeze(0)
                if forced and ref.fake_B_prev is not None:
                    hip.load_prev(ref.fake_B_prev)
                want = ref.inference(A)
                got, _ = hip.inference(A.to("cuda:0"))
                errs.append((got.cpu() - want).abs().max().item())
            print("%-22s forced=%d" % (name, forced), " ".join("%.2e" % e for e in errs), flush=True)
