"""Checkpoint -> device with the memory-mapped file registered as pinned host memory (hipHostRegister) against the plain pageable
copy: each in a fresh process (cold mapping, warm page cache), full-size flow generator (1.46 GB latest_net_G0.pth)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from text2video_amd.model import load_checkpoint
    mode = sys.argv[3]
    torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
    t0 = time.perf_counter(); sd = load_checkpoint(sys.argv[2]); t1 = time.perf_counter()
    reg = 0.0
    if mode == "register":
        lo = min(v.untyped_storage().data_ptr() for v in sd.values())
        hi = max(v.untyped_storage().data_ptr() + v.untyped_storage().nbytes() for v in sd.values())
        lo_p, hi_p = lo & ~4095, (hi + 4095) & ~4095
        rt = torch.cuda.cudart()
        r = rt.cudaHostRegister(lo_p, hi_p - lo_p, 0)
        reg = time.perf_counter() - t1
        print("   register %.1f MB: status %s, %.3f s" % ((hi_p - lo_p) / 1e6, r, reg), flush=True)
    t2 = time.perf_counter()
    d = {k: v.to("cuda:0", non_blocking=(mode == "register")) for k, v in sd.items()}
    torch.cuda.synchronize(); t3 = time.perf_counter()
    chk = float(sum(v.double().sum() for v in list(d.values())[:8]))
    print("%-9s read %.3f s, register %.3f s, copies %.3f s, total to device %.3f s (checksum %.6f)" % (mode, t1 - t0, reg, t3 - t2, t3 - t1, chk), flush=True)
    os._exit(0)
sys.path.insert(0, ROOT)
import torch
from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
tmp = tempfile.mkdtemp(prefix="t2v_up_")
path = os.path.join(tmp, "latest_net_G0.pth")
torch.save(synthetic_state_dict(GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=False, norm="batch"), seed=1, flow_gain=0.1), path)
for rep in range(3):
    for mode in ("plain", "register"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", path, mode])
import shutil; shutil.rmtree(tmp, ignore_errors=True)
