"""Where the torch-free command's start-up goes before the first byte is uploaded: dlopen of libt2v_hip.so + the HIP runtime,
the first t2v_create (hipInit, device properties, the side stream, the dispatch-order self-test and its code object), a second
context, a 1.46 GB hipMalloc, 8 MiB of page-locked memory.  MI355X box, round 4: import 0.03 s, dlopen 0.06 s, first t2v_create
0.195 s (0.109 s with T2V_HIP_RUNTIME=system, the ROCm 7.2 runtime instead of the 7.0 one PyTorch bundles), the rest < 10 ms.
    gpurun -- 'python scripts/init_probe.py; T2V_HIP_RUNTIME=system python scripts/init_probe.py'"""
import time, sys, os, ctypes
t0 = time.perf_counter()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from text2video_amd import _xp
_xp.use_lean()
from text2video_amd import _lib
t1 = time.perf_counter()
lib = _lib.load()
t2 = time.perf_counter()
h = ctypes.c_void_p()
lib.t2v_create(ctypes.byref(h), 0)
t3 = time.perf_counter()
h2 = ctypes.c_void_p()
lib.t2v_create(ctypes.byref(h2), 0)
t4 = time.perf_counter()
p = ctypes.c_void_p()
lib.t2v_device_malloc(h, ctypes.c_size_t(1459108100), ctypes.byref(p))
t5 = time.perf_counter()
q = ctypes.c_void_p()
lib.t2v_host_malloc(h, ctypes.c_size_t(8 << 20), ctypes.byref(q))
t6 = time.perf_counter()
print("import %.3f  load(dlopen) %.3f  first t2v_create %.3f  second %.3f  hipMalloc 1.46GB %.3f  hipHostMalloc 8MB %.3f" % (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t6-t5))
