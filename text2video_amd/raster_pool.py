"""Pool of rasteriser worker PROCESSES for the pose dataset.

The workers are fresh interpreters (`python -m text2video_amd.raster_pool`), not forks of the caller: the frame
loop's process holds a live HIP runtime (device mappings, pinned host buffers, RCCL in multi-GPU runs), and forking
it copies -- and at worker exit tears down -- all of those mappings per worker (measured: ~2 s per worker, 33 s for
a 16-worker pool inside bench.py).  A worker imports numpy / scipy / PIL only, reads length-prefixed pickled jobs
from stdin and writes the uint8 pose map back on stdout.  In the parent one thread per worker does the blocking
pipe I/O, so `submit()` returns a concurrent.futures.Future like an executor would.  Pools are cached per worker
count and reused by later frame loops of the same process (test_fifo.py serves many requests).
"""
import atexit
import os
import pickle
import struct
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_exact(f, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = f.read(n - len(buf))
        if not chunk:
            raise EOFError("rasteriser worker closed its pipe")
        buf += chunk
    return bytes(buf)


def _worker_main():
    """child side: job tuple in, (h, w, c) + bytes out; an exception travels back as text"""
    import numpy as np  # noqa: F401
    from text2video_amd.pose_dataset import _render_job
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr            # stray prints must not corrupt the result stream
    while True:
        head = inp.read(4)
        if len(head) < 4:
            return
        job = pickle.loads(_read_exact(inp, struct.unpack("<I", head)[0]))
        try:
            m = _render_job(job)
            out.write(struct.pack("<iii", *m.shape) + m.tobytes())
        except Exception as e:        # noqa: BLE001 -- reported to the caller, which raises
            msg = ("%s: %s" % (type(e).__name__, e)).encode()
            out.write(struct.pack("<iii", -1, len(msg), 0) + msg)
        out.flush()


class RasterPool:
    def __init__(self, workers):
        self.workers = workers
        self._local = threading.local()
        self._procs = []
        self._lock = threading.Lock()
        self._threads = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="raster")
        env = dict(os.environ, PYTHONPATH=_ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")   # 2-point fits: no BLAS threads
        self._env = env
        for _ in range(workers):       # start them all now: their imports run in parallel
            self._procs.append(self._spawn())
        self._free = list(self._procs)

    def _spawn(self):
        return subprocess.Popen([sys.executable, "-m", "text2video_amd.raster_pool"], stdin=subprocess.PIPE,
                                stdout=subprocess.PIPE, env=self._env, cwd=_ROOT)

    def _mine(self):
        p = getattr(self._local, "proc", None)
        if p is None:
            with self._lock:
                p = self._local.proc = self._free.pop()
        return p

    def _replace(self, dead):
        """the I/O thread's worker died or its pipe lost framing: start a fresh one in its place (the pool is cached per
        process -- test_fifo.py serves many requests -- so a dead worker would otherwise fail every later job of its thread)"""
        try:
            dead.kill()
        except OSError:
            pass
        fresh = self._spawn()
        with self._lock:
            self._procs = [fresh if q is dead else q for q in self._procs]
        self._local.proc = fresh
        return fresh

    def _exchange(self, p, blob):
        import numpy as np
        p.stdin.write(struct.pack("<I", len(blob)) + blob)
        p.stdin.flush()
        h, w, c = struct.unpack("<iii", _read_exact(p.stdout, 12))
        if h < 0:
            raise RuntimeError("rasteriser worker: " + _read_exact(p.stdout, w).decode(errors="replace"))
        # (a writable copy: np.frombuffer on bytes is read-only)
        return np.frombuffer(bytearray(_read_exact(p.stdout, h * w * c)), dtype=np.uint8).reshape(h, w, c)

    def _run(self, job):
        p = self._mine()
        blob = pickle.dumps(job, protocol=pickle.HIGHEST_PROTOCOL)
        try:
            return self._exchange(p, blob)
        except (EOFError, BrokenPipeError, OSError, struct.error):
            return self._exchange(self._replace(p), blob)      # one retry on a fresh worker; a second failure raises

    def submit(self, job):
        return self._threads.submit(self._run, job)

    def close(self, kill=False):
        """kill=True: the caller is about to leave (vid2vid/test.py's os._exit) and every job has been answered -- the workers
        hold nothing worth an orderly interpreter shutdown (numpy / PIL finalisation: tens of ms each, on the command's wall
        clock), so they are killed instead of being waited for."""
        self._threads.shutdown(wait=True)
        if kill:
            for p in self._procs:
                try:
                    p.kill()
                except OSError:
                    pass
        for p in self._procs:
            try:
                p.stdin.close()
            except OSError:
                pass
        for p in self._procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
        self._procs = []


_pools = {}


def get_pool(workers):
    pool = _pools.get(workers)
    if pool is None:
        pool = _pools[workers] = RasterPool(workers)
    return pool


@atexit.register
def _close_all(kill=False):
    for pool in list(_pools.values()):
        pool.close(kill)
    _pools.clear()


if __name__ == "__main__":
    _worker_main()
