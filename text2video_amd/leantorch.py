"""The handful of torch names the frame loop uses, without torch.

`vid2vid/test.py` is started once per utterance (/root/reference/text2video_audio.sh:37-44) and `import torch` was the
largest term of its start-up.  On the inference path torch only ever was an allocator, a stream and an event
(ops.py's docstring), so this module provides exactly those names -- empty / zeros / from_numpy / cat / load, Tensor with
data_ptr / shape / to / copy_ / clone / numpy, cuda.current_stream / Event / synchronize -- over the library's own
host-plumbing entry points (include/t2v.h ABI 14: t2v_device_malloc, t2v_memcpy, t2v_stream_create, ...).  `_xp.use_lean()`
selects it; ops.py / generator.py / model.py run unchanged on either provider and produce the same frames
(tests/test_gpu_e2e.py::test_lean_command_writes_the_same_files).

Not a tensor library: no arithmetic, no views, no autograd.  Anything the frame loop does not use raises AttributeError.

Device buffers come from a size-keyed free list (the frame loop allocates the same few shapes every frame; a buffer
returns to the list when its Tensor is collected and is handed out again in stream order -- the loop runs on ONE stream,
the generator's second stream is forked from and joined into it inside t2v_generator_forward).

load(): a reader for torch.save'd state dicts that needs no torch -- the zip container of torch >= 1.6
($SP/torch/serialization.py of any modern torch: `archive/data.pkl` + one stored record per storage) and the legacy
stream torch 0.4.1 wrote ($SP/torch/serialization.py:286-300 _save: magic, protocol, sys_info, the pickled object with
persistent storage ids, the storage keys, then numel + raw bytes per storage).  Tensors come back as numpy views of ONE
read-only memory map of the file, so the H2D copies read from the page cache.
"""
import collections
import ctypes
import io
import pickle
import struct
import threading
import zipfile

import numpy as np

from . import _lib
from ._lib import check


# ------------------------------------------------------------------------------------------------------------------
# dtypes / devices
# ------------------------------------------------------------------------------------------------------------------
class dtype:
    def __init__(self, name, np_dtype):
        self.name, self.np = name, np.dtype(np_dtype)
        self.itemsize = self.np.itemsize

    def __repr__(self):
        return "leantorch." + self.name


float32 = dtype("float32", np.float32)
float64 = dtype("float64", np.float64)
float16 = dtype("float16", np.float16)
uint8 = dtype("uint8", np.uint8)
int32 = dtype("int32", np.int32)
int64 = dtype("int64", np.int64)
_BY_NP = {d.np: d for d in (float32, float64, float16, uint8, int32, int64)}


class device:
    def __init__(self, spec="cuda", index=None):
        if isinstance(spec, device):
            self.type, self.index = spec.type, spec.index
            return
        if isinstance(spec, int):
            self.type, self.index = "cuda", spec
            return
        t, _, i = str(spec).partition(":")
        if t not in ("cuda", "cpu"):
            raise ValueError("leantorch.device: %r" % (spec,))
        self.type = t
        self.index = int(i) if i else (index if index is not None else (cuda.current_device() if t == "cuda" else None))

    def __eq__(self, other):
        other = device(other) if not isinstance(other, device) else other
        return (self.type, self.index) == (other.type, other.index)

    def __hash__(self):
        return hash((self.type, self.index))

    def __repr__(self):
        return "device(type=%r, index=%r)" % (self.type, self.index)

    def __str__(self):
        return self.type if self.index is None else "%s:%d" % (self.type, self.index)


_CPU = device("cpu")


# ------------------------------------------------------------------------------------------------------------------
# per-device plumbing: context, stream, free lists
# ------------------------------------------------------------------------------------------------------------------
class _Device:
    def __init__(self, index):
        self.index = index
        self.ctx = _lib.Context(index)
        self.lib = self.ctx.lib
        h = ctypes.c_void_p()
        check(self.lib.t2v_stream_create(self.ctx.handle, ctypes.byref(h)), "stream_create")
        self.stream = Stream(self, h.value)
        self.free = collections.defaultdict(list)     # rounded size -> device pointers
        self.events = []
        self.lock = threading.Lock()
        self.staging = []          # ring of [pinned ptr, bytes, event, in flight?] for small pageable uploads
        self.staging_next = 0

    def upload_staged(self, dst_ptr, src_ptr, nbytes):
        """H2D of a small pageable array through a ring of page-locked buffers.  The runtime's own pageable path returns
        only when ITS copy has run -- behind everything the stream still holds -- so a frame loop that uploads the next pose
        map while the previous frame computes would stop running ahead of the GPU (measured on the MI355X box: 1.77 s
        instead of 1.25 s for the 170-frame utterance).  A slot is reused after its own copy's event."""
        i = self.staging_next % STAGING_SLOTS
        self.staging_next += 1
        if i >= len(self.staging):
            self.staging.append([None, 0, None, False])
        slot = self.staging[i]
        if slot[3]:
            check(self.lib.t2v_event_synchronize(self.ctx.handle, ctypes.c_void_p(slot[2])), "event_synchronize")
            slot[3] = False
        if slot[1] < nbytes:
            if slot[0]:
                self.lib.t2v_host_free(self.ctx.handle, ctypes.c_void_p(slot[0]))
            p = ctypes.c_void_p()
            size = max(1 << 20, (nbytes + 4095) // 4096 * 4096)
            check(self.lib.t2v_host_malloc(self.ctx.handle, size, ctypes.byref(p)), "host_malloc")
            slot[0], slot[1] = p.value, size
        if slot[2] is None:
            h = ctypes.c_void_p()
            check(self.lib.t2v_event_create(self.ctx.handle, ctypes.byref(h)), "event_create")
            slot[2] = h.value
        ctypes.memmove(slot[0], src_ptr, nbytes)
        s = ctypes.c_void_p(self.stream.cuda_stream)
        check(self.lib.t2v_memcpy(self.ctx.handle, s, ctypes.c_void_p(dst_ptr), ctypes.c_void_p(slot[0]), nbytes, _lib.COPY_H2D),
              "memcpy h2d")
        check(self.lib.t2v_event_record(self.ctx.handle, ctypes.c_void_p(slot[2]), s), "event_record")
        slot[3] = True

    def malloc(self, nbytes):
        size = max(256, (nbytes + 255) // 256 * 256)
        with self.lock:
            lst = self.free.get(size)
            if lst:
                return lst.pop(), size
        p = ctypes.c_void_p()
        st = self.lib.t2v_device_malloc(self.ctx.handle, size, ctypes.byref(p))
        if st != _lib.T2V_OK:       # out of memory: give the cached blocks back and try once more
            self.release_cached()
            check(self.lib.t2v_device_malloc(self.ctx.handle, size, ctypes.byref(p)), "device_malloc")
        return p.value, size

    def give_back(self, ptr, size):
        with self.lock:
            self.free[size].append(ptr)

    def release_cached(self):
        check(self.lib.t2v_device_synchronize(self.ctx.handle), "device_synchronize")
        with self.lock:
            blocks = [p for lst in self.free.values() for p in lst]
            self.free.clear()
        for p in blocks:
            self.lib.t2v_device_free(self.ctx.handle, ctypes.c_void_p(p))


STAGING_SLOTS = 8
STAGING_MAX_BYTES = 8 << 20     # non_blocking uploads up to this size are staged; everything else takes the runtime's own path

_devices = {}
_current = [0]
_dev_lock = threading.Lock()


def _dev(index=None):
    index = _current[0] if index is None else index
    d = _devices.get(index)
    if d is None:
        with _dev_lock:
            d = _devices.get(index)
            if d is None:
                d = _devices[index] = _Device(index)
    return d


class Stream:
    def __init__(self, dev, handle):
        self._dev, self.cuda_stream = dev, handle or 0

    def synchronize(self):
        check(self._dev.lib.t2v_stream_synchronize(self._dev.ctx.handle, ctypes.c_void_p(self.cuda_stream)), "stream_synchronize")


class Event:
    """torch.cuda.Event(): record() on the current stream, synchronize() on the host.  Handles are recycled."""

    def __init__(self):
        self._dev = _dev()
        with self._dev.lock:
            self._h = self._dev.events.pop() if self._dev.events else None
        if self._h is None:
            h = ctypes.c_void_p()
            check(self._dev.lib.t2v_event_create(self._dev.ctx.handle, ctypes.byref(h)), "event_create")
            self._h = h.value

    def record(self, stream=None):
        s = stream if stream is not None else self._dev.stream
        check(self._dev.lib.t2v_event_record(self._dev.ctx.handle, ctypes.c_void_p(self._h), ctypes.c_void_p(s.cuda_stream)),
              "event_record")

    def synchronize(self):
        check(self._dev.lib.t2v_event_synchronize(self._dev.ctx.handle, ctypes.c_void_p(self._h)), "event_synchronize")

    def __del__(self):
        try:
            with self._dev.lock:
                self._dev.events.append(self._h)
        except Exception:
            pass


class _DeviceScope:
    def __init__(self, dev):
        self.index = device(dev).index if not isinstance(dev, int) else dev

    def __enter__(self):
        self.prev = _current[0]
        if self.index is not None:
            _current[0] = self.index

    def __exit__(self, *exc):
        _current[0] = self.prev


class cuda:
    Event = Event

    @staticmethod
    def is_available():
        try:
            return _lib.load() is not None
        except Exception:
            return False

    @staticmethod
    def current_device():
        return _current[0]

    @staticmethod
    def set_device(dev):
        _current[0] = device(dev).index if not isinstance(dev, int) else dev

    @staticmethod
    def device(dev):
        return _DeviceScope(dev)

    @staticmethod
    def current_stream(dev=None):
        return _dev(None if dev is None else device(dev).index).stream

    @staticmethod
    def synchronize(dev=None):
        d = _dev(None if dev is None else device(dev).index)
        check(d.lib.t2v_device_synchronize(d.ctx.handle), "device_synchronize")

    @staticmethod
    def empty_cache():
        for d in list(_devices.values()):
            d.release_cached()


class no_grad:
    """decorator / context manager; there is no autograd here"""

    def __call__(self, fn):
        return fn

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


# ------------------------------------------------------------------------------------------------------------------
# Tensor
# ------------------------------------------------------------------------------------------------------------------
def _carray(a, dt=None):
    """C-contiguous array of dtype dt, copied only when needed (np.ascontiguousarray would turn 0-d into 1-d)"""
    return np.asarray(a, dtype=dt, order="C")


def _shape(args):
    if len(args) == 1 and isinstance(args[0], (tuple, list)):
        args = tuple(args[0])
    return tuple(int(a) for a in args)


class Tensor:
    """A contiguous buffer with a shape: on the device (`_ptr` from the free list), in page-locked host memory, or a numpy
    array on the host (`_np`)."""

    def __init__(self, shape, dt, where, ptr=None, size=0, dev=None, arr=None):
        self.shape, self.dtype = tuple(shape), dt
        self._where = where          # "cuda" | "pinned" | "cpu"
        self._ptr, self._size, self._dev, self._np = ptr, size, dev, arr
        self._file = None            # host tensors read by load(): (path, byte offset) of their bytes in the file
        self._base = None            # device views: the tensor that owns the memory (size == 0: nothing to give back)

    # -- what the bindings read ----------------------------------------------------------------
    def data_ptr(self):
        if self._where == "cpu":
            return self._np.ctypes.data
        return self._ptr

    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def is_cuda(self):
        return self._where == "cuda"

    @property
    def device(self):
        return device("cuda", self._dev.index) if self._where == "cuda" else _CPU

    def is_contiguous(self):
        return True

    def contiguous(self):
        return self

    def detach(self):
        return self

    def dim(self):
        return len(self.shape)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def nbytes(self):
        return self.numel() * self.dtype.itemsize

    # -- host side ----------------------------------------------------------------------------
    def numpy(self):
        if self._where == "cpu":
            return self._np
        if self._where == "pinned":
            if self._np is None:
                buf = (ctypes.c_ubyte * max(1, self.nbytes())).from_address(self._ptr)
                self._np = np.frombuffer(buf, dtype=self.dtype.np, count=self.numel()).reshape(self.shape)
            return self._np
        raise TypeError("leantorch: numpy() of a device tensor (copy it to the host first)")

    def float(self):
        if self.dtype is float32:
            return self
        if self._where != "cpu":
            raise TypeError("leantorch: float() of a %s %s tensor" % (self._where, self.dtype))
        return from_numpy(_carray(self._np, np.float32))

    def cpu(self):
        if self._where != "cuda":
            return self
        out = from_numpy(np.empty(self.shape, dtype=self.dtype.np))
        d = self._dev
        check(d.lib.t2v_memcpy(d.ctx.handle, ctypes.c_void_p(d.stream.cuda_stream), ctypes.c_void_p(out.data_ptr()),
                               ctypes.c_void_p(self._ptr), self.nbytes(), _lib.COPY_D2H), "memcpy d2h")
        d.stream.synchronize()
        return out

    # -- copies -------------------------------------------------------------------------------
    def to(self, dev=None, dt=None, non_blocking=False, **kw):
        dt = kw.get("dtype", dt)
        if isinstance(dev, dtype):
            dev, dt = None, dev
        src = self
        if dt is not None and dt is not self.dtype:
            if self._where != "cpu":
                raise TypeError("leantorch: dtype conversion on the device")
            src = from_numpy(_carray(self._np, dt.np))
        if dev is None:
            return src
        dev = device(dev)
        if dev.type == "cpu":
            return src.cpu()
        if src._where == "cuda":
            if src._dev.index != dev.index:
                raise RuntimeError("leantorch: device-to-device moves between GPUs are not part of the frame loop")
            return src
        out = empty(src.shape, dtype=src.dtype, device=dev)
        d = out._dev
        if non_blocking and src._where == "cpu" and 0 < src.nbytes() <= STAGING_MAX_BYTES:
            d.upload_staged(out._ptr, src.data_ptr(), src.nbytes())
            return out
        check(d.lib.t2v_memcpy(d.ctx.handle, ctypes.c_void_p(d.stream.cuda_stream), ctypes.c_void_p(out._ptr),
                               ctypes.c_void_p(src.data_ptr()), src.nbytes(), _lib.COPY_H2D), "memcpy h2d")
        out._keep = src          # (a page-locked source is read asynchronously; a pageable one is staged before the call returns)
        return out

    def copy_(self, src, non_blocking=False):
        if src.nbytes() != self.nbytes() or src.dtype is not self.dtype:
            raise ValueError("leantorch: copy_ between different sizes / dtypes")
        a, b = self._where == "cuda", src._where == "cuda"
        if not a and not b:
            np.copyto(self.numpy(), src.numpy().reshape(self.shape))
            return self
        d = self._dev if a else src._dev
        kind = _lib.COPY_D2D if (a and b) else (_lib.COPY_H2D if a else _lib.COPY_D2H)
        check(d.lib.t2v_memcpy(d.ctx.handle, ctypes.c_void_p(d.stream.cuda_stream), ctypes.c_void_p(self.data_ptr()),
                               ctypes.c_void_p(src.data_ptr()), self.nbytes(), kind), "memcpy")
        if kind == _lib.COPY_D2H and not (non_blocking and self._where == "pinned"):
            d.stream.synchronize()
        return self

    def clone(self):
        if self._where == "cpu":
            return from_numpy(self._np.copy())
        out = empty(self.shape, dtype=self.dtype, device=self.device) if self._where == "cuda" else \
            empty(self.shape, dtype=self.dtype, pin_memory=True)
        return out.copy_(self)

    def zero_(self):
        if self._where == "cuda":
            d = self._dev
            check(d.lib.t2v_zero(d.ctx.handle, ctypes.c_void_p(d.stream.cuda_stream), ctypes.c_void_p(self._ptr), self.nbytes()),
                  "zero")
        else:
            self.numpy()[...] = 0
        return self

    def __del__(self):
        try:
            if self._where == "cuda" and self._ptr and self._size:
                self._dev.give_back(self._ptr, self._size)
            elif self._where == "pinned" and self._ptr:
                self._np = None
                self._dev.lib.t2v_host_free(self._dev.ctx.handle, ctypes.c_void_p(self._ptr))
        except Exception:
            pass

    def __repr__(self):
        return "leantorch.Tensor(shape=%s, dtype=%s, %s)" % (self.shape, self.dtype.name, self._where)


def empty(*shape, dtype=float32, device=None, pin_memory=False):       # noqa: A002 -- torch's argument names
    shp = _shape(shape)
    n = 1
    for s in shp:
        n *= s
    dev = globals()["device"](device) if device is not None else _CPU
    if pin_memory:
        d = _dev()
        p = ctypes.c_void_p()
        check(d.lib.t2v_host_malloc(d.ctx.handle, max(1, n * dtype.itemsize), ctypes.byref(p)), "host_malloc")
        return Tensor(shp, dtype, "pinned", ptr=p.value, dev=d)
    if dev.type == "cpu":
        return Tensor(shp, dtype, "cpu", arr=np.empty(shp, dtype=dtype.np))
    d = _dev(dev.index)
    ptr, size = d.malloc(n * dtype.itemsize)
    return Tensor(shp, dtype, "cuda", ptr=ptr, size=size, dev=d)


def zeros(*shape, dtype=float32, device=None):       # noqa: A002
    return empty(*shape, dtype=dtype, device=device).zero_()


def zeros_like(t):
    return zeros(t.shape, dtype=t.dtype, device=t.device)


def empty_like(t):
    return empty(t.shape, dtype=t.dtype, device=t.device)


def from_numpy(a):
    dt = _BY_NP.get(a.dtype)
    if dt is None:
        raise TypeError("leantorch.from_numpy: dtype %s" % a.dtype)
    if not a.flags["C_CONTIGUOUS"]:
        a = _carray(a)
    return Tensor(a.shape, dt, "cpu", arr=a)


def cat(tensors, dim=0):
    """dim 0 only (the flow head and the weight head packed as one 3-output conv)"""
    if dim != 0:
        raise NotImplementedError("leantorch.cat: dim %d" % dim)
    t0 = tensors[0]
    if any(t.shape[1:] != t0.shape[1:] or t.dtype is not t0.dtype or t._where != t0._where for t in tensors):
        raise ValueError("leantorch.cat: mismatched tensors")
    if t0._where == "cpu":
        return from_numpy(np.concatenate([t._np for t in tensors], 0))
    out = empty((sum(t.shape[0] for t in tensors),) + t0.shape[1:], dtype=t0.dtype, device=t0.device)
    d, off = out._dev, 0
    for t in tensors:
        check(d.lib.t2v_memcpy(d.ctx.handle, ctypes.c_void_p(d.stream.cuda_stream), ctypes.c_void_p(out._ptr + off),
                               ctypes.c_void_p(t._ptr), t.nbytes(), _lib.COPY_D2D), "memcpy d2d")
        off += t.nbytes()
    return out


UPLOAD_THREADS = 4
UPLOAD_CHUNK = 8 << 20
# smaller state dicts are not worth the threads and the pinned buffers (T2V_UPLOAD_MIN_SPAN: bytes; tests lower it)
UPLOAD_MIN_SPAN = int(__import__("os").environ.get("T2V_UPLOAD_MIN_SPAN", 64 << 20))
LAST_UPLOAD = {}                # what the last upload_many did (vid2vid/test.py --timing_json reports it)


def upload_many(tensors, dev):
    """{name: host tensor} -> {name: device tensor}.  Tensors that load() mapped from ONE file (a checkpoint: 1.46 GB for the
    full-size generator) go up together: the span of the file that holds them is read by UPLOAD_THREADS threads with
    os.preadv straight into page-locked chunk buffers (the kernel copies out of the page cache; no page of the mapping is
    touched) and sent from there on the threads' own streams into one device slab, of which the tensors are views.  The
    runtime's pageable copy of the mapped tensors -- what `.to(device)` does -- runs at ~6 GB/s on the MI355X box (0.24 s),
    bound by faulting the mapping's pages in on one thread.  Both containers: the zip archive's aligned records are mirrored
    chunk by chunk, the tensors of a legacy stream (an 8-byte count in front of every storage) are sent piece by piece to
    aligned places of their own.  Anything else (other dtypes, small dicts) takes `.to(device)`.  Host-synchronous: the data is on the device when this returns."""
    import os
    dev = device(dev)
    out = {}
    cand = {}
    for k, t in tensors.items():
        f = getattr(t, "_file", None)
        if isinstance(t, Tensor) and t._where == "cpu" and t.dtype is float32 and f is not None and t.nbytes():
            cand.setdefault(f[0], []).append((k, t, f[1]))
    for path, items in cand.items():
        items.sort(key=lambda it: it[2])
        lo = items[0][2] // 256 * 256
        hi = max(off + t.nbytes() for _, t, off in items)
        total = sum(t.nbytes() for _, t, _ in items)
        if hi - lo < UPLOAD_MIN_SPAN or hi - lo > 1.25 * total + (1 << 20):
            continue
        if all(off % 16 == 0 for _, _, off in items):
            # (the zip container aligns its records to 64 bytes) the device slab mirrors the file span: one copy per chunk
            pieces, place, slab_bytes = None, [off - lo for _, _, off in items], hi - lo
        else:
            # (the legacy stream puts an 8-byte count in front of every storage) every tensor gets its own 256-byte-aligned
            # place in the slab; a chunk is sent as the pieces of the tensors it holds
            place, slab_bytes = [], 0
            for _, t, _ in items:
                place.append(slab_bytes)
                slab_bytes += (t.nbytes() + 255) // 256 * 256
            pieces = [(off, t.nbytes(), d_off) for (_, t, off), d_off in zip(items, place)]
        try:
            slab = _upload_file_span(path, lo, hi, dev, pieces, slab_bytes)
        except Exception as e:       # noqa: BLE001 -- e.g. no pinned memory to be had: the plain path below
            LAST_UPLOAD["error"] = "%s: %s" % (type(e).__name__, e)
            continue
        for (k, t, _), d_off in zip(items, place):
            v = Tensor(t.shape, float32, "cuda", ptr=slab._ptr + d_off, size=0, dev=slab._dev)
            v._base = slab
            out[k] = v
    LAST_UPLOAD["as_views"] = len(out)
    for k, t in tensors.items():
        if k not in out:
            out[k] = t.detach().to(dev, float32).contiguous()
    LAST_UPLOAD["tensors"] = len(out)
    return {k: out[k] for k in tensors}


def _upload_file_span(path, lo, hi, dev, pieces, slab_bytes):
    """file bytes [lo, hi) -> a device slab.  pieces None: the slab mirrors the span.  Else pieces = sorted (file offset,
    bytes, slab offset): only those ranges are sent, each to its place."""
    import bisect
    import os
    import time
    t0 = time.perf_counter()
    d = _dev(dev.index)
    slab = empty(slab_bytes, dtype=uint8, device=dev)
    starts = [f for f, _, _ in pieces] if pieces is not None else None
    LAST_UPLOAD.update(span_bytes=hi - lo, slab_bytes=slab_bytes, mirrored=pieces is None,
                       slab_alloc_s=round(time.perf_counter() - t0, 4))
    nchunks = (hi - lo + UPLOAD_CHUNK - 1) // UPLOAD_CHUNK
    nthreads = max(1, min(UPLOAD_THREADS, nchunks, os.cpu_count() or 1))
    errors = []
    fd = os.open(path, os.O_RDONLY)

    def worker(w):
        lib, ctx = d.lib, d.ctx.handle
        stream, slots, views = ctypes.c_void_p(), [], []
        try:
            check(lib.t2v_stream_create(ctx, ctypes.byref(stream)), "stream_create")
            for _ in range(2):
                p, e = ctypes.c_void_p(), ctypes.c_void_p()
                check(lib.t2v_host_malloc(ctx, UPLOAD_CHUNK, ctypes.byref(p)), "host_malloc")
                slots.append([p, None, False])
                check(lib.t2v_event_create(ctx, ctypes.byref(e)), "event_create")
                slots[-1][1] = e
            views = [memoryview((ctypes.c_char * UPLOAD_CHUNK).from_address(sl[0].value)) for sl in slots]
            if w == 0:
                LAST_UPLOAD["thread_setup_s"] = round(time.perf_counter() - t0, 4)
            for j, c in enumerate(range(w, nchunks, nthreads)):
                sl, mv = slots[j % 2], views[j % 2]
                if sl[2]:
                    check(lib.t2v_event_synchronize(ctx, sl[1]), "event_synchronize")
                off = lo + c * UPLOAD_CHUNK
                n = min(UPLOAD_CHUNK, hi - off)
                got = 0
                while got < n:
                    r = os.preadv(fd, [mv[got:n]], off + got)
                    if r <= 0:
                        raise IOError("leantorch.upload_many: short read of %s" % path)
                    got += r
                if pieces is None:
                    check(lib.t2v_memcpy(ctx, stream, ctypes.c_void_p(slab._ptr + c * UPLOAD_CHUNK), sl[0], n, _lib.COPY_H2D), "memcpy h2d")
                else:       # the tensors (or parts of tensors) this chunk [off, off + n) holds
                    i = max(0, bisect.bisect_right(starts, off) - 1)
                    while i < len(pieces) and pieces[i][0] < off + n:
                        f0, fn, d0 = pieces[i]
                        a, b = max(off, f0), min(off + n, f0 + fn)
                        if b > a:
                            check(lib.t2v_memcpy(ctx, stream, ctypes.c_void_p(slab._ptr + d0 + (a - f0)),
                                                 ctypes.c_void_p(sl[0].value + (a - off)), b - a, _lib.COPY_H2D), "memcpy h2d")
                        i += 1
                check(lib.t2v_event_record(ctx, sl[1], stream), "event_record")
                sl[2] = True
            check(lib.t2v_stream_synchronize(ctx, stream), "stream_synchronize")
        except BaseException as e:      # noqa: BLE001 -- re-raised by the caller
            errors.append(e)
        finally:
            del views
            for sl in slots:
                if sl[1]:
                    lib.t2v_event_destroy(ctx, sl[1])
                lib.t2v_host_free(ctx, sl[0])
            if stream:
                lib.t2v_stream_destroy(ctx, stream)

    try:
        threads = [threading.Thread(target=worker, args=(w,)) for w in range(nthreads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        os.close(fd)
    if errors:
        raise errors[0]
    LAST_UPLOAD.update(threads=nthreads, total_s=round(time.perf_counter() - t0, 4))
    return slab


# ------------------------------------------------------------------------------------------------------------------
# torch.save'd state dicts without torch
# ------------------------------------------------------------------------------------------------------------------
_STORAGE_DTYPES = {"FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16,
                   "LongStorage": np.int64, "IntStorage": np.int32, "ByteStorage": np.uint8, "CharStorage": np.int8,
                   "ShortStorage": np.int16, "BoolStorage": np.bool_}
_MAGIC = 0x1950a86a20f9469cfc6c


class _StorageType:
    def __init__(self, name):
        self.name, self.np = name, np.dtype(_STORAGE_DTYPES[name])


class _LazyStorage:
    """a storage named in the pickle; its bytes are located after the pickle has been read"""

    def __init__(self, key, st, numel):
        self.key, self.st, self.numel, self.array = key, st, numel, None
        self.file = None        # (path, byte offset of element 0) once located


class _LazyTensor:
    def __init__(self, storage, offset, size, stride):
        self.storage, self.offset, self.size, self.stride = storage, offset, tuple(size), tuple(stride)
        self.file = None        # contiguous tensors: (path, byte offset) of their first element

    def materialise(self):
        base = self.storage.array
        n = 1
        for s in self.size:
            n *= s
        contiguous, acc = True, 1
        for s, st in zip(reversed(self.size), reversed(self.stride)):
            if s != 1 and st != acc:
                contiguous = False
            acc *= s
        if contiguous:
            self.file = None if self.storage.file is None else \
                (self.storage.file[0], self.storage.file[1] + self.offset * base.dtype.itemsize)
            return base[self.offset:self.offset + n].reshape(self.size)
        isz = base.dtype.itemsize
        v = np.lib.stride_tricks.as_strided(base[self.offset:], shape=self.size, strides=tuple(s * isz for s in self.stride),
                                            writeable=False)
        return _carray(v)


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    return _LazyTensor(storage, storage_offset, size, stride)


def _rebuild_tensor(storage, storage_offset, size, stride):
    return _LazyTensor(storage, storage_offset, size, stride)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None, *rest):
    return data


class _Unpickler(pickle.Unpickler):
    """resolves exactly the globals a state dict of tensors needs; anything else is refused (as weights_only=True does)"""

    def __init__(self, fh, storages):
        super().__init__(fh, encoding="utf-8")
        self._storages = storages

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return collections.OrderedDict
        if module == "torch._utils":
            fn = {"_rebuild_tensor_v2": _rebuild_tensor_v2, "_rebuild_tensor": _rebuild_tensor,
                  "_rebuild_parameter": _rebuild_parameter}.get(name)
            if fn is not None:
                return fn
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "torch" and name == "Size":
            return tuple
        raise pickle.UnpicklingError("leantorch.load: refusing global %s.%s (tensors-only state dicts)" % (module, name))

    def persistent_load(self, pid):
        if not isinstance(pid, tuple) or pid[0] != "storage":
            raise pickle.UnpicklingError("leantorch.load: unknown persistent id %r" % (pid,))
        st, key, numel = pid[1], str(pid[2]), int(pid[4])
        s = self._storages.get(key)
        if s is None:
            s = self._storages[key] = _LazyStorage(key, st, numel)
        if len(pid) > 5 and pid[5] is not None:        # legacy view_metadata: (view_key, offset, view_size)
            raise pickle.UnpicklingError("leantorch.load: storage views")
        return s


def _materialise(obj):
    if isinstance(obj, _LazyTensor):
        t = from_numpy_any(obj.materialise())
        if isinstance(t, Tensor):
            t._file = obj.file
        return t
    if isinstance(obj, collections.OrderedDict):
        return collections.OrderedDict((k, _materialise(v)) for k, v in obj.items())
    if isinstance(obj, dict):
        return {k: _materialise(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_materialise(v) for v in obj)
    return obj


class _HostArray:
    """a checkpoint tensor of a dtype the device path has no use for (int64 step counters): .float() converts"""

    def __init__(self, a):
        self._a, self.shape = a, a.shape

    def float(self):
        return from_numpy(_carray(self._a, np.float32))

    def numpy(self):
        return self._a


def from_numpy_any(a):
    return from_numpy(a) if a.dtype in _BY_NP else _HostArray(a)


def _load_zip(path, mm):
    with zipfile.ZipFile(path) as zf:
        infos = {i.filename: i for i in zf.infolist()}
        pkl = [n for n in infos if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pkl) != 1:
            raise pickle.UnpicklingError("leantorch.load: %s is not a torch.save archive" % path)
        prefix = pkl[0][:-len("data.pkl")]
        bo = infos.get(prefix + "byteorder")
        if bo is not None and zf.read(bo).strip() != b"little":
            raise pickle.UnpicklingError("leantorch.load: big-endian archive")
        storages = {}
        obj = _Unpickler(io.BytesIO(zf.read(pkl[0])), storages).load()
        for key, s in storages.items():
            info = infos.get(prefix + "data/" + key)
            if info is None or info.compress_type != zipfile.ZIP_STORED:
                raise pickle.UnpicklingError("leantorch.load: storage %s missing or compressed" % key)
            sig, _, _, _, _, _, _, _, _, n_name, n_extra = struct.unpack("<4s5H3L2H", mm[info.header_offset:info.header_offset + 30])
            if sig != b"PK\x03\x04":
                raise pickle.UnpicklingError("leantorch.load: bad local header for storage %s" % key)
            off = info.header_offset + 30 + n_name + n_extra
            if info.file_size < s.numel * s.st.np.itemsize:
                raise pickle.UnpicklingError("leantorch.load: storage %s is short" % key)
            s.array = np.frombuffer(mm, dtype=s.st.np, count=s.numel, offset=off)
            s.file = (path, off)
    return obj


class _PlainUnpickler(pickle.Unpickler):
    """for the four header / trailer records of a legacy stream (an int, an int, a dict of ints and bools, a list of
    str): no global resolves and no persistent id is accepted, so a __reduce__ payload there raises before anything
    of it runs"""

    def find_class(self, module, name):
        raise pickle.UnpicklingError("leantorch.load: refusing global %s.%s in a stream header" % (module, name))

    def persistent_load(self, pid):
        raise pickle.UnpicklingError("leantorch.load: persistent id in a stream header")


def _plain(fh, kind):
    obj = _PlainUnpickler(fh, encoding="utf-8").load()

    def ok(o, depth=0):
        if isinstance(o, (bool, int, float, str, bytes, type(None))):
            return True
        if depth < 4 and isinstance(o, (list, tuple)):
            return all(ok(v, depth + 1) for v in o)
        if depth < 4 and isinstance(o, dict):
            return all(ok(k, depth + 1) and ok(v, depth + 1) for k, v in o.items())
        return False
    if not isinstance(obj, kind) or not ok(obj):
        raise pickle.UnpicklingError("leantorch.load: malformed stream header (%s)" % type(obj).__name__)
    return obj


# the magic number as pickle.dump writes it, every protocol torch.save's pickle_protocol argument admits: compared as
# raw bytes, before anything in the file is unpickled (torch/serialization.py:286-300 of 0.4.1 writes it first)
_MAGIC_BYTES = tuple(sorted({pickle.dumps(_MAGIC, protocol=p) for p in range(0, pickle.HIGHEST_PROTOCOL + 1)},
                            key=len, reverse=True))


def _load_legacy(path, mm):
    head = bytes(mm[:max(len(b) for b in _MAGIC_BYTES)]) if len(mm) else b""
    magic = next((b for b in _MAGIC_BYTES if head.startswith(b)), None)
    if magic is None:
        raise pickle.UnpicklingError("leantorch.load: %s is neither a zip archive nor a legacy torch.save stream" % path)
    with open(path, "rb") as fh:
        fh.seek(len(magic))
        _plain(fh, int)       # protocol version
        _plain(fh, dict)      # sys_info
        storages = {}
        obj = _Unpickler(fh, storages).load()
        keys = _plain(fh, (list, tuple))
        off = fh.tell()
    for key in keys:
        s = storages.get(str(key))
        (numel,) = struct.unpack("<q", mm[off:off + 8])
        off += 8
        if s is not None:
            if numel != s.numel:
                raise pickle.UnpicklingError("leantorch.load: storage %s: %d elements, header says %d" % (key, s.numel, numel))
            s.array = np.frombuffer(mm, dtype=s.st.np, count=numel, offset=off)
            s.file = (path, off)
            off += numel * s.st.np.itemsize
        else:
            raise pickle.UnpicklingError("leantorch.load: unreferenced storage %s" % key)
    return obj


def load(path, map_location="cpu", weights_only=True, mmap=True):
    """torch.load for tensors-only state dicts (both containers); every tensor a view of one read-only memory map"""
    import mmap as _mmap
    with open(path, "rb") as fh:
        head = fh.read(4)
        mm = _mmap.mmap(fh.fileno(), 0, access=_mmap.ACCESS_READ)
    obj = _load_zip(path, mm) if head == b"PK\x03\x04" else _load_legacy(path, mm)
    return _materialise(obj)
