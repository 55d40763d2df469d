"""text2video_amd/leantorch.py on the CPU: the torch-free checkpoint reader against torch.load on both containers, and the
Tensor / allocator / event logic over a stand-in for the library's host-plumbing entry points (plain host memory)."""
import collections
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from text2video_amd import _lib, leantorch as lt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _state_dict():
    g = torch.Generator().manual_seed(0)
    sd = collections.OrderedDict()
    sd["model.0.weight"] = torch.randn(8, 4, 3, 3, generator=g)
    sd["model.0.bias"] = torch.randn(8, generator=g)
    sd["model.1.num_batches_tracked"] = torch.tensor(7)
    sd["half.weight"] = torch.randn(6, 5, generator=g).half()
    sd["transposed.weight"] = torch.randn(6, 5, generator=g).t()            # not contiguous
    sd["param"] = torch.nn.Parameter(torch.randn(3, generator=g))
    big = torch.randn(100000, generator=g)
    sd["view1"], sd["view2"] = big[10:20], big[500:1000]                     # one storage, two offsets
    return sd


@pytest.mark.parametrize("legacy", [False, True], ids=["zip", "legacy-0.4.1-stream"])
def test_reader_equals_torch_load(tmp_path, legacy):
    sd = _state_dict()
    path = str(tmp_path / "net.pth")
    torch.save(sd, path, _use_new_zipfile_serialization=not legacy)
    want = torch.load(path, map_location="cpu", weights_only=True)
    got = lt.load(path, map_location="cpu", weights_only=True, mmap=True)
    assert isinstance(got, collections.OrderedDict) and list(got) == list(want)
    for k, v in want.items():
        a = got[k].float().numpy()
        assert a.dtype == np.float32 and a.shape == tuple(v.shape), k
        assert np.array_equal(a, v.detach().float().numpy()), k
    # fp32 tensors are views of the file's memory map, not copies
    assert not got["model.0.weight"].numpy().flags["OWNDATA"]


def test_reader_wrapped_state_dict_and_refusals(tmp_path):
    path = str(tmp_path / "wrapped.pth")
    torch.save({"state_dict": _state_dict(), "epoch": 3}, path)
    got = lt.load(path)
    assert got["epoch"] == 3 and np.array_equal(got["state_dict"]["model.0.bias"].numpy(), _state_dict()["model.0.bias"].numpy())
    # anything but tensors is refused, as weights_only=True refuses it
    bad = str(tmp_path / "bad.pth")
    torch.save({"fn": os.path.join}, bad)
    with pytest.raises(Exception, match="refusing global"):
        lt.load(bad)
    junk = str(tmp_path / "junk.pth")
    with open(junk, "wb") as fh:
        fh.write(b"\x80\x02K\x01.")          # a pickle, but not torch's magic number
    with pytest.raises(Exception, match="neither a zip archive nor a legacy"):
        lt.load(junk)


class _FakePlumbing:
    """host memory behind the ABI-14 entry points (include/t2v.h): enough to run leantorch's Python on a CPU box"""

    def __init__(self):
        self.live, self.copies, self.events = {}, [], 0
        self._libc = ctypes.CDLL(None)
        self._libc.malloc.restype = ctypes.c_void_p
        self._libc.malloc.argtypes = [ctypes.c_size_t]
        self._libc.free.argtypes = [ctypes.c_void_p]

    def _malloc(self, ctx, n, out):
        p = self._libc.malloc(max(1, n))
        self.live[p] = n
        ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = p
        return 0

    def _free(self, ctx, p):
        self.live.pop(p.value, None)
        self._libc.free(p)
        return 0

    t2v_device_malloc = t2v_host_malloc = _malloc
    t2v_device_free = t2v_host_free = _free

    def t2v_memcpy(self, ctx, stream, dst, src, n, kind):
        ctypes.memmove(dst, src, n)
        self.copies.append((kind, n))
        return 0

    def t2v_zero(self, ctx, stream, p, n):
        ctypes.memset(p, 0, n)
        return 0

    def t2v_stream_create(self, ctx, out):
        ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = 0x5EED
        return 0

    def t2v_event_create(self, ctx, out):
        self.events += 1
        ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = 0xE000 + self.events
        return 0

    def t2v_stream_synchronize(self, *a):
        return 0

    t2v_event_record = t2v_event_synchronize = t2v_device_synchronize = t2v_stream_synchronize
    t2v_stream_destroy = t2v_event_destroy = t2v_stream_synchronize


@pytest.fixture
def fake(monkeypatch):
    plumbing = _FakePlumbing()

    class Ctx:
        def __init__(self, device=0):
            self.lib, self.handle, self.device = plumbing, ctypes.c_void_p(1), device
    monkeypatch.setattr(_lib, "Context", Ctx)
    monkeypatch.setattr(lt, "_devices", {})
    yield plumbing
    lt._devices.clear()


def test_tensor_round_trips_and_free_list(fake):
    dev = lt.device("cuda:0")
    assert dev == "cuda:0" and dev.index == 0 and str(dev) == "cuda:0" and lt.device("cuda").index == 0
    a = np.arange(2 * 3 * 4, dtype=np.uint8).reshape(2, 3, 4)
    d = lt.from_numpy(a).to(dev)
    assert d.is_cuda and d.shape == (2, 3, 4) and d.dtype is lt.uint8 and d.numel() == 24 and d.device == dev
    assert np.array_equal(d.cpu().numpy(), a)
    host = lt.empty(d.shape, dtype=lt.uint8, pin_memory=True)
    host.copy_(d, non_blocking=True)
    assert np.array_equal(host.numpy(), a) and np.array_equal(host.numpy()[..., :3], a[..., :3])
    z = lt.zeros(4, 5, 8, dtype=lt.float32, device=dev)
    assert not z.cpu().numpy().any() and lt.zeros_like(z).shape == (4, 5, 8)
    c = d.clone()
    assert c.data_ptr() != d.data_ptr() and np.array_equal(c.cpu().numpy(), a)
    # float conversion happens on the host; the device side only ever sees fp32 / uint8
    h = lt.from_numpy(np.ones((2, 2), np.float16)).float().to(dev, lt.float32)
    assert h.dtype is lt.float32 and np.array_equal(h.cpu().numpy(), np.ones((2, 2), np.float32))
    w = lt.cat([lt.from_numpy(np.full((2, 3), 1, np.float32)).to(dev), lt.from_numpy(np.full((1, 3), 2, np.float32)).to(dev)], 0)
    assert w.shape == (3, 3) and np.array_equal(w.cpu().numpy(), np.array([[1] * 3, [1] * 3, [2] * 3], np.float32))
    # a collected buffer is handed out again for the next request of its (rounded) size; nothing is freed meanwhile
    p0 = lt.empty(1000, dtype=lt.float32, device=dev)
    ptr, live = p0.data_ptr(), len(fake.live)
    del p0
    p1 = lt.empty(990, dtype=lt.float32, device=dev)      # 3960 B rounds up to the same 4096 B block
    assert p1.data_ptr() == ptr and len(fake.live) == live
    lt.cuda.empty_cache()
    assert len(fake.live) < live + 1
    with pytest.raises(TypeError):
        d.numpy()
    # non_blocking uploads of small pageable arrays go through a ring of page-locked staging buffers (one event per slot)
    ups = [lt.from_numpy(np.full((5, 7, 3), i, np.uint8)).to(dev, non_blocking=True) for i in range(2 * lt.STAGING_SLOTS + 1)]
    assert all(np.array_equal(u.cpu().numpy(), np.full((5, 7, 3), i, np.uint8)) for i, u in enumerate(ups))
    assert len(lt._dev(0).staging) == lt.STAGING_SLOTS and fake.events >= lt.STAGING_SLOTS


@pytest.mark.parametrize("legacy", [False, True], ids=["zip", "legacy-0.4.1-stream"])
def test_checkpoint_goes_up_as_one_file_span(fake, monkeypatch, tmp_path, legacy):
    """upload_many: the tensors load() mapped from one file are read by several threads into page-locked chunks (os.preadv)
    and become views of one device slab; what does not qualify (a legacy stream's unaligned storages, other dtypes) takes
    .to(device).  Same values either way."""
    monkeypatch.setattr(lt, "UPLOAD_CHUNK", 4096)
    monkeypatch.setattr(lt, "UPLOAD_MIN_SPAN", 1024)
    g = torch.Generator().manual_seed(3)
    sd = collections.OrderedDict(("layer%d.weight" % i, torch.randn(7 + i, 5, 3, 3, generator=g)) for i in range(12))
    sd["layer3.bias"] = torch.randn(9, generator=g)
    path = str(tmp_path / "net.pth")
    torch.save(sd, path, _use_new_zipfile_serialization=not legacy)
    host = {k: v.float() for k, v in lt.load(path).items()}
    up = lt.upload_many(host, "cuda:0")
    assert list(up) == list(host)
    for k, v in sd.items():
        assert up[k].is_cuda and up[k].shape == tuple(v.shape) and np.array_equal(up[k].cpu().numpy(), v.numpy()), k
    views = [t for t in up.values() if t._base is not None]
    assert len(views) == len(up) and len({id(t._base) for t in views}) == 1
    assert lt.LAST_UPLOAD["mirrored"] is (not legacy) and "error" not in lt.LAST_UPLOAD
    if legacy:      # an 8-byte count sits in front of every storage: each tensor gets an aligned place of its own
        assert all(t.data_ptr() % 256 == views[0]._base.data_ptr() % 256 for t in views)
    else:
        assert any(kind == 1 and n == 4096 for kind, n in fake.copies)        # mirrored chunk by chunk
    w = lt.cat([up["layer0.weight"], up["layer0.weight"]], 0)       # (views work as kernel / copy operands)
    assert np.array_equal(w.cpu().numpy()[:7], sd["layer0.weight"].numpy())


def test_events_streams_and_decorator(fake):
    s = lt.cuda.current_stream()
    assert s.cuda_stream == 0x5EED and lt.cuda.current_stream() is s
    e = lt.cuda.Event()
    e.record()
    e.synchronize()
    h = e._h
    del e
    assert lt.cuda.Event()._h == h and fake.events == 1       # handles are recycled
    lt.cuda.synchronize("cuda:0")
    with lt.cuda.device(lt.device("cuda:0")):
        assert lt.cuda.current_device() == 0

    @lt.no_grad()
    def f(x):
        return x + 1
    assert f(1) == 2
    with lt.no_grad():
        pass


def test_provider_choice_is_per_process():
    """use_lean() before the first import of the model selects leantorch and torch stays out of the process; a process that
    has torch already keeps it"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from text2video_amd import _xp\n"
            "assert _xp.use_lean() is True\n"
            "from text2video_amd import ops, generator, model\n"
            "assert ops.torch.__name__ == 'text2video_amd.leantorch' and model.torch is ops.torch is generator.torch\n"
            "assert 'torch' not in sys.modules, 'torch was imported'\n"
            "sd = generator.synthetic_state_dict(generator.GeneratorSpec(ngf=8, n_downsample=1, n_blocks=1), 1)\n"
            "assert all(v.dtype is ops.torch.float32 for v in sd.values())\n"
            "print('lean ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "lean ok" in r.stdout, r.stderr[-2000:]
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from text2video_amd import _xp\n"
            "assert _xp.use_lean() is False\n"
            "from text2video_amd import ops\n"
            "assert ops.torch is torch\n"
            "print('torch ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "torch ok" in r.stdout, r.stderr[-2000:]


def test_unreadable_checkpoint_raises_lean_unsupported(tmp_path):
    """a state dict the torch-free reader refuses (anything but tensors) surfaces as model.LeanUnsupported in the lean
    process -- vid2vid/test.py then starts the command over with torch; a readable one comes back as fp32 host tensors with
    upstream's `module.` prefix and BatchNorm running statistics dropped"""
    bad = str(tmp_path / "bad.pth")
    torch.save({"model.0.weight": torch.zeros(2, 2), "fn": os.path.join}, bad)
    good = str(tmp_path / "good.pth")
    torch.save({"module.model.0.weight": torch.ones(2, 3).half(), "module.model.1.running_mean": torch.zeros(3)}, good)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from text2video_amd import _xp\n"
            "assert _xp.use_lean()\n"
            "from text2video_amd import model\n"
            "sd = model.load_checkpoint(%r)\n"
            "assert list(sd) == ['model.0.weight'] and sd['model.0.weight'].dtype is model.torch.float32, sd\n"
            "assert sd['model.0.weight'].numpy().tolist() == [[1.0] * 3] * 2\n"
            "try:\n"
            "    model.load_checkpoint(%r)\n"
            "except model.LeanUnsupported as e:\n"
            "    print('lean unsupported:', e)\n"
            "assert 'torch' not in sys.modules\n" % (ROOT, good, bad))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "lean unsupported:" in r.stdout and "refusing global" in r.stdout, r.stdout + r.stderr[-2000:]


class _Payload:
    """unpickling this object runs os.mkdir(<marker>): the side effect the reader must never perform"""

    def __init__(self, marker):
        self.marker = marker

    def __reduce__(self):
        return (os.mkdir, (self.marker,))


@pytest.mark.parametrize("position", ["magic", "protocol", "sys_info", "keys", "legacy-body", "zip-data.pkl"])
def test_crafted_checkpoint_never_executes(tmp_path, position):
    """a __reduce__ payload in any of the four header / trailer pickles of the legacy 0.4.1 stream
    ($SP/torch/serialization.py:286-300: magic, protocol, sys_info, <state dict>, storage keys), in its body, or in a zip
    archive's data.pkl is refused AND its side effect is absent -- the torch-free reader is vid2vid/test.py's default"""
    import io
    import pickle
    import zipfile
    marker = str(tmp_path / "pwned")
    path = str(tmp_path / "crafted.pth")
    evil = _Payload(marker)
    if position == "zip-data.pkl":
        with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as zf:
            zf.writestr("archive/data.pkl", pickle.dumps({"w": evil}, protocol=2))
            zf.writestr("archive/version", b"3\n")
    else:
        good = io.BytesIO()
        torch.save(_state_dict(), good, _use_new_zipfile_serialization=False)
        good.seek(0)
        recs = []
        for _ in range(3):                           # magic, protocol, sys_info
            a = good.tell()
            pickle.load(good)
            recs.append(good.getvalue()[a:good.tell()])
        a = good.tell()

        class _Skip(pickle.Unpickler):               # the state dict needs torch's persistent_load to be skipped over
            def persistent_load(self, pid):
                return None

            def find_class(self, module, name):
                if module.startswith("torch"):
                    return lambda *a, **k: None
                return pickle.Unpickler.find_class(self, module, name)
        _Skip(good).load()
        recs.append(good.getvalue()[a:good.tell()])
        a = good.tell()
        pickle.load(good)
        recs.append(good.getvalue()[a:good.tell()])
        tail = good.getvalue()[good.tell():]
        idx = {"magic": 0, "protocol": 1, "sys_info": 2, "legacy-body": 3, "keys": 4}[position]
        recs[idx] = pickle.dumps(evil if position != "legacy-body" else {"w": evil}, protocol=2)
        with open(path, "wb") as fh:
            fh.write(b"".join(recs) + tail)
    with pytest.raises(pickle.UnpicklingError):
        lt.load(path)
    assert not os.path.exists(marker), "the payload in the %s record ran" % position
    # the same file through the product's entry (lean process): LeanUnsupported, side effect still absent
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from text2video_amd import _xp\n"
            "assert _xp.use_lean()\n"
            "from text2video_amd import model\n"
            "try:\n"
            "    model.load_checkpoint(%r)\n"
            "except model.LeanUnsupported as e:\n"
            "    print('lean unsupported:', e)\n" % (ROOT, path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "lean unsupported:" in r.stdout, r.stdout + r.stderr[-2000:]
    assert not os.path.exists(marker)


def test_short_or_empty_checkpoint_is_lean_unsupported(tmp_path):
    """a load failure of any type (an empty file cannot even be mapped) becomes LeanUnsupported in the lean
    process, so vid2vid/test.py starts over with torch instead of dying with a raw exception"""
    empty = str(tmp_path / "empty.pth")
    open(empty, "wb").close()
    short = str(tmp_path / "short.pth")
    with open(short, "wb") as fh:
        fh.write(b"PK\x03\x04trunc")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from text2video_amd import _xp\n"
            "assert _xp.use_lean()\n"
            "from text2video_amd import model\n"
            "for p in %r:\n"
            "    try:\n"
            "        model.load_checkpoint(p)\n"
            "    except model.LeanUnsupported as e:\n"
            "        print('lean unsupported:', e)\n" % (ROOT, [empty, short]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.count("lean unsupported:") == 2, r.stdout + r.stderr[-2000:]
