"""CPU tests of the resident-server client side (text2video_amd/resident.py): the framing the thin `test.py --resident` client
speaks, its fallbacks, and what keys a server / a resident model.  The server side needs the GPU: tests/test_gpu_e2e.py."""
import json
import os
import socket
import struct
import sys
import threading

import pytest


def _fake_server(path, reply_frames, seen):
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(path)
    srv.listen(1)

    def run():
        conn, _ = srv.accept()
        with conn:
            line = bytearray()
            while not line.endswith(b"\n"):
                line += conn.recv(4096)
            seen.append(json.loads(line.decode()))
            for tag, payload in reply_frames:
                if tag == b"x":
                    conn.sendall(b"x" + struct.pack("<i", payload))
                else:
                    conn.sendall(tag + struct.pack("<I", len(payload)) + payload)
        srv.close()
    t = threading.Thread(target=run, daemon=True)
    t.start()
    return t


def test_client_streams_the_servers_output_and_returns_its_status(tmp_path, monkeypatch, capsys):
    from text2video_amd import resident
    path = str(tmp_path / "s.sock")
    monkeypatch.setattr(resident, "socket_path", lambda argv: path)
    monkeypatch.setenv("T2V_STREAMS", "1")
    seen = []
    t = _fake_server(path, [(b"o", b"process image... a.jpg\n"), (b"e", b"warning: x\n"), (b"o", b"done\n"), (b"x", 3)], seen)
    rc = resident.client(["--name", "fadg0", "--resident"])
    t.join(5)
    out = capsys.readouterr()
    assert rc == 3 and out.out == "process image... a.jpg\ndone\n" and "warning: x" in out.err
    # the request carries the command line, the working directory and the T2V_* environment -- nothing else is needed
    assert seen[0]["argv"] == ["--name", "fadg0", "--resident"] and seen[0]["cwd"] == os.getcwd()
    assert seen[0]["env"].get("T2V_STREAMS") == "1" and all(k.startswith("T2V_") for k in seen[0]["env"])


def test_client_reports_a_lost_connection_and_does_not_start_a_server_just_to_stop_it(tmp_path, monkeypatch, capsys):
    from text2video_amd import resident
    path = str(tmp_path / "s.sock")
    monkeypatch.setattr(resident, "socket_path", lambda argv: path)
    assert resident.client(["--resident_stop"]) == 0                      # nothing running: nothing to do, nothing spawned
    assert not os.path.exists(path)
    seen = []
    t = _fake_server(path, [(b"o", b"half a")], seen)                     # the server dies mid-run
    rc = resident.client(["--resident"])
    t.join(5)
    assert rc == 1 and "connection to the server lost" in capsys.readouterr().err


def test_server_and_model_keys(tmp_path, monkeypatch):
    from text2video_amd import resident
    from text2video_amd.options import TestOptions
    monkeypatch.delenv("T2V_RESIDENT_KEY", raising=False)
    a = resident.socket_path(["--gpu_ids", "0"])
    assert a == resident.socket_path(["--name", "x"]) and os.path.dirname(a) == resident.run_dir()
    assert resident.socket_path(["--gpu_ids", "1"]) != a                  # another device: another server
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "3")
    assert resident.socket_path(["--gpu_ids", "0"]) != a
    monkeypatch.setenv("T2V_RESIDENT_KEY", "mine")
    assert resident.socket_path(["--gpu_ids", "0"]) != a
    # a resident model serves a request only if the checkpoint files and the architecture flags are the ones it was built from
    ck = tmp_path / "ckpt" / "fadg0"
    ck.mkdir(parents=True)
    f = ck / "latest_net_G0.pth"
    f.write_bytes(b"0" * 10)
    argv = ["--name", "fadg0", "--checkpoints_dir", str(tmp_path / "ckpt"), "--openpose_only", "--no_first_img"]
    k0 = resident.model_key(TestOptions().parse(argv))
    assert k0 == resident.model_key(TestOptions().parse(argv + ["--how_many", "7", "--dataroot", "elsewhere"]))
    assert k0 != resident.model_key(TestOptions().parse(argv + ["--ngf", "64"]))
    assert k0 != resident.model_key(TestOptions().parse(argv + ["--synthetic_weights", "2"]))
    f.write_bytes(b"0" * 11)                                              # the file changed: reload
    assert k0 != resident.model_key(TestOptions().parse(argv))


def test_thin_client_imports_no_torch():
    """`python test.py --resident` must not pay for `import torch` (1.1 s of the 1.3 s a warm call takes otherwise)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import text2video_amd.resident, text2video_amd.options; "
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]


def test_run_dir_is_private_and_refuses_a_planted_one(tmp_path, monkeypatch):
    """socket, log and lock live in a directory only this user can enter; a symlink or a foreign directory planted at
    that name is refused, a log planted as a symlink is not followed"""
    import stat
    from text2video_amd import resident
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(tmp_path))
    d = resident.run_dir()
    assert d == str(tmp_path / "t2v_resident") and stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    os.chmod(d, 0o755)
    assert resident.run_dir() == d and stat.S_IMODE(os.lstat(d).st_mode) == 0o700       # widened: narrowed again
    os.rmdir(d)
    os.symlink(str(tmp_path), d)                                          # planted symlink
    with pytest.raises(PermissionError):
        resident.run_dir()
    os.unlink(d)
    # the spawn path opens the log O_NOFOLLOW: a planted symlink makes the open fail instead of appending to its target
    victim = tmp_path / "victim.txt"
    victim.write_text("keep")
    d = resident.run_dir()
    monkeypatch.setattr(resident, "socket_path", lambda argv: os.path.join(d, "k.sock"))
    os.symlink(str(victim), os.path.join(d, "k.log"))
    monkeypatch.setattr(resident.subprocess, "Popen", lambda *a, **k: (_ for _ in ()).throw(AssertionError("spawned")))
    with pytest.raises(OSError):
        resident.client(["--resident"], start_timeout=0.2)
    assert victim.read_text() == "keep"


def test_client_takes_over_when_the_server_says_lean_unsupported(tmp_path, monkeypatch, capsys):
    """a checkpoint the torch-free server cannot read: the status frame tells the client to run the command itself (None), where
    vid2vid/test.py's LeanUnsupported handling starts it over with torch"""
    from text2video_amd import resident
    path = str(tmp_path / "s.sock")
    monkeypatch.setattr(resident, "socket_path", lambda argv: path)
    seen = []
    t = _fake_server(path, [(b"e", b"resident: x.pth: refusing global -- the client runs this one itself\n"),
                            (b"x", resident.RC_LEAN_UNSUPPORTED)], seen)
    assert resident.client(["--resident"]) is None
    t.join(5)
    assert "the client runs this one itself" in capsys.readouterr().err


def test_two_simultaneous_first_calls_start_one_server(tmp_path, monkeypatch):
    """the spawn is serialised by a lock file: the second caller finds the first one's server instead of starting its own"""
    from text2video_amd import resident
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(tmp_path))
    path = os.path.join(resident.run_dir(), "k.sock")
    monkeypatch.setattr(resident, "socket_path", lambda argv: path)
    spawned, seen = [], []

    class _P:
        def __init__(self, cmd, **kw):
            spawned.append(cmd)
            # the "server": answers two requests, one after the other
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(path)
            srv.listen(4)

            def run():
                for _ in range(2):
                    conn, _a = srv.accept()
                    with conn:
                        line = bytearray()
                        while not line.endswith(b"\n"):
                            line += conn.recv(4096)
                        seen.append(1)
                        conn.sendall(b"x" + struct.pack("<i", 0))
                srv.close()
            threading.Thread(target=run, daemon=True).start()
    monkeypatch.setattr(resident.subprocess, "Popen", _P)
    rcs = []
    ts = [threading.Thread(target=lambda: rcs.append(resident.client(["--resident"], start_timeout=10))) for _ in range(2)]
    [t.start() for t in ts]
    [t.join(20) for t in ts]
    assert rcs == [0, 0] and len(spawned) == 1 and len(seen) == 2


def test_a_run_directory_owned_by_somebody_else_falls_back_to_the_in_process_run(monkeypatch, capsys):
    """any local user can pre-create /tmp/t2v_resident_<uid>; the client must then hand the call back to the caller
    (None: run the frame loop in this process) instead of crashing on the PermissionError -- and --resident_stop is a no-op."""
    from text2video_amd import resident

    def refuse():
        raise PermissionError("resident: /tmp/t2v_resident_0 is not a directory owned by uid 0")
    monkeypatch.setattr(resident, "run_dir", refuse)
    assert resident.client(["--name", "x"]) is None
    assert resident.client(["--name", "x", "--resident_stop"]) == 0
    assert "running in this process" in capsys.readouterr().err
