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
    assert a == resident.socket_path(["--name", "x"]) and a.startswith("/tmp/t2v_resident_%d_" % os.getuid())
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
