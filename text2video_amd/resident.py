"""Resident frame-synthesis server behind the one-shot `vid2vid/test.py` command (opt-in: --resident or T2V_RESIDENT=1).

The reference starts a fresh `python test.py ...` per utterance (/root/reference/text2video_audio.sh:37-44,
text2video_tts.sh:45, text2video_tts_chinese.sh:35).  On the MI355X path an utterance of 2 x 85 frames is 1.35 s of frame loop
behind ~1.7 s of start-up that no single process can avoid: the interpreter importing torch (1.1 s) and the 1.5 GB checkpoint
going to the device (0.4 s).  With --resident the command becomes a thin client (no torch import: ~30 ms) of a server process
that keeps the HIP context, the packed / Winograd-transformed weights and the rasteriser workers across calls:

    client (test.py)  --AF_UNIX <run dir>/<key>.sock-->  server (this module, `python -m text2video_amd.resident`)
      request : one JSON line {argv, cwd, env (T2V_*)}
      reply   : frames  b"o" | b"e" + u32 length + bytes (the run's stdout / stderr, streamed) ... b"x" + i32 exit status

<run dir> is a directory only this user can enter ($XDG_RUNTIME_DIR/t2v_resident, else /tmp/t2v_resident_<uid>: mode 0700,
owner checked, never a symlink); the log beside the socket is opened O_NOFOLLOW, mode 0600; both ends check the peer's uid
(SO_PEERCRED); a lock file (flock) makes two simultaneous first calls start ONE server.
The first call finds no server, starts one (detached, its own session) and is served by it; later calls connect.
The server handles one request at a time (the calls of the reference's scripts are sequential), reloads a checkpoint whose
file changed, and exits after `--resident_idle_s` seconds without a request (default 600).  `key` = interpreter, package
location, device selection: two installations or two GPUs never share a server.  test_fifo.py (a named pipe, the model loaded
once) is the reference-named variant of the same idea; this one needs no change to the calling scripts beyond the flag / the
environment variable.
"""
import hashlib
import json
import os
import socket
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVICE_ENV = ("CUDA_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")
RC_LEAN_UNSUPPORTED = -0x4C45414E          # exit-status frame meaning "run it yourself" (never a process exit status)


def _gpu_ids(argv):
    for i, a in enumerate(argv):
        if a == "--gpu_ids" and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith("--gpu_ids="):
            return a.split("=", 1)[1]
    return "0"


def run_dir():
    """a directory only this user can enter: the socket, the log and the spawn lock live here (never in shared /tmp itself)"""
    base = os.environ.get("XDG_RUNTIME_DIR")
    d = os.path.join(base, "t2v_resident") if base and os.path.isdir(base) else "/tmp/t2v_resident_%d" % os.getuid()
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
        raise PermissionError("resident: %s is not a directory owned by uid %d" % (d, os.getuid()))
    if st.st_mode & 0o077:
        os.chmod(d, 0o700)
    return d


def socket_path(argv):
    # (T2V_RESIDENT_KEY: any string; lets independent servers coexist on one device selection)
    key = json.dumps([sys.executable, ROOT, {k: os.environ.get(k, "") for k in DEVICE_ENV}, _gpu_ids(argv),
                      os.environ.get("T2V_RESIDENT_KEY", "")])
    return os.path.join(run_dir(), "%s.sock" % hashlib.sha1(key.encode()).hexdigest()[:16])


def _peer_uid(s):
    pid, uid, gid = struct.unpack("3i", s.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))
    return uid


def _connect(path, timeout=None):
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        s.settimeout(timeout)
        s.connect(path)
        s.settimeout(None)
        if _peer_uid(s) != os.getuid():          # whoever is bound there is not us: never hand it argv / cwd / env
            raise OSError("resident: %s is served by another user" % path)
        return s
    except OSError:
        s.close()
        return None


def _recv_exact(s, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = s.recv(n - len(buf))
        if not chunk:
            raise EOFError("resident server closed the connection")
        buf += chunk
    return bytes(buf)


def client(argv, start_timeout=180.0):
    """Run `test.py argv` through the resident server (started if there is none).  Returns the run's exit status, or None
    when no server could be reached -- the caller then runs the frame loop itself."""
    try:
        path = socket_path(argv)
    except OSError as e:       # e.g. PermissionError: another user pre-created /tmp/t2v_resident_<uid> -- no server,
        print("resident: %s -- running in this process" % e, file=sys.stderr)      # not a crash: the caller runs the frame loop
        return 0 if "--resident_stop" in argv else None
    s = _connect(path, 1.0)
    if s is None and "--resident_stop" in argv:
        return 0
    if s is None:
        idle = "600"
        for i, a in enumerate(argv):
            if a == "--resident_idle_s" and i + 1 < len(argv):
                idle = argv[i + 1]
        import fcntl
        lock = os.open(path[:-5] + ".lock", os.O_CREAT | os.O_RDWR | os.O_NOFOLLOW, 0o600)
        try:
            fcntl.flock(lock, fcntl.LOCK_EX)       # two first calls at once: the second finds the first's server below
            s = _connect(path, 1.0)
            if s is None:
                log = os.open(path[:-5] + ".log", os.O_CREAT | os.O_WRONLY | os.O_APPEND | os.O_NOFOLLOW, 0o600)
                try:
                    subprocess.Popen([sys.executable, "-m", "text2video_amd.resident", path, idle], cwd=ROOT,
                                     stdin=subprocess.DEVNULL, stdout=log, stderr=log, start_new_session=True,
                                     env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))
                finally:
                    os.close(log)
                t_end = time.time() + start_timeout
                while s is None and time.time() < t_end:
                    time.sleep(0.05)
                    s = _connect(path, 1.0)
        finally:
            os.close(lock)                          # (closing drops the flock)
        if s is None:
            print("resident: no server came up at %s (see %s.log); running in this process" % (path, path[:-5]), file=sys.stderr)
            return None
    try:
        req = {"argv": list(argv), "cwd": os.getcwd(), "env": {k: v for k, v in os.environ.items() if k.startswith("T2V_")}}
        s.sendall(json.dumps(req).encode() + b"\n")
        while True:
            tag = _recv_exact(s, 1)
            if tag == b"x":
                rc = struct.unpack("<i", _recv_exact(s, 4))[0]
                # the torch-free server met a checkpoint only torch reads: this process runs the command itself (and
                # vid2vid/test.py's own LeanUnsupported handling starts it over with torch)
                return None if rc == RC_LEAN_UNSUPPORTED else rc
            data = _recv_exact(s, struct.unpack("<I", _recv_exact(s, 4))[0])
            out = sys.stdout if tag == b"o" else sys.stderr
            out.write(data.decode(errors="replace"))
            out.flush()
    except (EOFError, OSError) as e:
        print("resident: connection to the server lost (%s)" % e, file=sys.stderr)
        return 1
    finally:
        s.close()


class _Pipe:
    """file-like: what the frame loop prints goes to the client as it is produced"""

    def __init__(self, conn, tag):
        self.conn, self.tag = conn, tag

    def write(self, text):
        if text:
            data = text.encode()
            try:
                self.conn.sendall(self.tag + struct.pack("<I", len(data)) + data)
            except OSError:
                pass            # the client went away: the run itself goes on to its end
        return len(text)

    def flush(self):
        pass


def model_key(opt):
    """what decides whether a resident model can serve a request: the checkpoint files (path, size, mtime) or the synthetic
    seed, and the architecture flags"""
    files = []
    for s in range(opt.n_scales_spatial):
        p = os.path.abspath(os.path.join(opt.checkpoints_dir, opt.name, "%s_net_G%d.pth" % (opt.which_epoch, s)))
        try:
            st = os.stat(p)
            files.append((p, st.st_size, st.st_mtime_ns))
        except OSError:
            files.append((p, None, None))
    return json.dumps([files, opt.synthetic_weights, opt.ngf, opt.n_blocks, opt.n_blocks_local, opt.n_downsample_G,
                       opt.n_frames_G, opt.input_nc, opt.label_nc, opt.output_nc, opt.norm, bool(opt.no_flow),
                       bool(getattr(opt, "no_flow_explicit", False)), opt.n_scales_spatial, bool(opt.no_first_img), str(opt.gpu_ids)])


def serve(path, idle_s):
    import contextlib
    # the server's frame loop needs torch as little as the one-shot command's (text2video_amd/_xp.py): without it the call
    # that starts the server is ~1 s shorter and the resident process ~1 GB of host memory smaller (T2V_LEAN=0 keeps torch)
    if os.environ.get("T2V_LEAN", "1") != "0":
        from text2video_amd import _xp
        _xp.use_lean()
    from text2video_amd.model import LeanUnsupported, create_model, run_test
    from text2video_amd.options import TestOptions
    try:
        os.unlink(path)
    except OSError:
        pass
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    old = os.umask(0o177)
    try:
        srv.bind(path)
    finally:
        os.umask(old)
    srv.listen(8)
    print("resident: serving on %s (pid %d), idle limit %.0f s" % (path, os.getpid(), idle_s), flush=True)
    models = {}
    env0 = {k: v for k, v in os.environ.items() if k.startswith("T2V_")}
    try:
        while True:
            srv.settimeout(idle_s)
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                print("resident: idle for %.0f s, leaving" % idle_s, flush=True)
                return
            with conn:
                rc = 1
                try:
                    if _peer_uid(conn) != os.getuid():
                        continue
                    line = bytearray()
                    while not line.endswith(b"\n"):
                        chunk = conn.recv(65536)
                        if not chunk:
                            break
                        line += chunk
                    req = json.loads(line.decode())
                    argv = [a for a in req["argv"] if a != "--resident"]
                    if "--resident_stop" in argv:
                        conn.sendall(b"x" + struct.pack("<i", 0))
                        return
                    os.chdir(req["cwd"])
                    for k in [k for k in os.environ if k.startswith("T2V_")]:
                        del os.environ[k]
                    os.environ.update(env0)
                    os.environ.update(req.get("env", {}))
                    from text2video_amd import ops
                    ops.reload_env()                 # the library reads its switches once: this request's apply from here on
                    with contextlib.redirect_stdout(_Pipe(conn, b"o")), contextlib.redirect_stderr(_Pipe(conn, b"e")):
                        try:
                            opt = TestOptions().parse(argv)
                            device = "cuda:%d" % (opt.gpu_ids[0] if opt.gpu_ids else 0)
                            key = model_key(opt)
                            model = models.get(key)
                            if model is None:
                                models.clear()                      # one set of weights resident at a time
                                model = None
                                ops.torch.cuda.empty_cache()        # (either provider: the old weights' blocks go back to the driver)
                                model = models[key] = create_model(opt, device)
                            stats = run_test(opt, model=model, device=device)
                            print("done: %d frames, %.2f fps in the frame loop -> %s (resident server, pid %d)"
                                  % (stats["frames"], stats["fps_loop"], stats["results_dir"], os.getpid()))
                            rc = 0
                        except SystemExit as e:
                            rc = e.code if isinstance(e.code, int) else 1
                        except LeanUnsupported as e:
                            print("resident: %s -- the client runs this one itself" % e, file=sys.stderr)
                            models.clear()
                            rc = RC_LEAN_UNSUPPORTED
                        except BaseException:        # noqa: BLE001 -- reported to the client, the server lives on
                            import traceback
                            traceback.print_exc()
                            rc = 1
                finally:
                    try:
                        conn.sendall(b"x" + struct.pack("<i", int(rc)))
                    except OSError:
                        pass
    finally:
        srv.close()
        try:
            os.unlink(path)
        except OSError:
            pass


if __name__ == "__main__":
    serve(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 600.0)
