"""The self-launcher (text2video_amd/launch.py): a plain `python script --gpus N` / `--gpu_ids a,b` command line fans out
into one rank per device, as the reference's single command does through nn.DataParallel
(/root/reference/README.md:171-176; torch/nn/parallel/data_parallel.py:116-137).  CPU: the ranks talk over gloo."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    from text2video_amd import launch
    launch.fan_out_if_needed(int(sys.argv[1]), [int(v) for v in sys.argv[2].split(",")])
    # from here on: a rank (or a single-device run)
    import torch, torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if len(sys.argv) > 4 and rank == int(sys.argv[4]):
        sys.exit(7)                                  # a failing rank: the launcher must stop the others
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        total = float(t)
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1.0
    with open(os.path.join(sys.argv[3], "rank%%d.json" %% rank), "w") as fh:
        json.dump({"rank": rank, "world": world, "sum": total, "ids": os.environ.get("T2V_DEVICE_IDS"),
                   "addr": os.environ.get("MASTER_ADDR"), "local": os.environ.get("LOCAL_RANK")}, fh)
    if rank == 0:
        print(json.dumps({"n_gpus": world}))
""") % ROOT


def _run(tmp_path, n, ids, extra=(), env=None):
    script = tmp_path / "job.py"
    script.write_text(RANK_SCRIPT)
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    return subprocess.run([sys.executable, str(script), str(n), ids, str(tmp_path)] + list(extra), capture_output=True,
                          text=True, timeout=300, env=e)


def test_plain_command_fans_out_one_rank_per_listed_device(tmp_path):
    r = _run(tmp_path, 2, "3,5")
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)]
    assert [o["rank"] for o in outs] == [0, 1] and all(o["world"] == 2 and o["sum"] == 3.0 for o in outs)
    assert all(o["ids"] == "3,5" and o["addr"] == "127.0.0.1" for o in outs)
    assert [l for l in r.stdout.splitlines() if l.startswith("{")] == ['{"n_gpus": 2}']     # exactly one line, from rank 0


def test_single_device_does_not_spawn(tmp_path):
    r = _run(tmp_path, 1, "0")
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.load(open(tmp_path / "rank0.json"))
    assert o["world"] == 1 and o["ids"] is None and not (tmp_path / "rank1.json").exists()


def test_failing_rank_stops_the_job_with_its_status(tmp_path):
    r = _run(tmp_path, 2, "0,1", extra=["1"])
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with status 7" in r.stderr


def test_a_taken_rendezvous_port_is_retried_on_a_fresh_one(tmp_path):
    """free_port() only probes: a stranger may bind the port before rank 0 does.  The launcher sees the job fail early with the
    port held by somebody else and starts the ranks again on a fresh port; a failure with the port free is final (above)."""
    import socket
    held = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    held.bind(("127.0.0.1", 0))
    held.listen(1)
    try:
        env = dict(os.environ, T2V_LAUNCH_PORT=str(held.getsockname()[1]))
        r = _run(tmp_path, 2, "0,1", env=env)
    finally:
        held.close()
    assert r.returncode == 0, r.stderr[-2000:]
    assert "starting the ranks again on a fresh port" in r.stderr
    outs = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)]
    assert all(o["world"] == 2 and o["sum"] == 3.0 for o in outs)


def test_local_device_index_follows_gpu_ids(monkeypatch):
    from text2video_amd import launch
    monkeypatch.delenv("T2V_DIST_BACKEND", raising=False)
    monkeypatch.setenv("T2V_DEVICE_IDS", "4,6,7")
    assert [launch.local_device_index(r) for r in range(3)] == [4, 6, 7]
    monkeypatch.delenv("T2V_DEVICE_IDS")
    assert launch.local_device_index(5) == 5


def test_entry_points_call_the_launcher():
    """bench.py, vid2vid/train.py and vid2vid/test.py fan out themselves; under torchrun they must not (RANK set)."""
    from text2video_amd import launch
    for rel in ("bench.py", "vid2vid/train.py", "vid2vid/test.py"):
        assert "fan_out_if_needed" in open(os.path.join(ROOT, rel)).read(), rel
    os.environ.update(RANK="0", WORLD_SIZE="2")
    try:
        assert launch.under_launcher()
        launch.fan_out_if_needed(2)          # returns: already a rank
    finally:
        del os.environ["RANK"], os.environ["WORLD_SIZE"]
