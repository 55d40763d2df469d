"""The self-launcher (text2video_amd/launch.py): a plain `python script --gpus N` / `--gpu_ids a,b` command line fans out
into one rank per device, as the reference's single command does through nn.DataParallel
(/root/reference/README.md:171-176; torch/nn/parallel/data_parallel.py:116-137).  CPU: the ranks talk over gloo."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

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


# ---- a rank that goes missing: named, within the timeout, nothing left behind ---------------------------
WEDGE_SCRIPT = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    from text2video_amd import launch
    n, mode, bad = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    launch.fan_out_if_needed(n, list(range(n)))
    import torch
    import torch.distributed as dist
    from text2video_amd import distributed as D

    def main():
        rank = int(os.environ["RANK"])
        if mode == "never_starts" and rank == bad:
            os._exit(0)                    # gone before the process group exists
        rank, _, world = D.init_from_env("gloo")
        if mode == "exits_before_collective" and rank == bad:
            os._exit(0)                    # a clean exit status: the launcher alone would not notice
        if mode == "wedged" and rank == bad:
            time.sleep(600)                # alive but stuck: only the timeout can end this
        D.rendezvous("first collective")
        t = torch.ones(1)
        dist.all_reduce(t)
        print("rank %%d done %%g" %% (rank, float(t)), flush=True)
        dist.barrier()
        dist.destroy_process_group()       # (leaving with a live group aborts now and then at interpreter exit)

    D.fail_loudly(main)
""") % ROOT


def _wedge(tmp_path, mode, n=8, bad=5, timeout_s=6):
    import time
    script = tmp_path / "wedge.py"
    script.write_text(WEDGE_SCRIPT)
    e = dict(os.environ, T2V_DIST_TIMEOUT_S=str(timeout_s), T2V_CPU_AFFINITY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    t0 = time.time()
    p = subprocess.Popen([sys.executable, str(script), str(n), mode, str(bad)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, env=e)
    try:
        out, err = p.communicate(timeout=240)
    finally:
        if p.poll() is None:
            p.kill()
    return p.returncode, out, err, time.time() - t0


def _no_rank_left(marker):
    """no process whose command line carries this run's script path is alive"""
    import time
    psutil = pytest.importorskip("psutil")
    for _ in range(50):
        live = []
        for pr in psutil.process_iter(["cmdline", "status"]):
            try:
                if pr.info["status"] != psutil.STATUS_ZOMBIE and any(marker in a for a in (pr.info["cmdline"] or [])):
                    live.append(pr.pid)
            except (psutil.NoSuchProcess, psutil.AccessDenied):
                pass
        if not live:
            return True
        time.sleep(0.2)
    return False


@pytest.mark.parametrize("mode", ["exits_before_collective", "never_starts", "wedged"])
def test_eight_rank_job_names_the_missing_rank_and_leaves_nothing_behind(tmp_path, mode):
    """8 gloo ranks, rank 5 exits (status 0) before the first collective / never joins / hangs: the job ends inside the
    timeout (6 s here, 300 s by default -- not the backends' 10-30 minutes), with a non-zero status, rank 5 named on
    stderr, and no rank process left."""
    rc, out, err, dt = _wedge(tmp_path, mode)
    assert rc == 3, (rc, err[-3000:])
    assert "rank(s) [5] did not reach" in err, err[-3000:]
    assert "FAILED: RankFailure" in err
    assert "done" not in out                      # nobody ran the collective with a peer missing
    assert dt < 90, dt
    assert _no_rank_left(str(tmp_path / "wedge.py"))


def test_healthy_eight_rank_job_passes_the_roll_calls(tmp_path):
    rc, out, err, dt = _wedge(tmp_path, "none", bad=-1)
    assert rc == 0, err[-3000:]
    assert sorted(l for l in out.splitlines() if l.startswith("rank")) == sorted("rank %d done 8" % r for r in range(8))


# ---- CPU placement of a rank: the cores of its GPU's NUMA node, split between the ranks of that node ------------------------
def _fake_sysfs(root, nodes, pci_nodes, smt=True):
    """nodes: {node: [cpus]}; pci_nodes: {pci address: node}"""
    for node, cpus in nodes.items():
        d = root / "devices/system/node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    allc = sorted(c for cpus in nodes.values() for c in cpus)
    half = len(allc) // 2
    for c in allc:
        d = root / "devices/system/cpu" / ("cpu%d" % c) / "topology"
        d.mkdir(parents=True)
        sib = sorted({c, (c + half) % len(allc)}) if smt else [c]      # cpu c and c + N/2 share a core (the usual x86 numbering)
        (d / "thread_siblings_list").write_text(",".join(str(x) for x in sib) + "\n")
    for addr, node in pci_nodes.items():
        d = root / "bus/pci/devices" / addr
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)


def test_numa_pinning_splits_a_nodes_cores_between_its_ranks(tmp_path, monkeypatch):
    from text2video_amd import launch
    # 2 sockets x 8 cores x 2 threads: node 0 = cpus 0-7 + 16-23, node 1 = 8-15 + 24-31; GPUs 0-3 on node 0, 4-7 on node 1
    nodes = {0: list(range(0, 8)) + list(range(16, 24)), 1: list(range(8, 16)) + list(range(24, 32))}
    pci = {"0000:%02x:00.0" % (0x10 + g): (0 if g < 4 else 1) for g in range(8)}
    _fake_sysfs(tmp_path, nodes, pci)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(32)), raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.delenv("T2V_DEVICE_IDS", raising=False)
    monkeypatch.delenv("T2V_DIST_BACKEND", raising=False)
    monkeypatch.delenv("T2V_CPU_AFFINITY", raising=False)
    shares = []
    for r in range(8):
        monkeypatch.setenv("LOCAL_RANK", str(r))
        shares.append(launch.pin_to_numa_node(r, sysfs=str(tmp_path), pci_of=lambda i: "0000:%02x:00.0" % (0x10 + i), apply=False))
    assert shares[0] == [0, 1, 16, 17] and shares[3] == [6, 7, 22, 23]          # whole cores (both SMT threads), contiguous
    assert shares[4] == [8, 9, 24, 25] and shares[7] == [14, 15, 30, 31]
    assert sorted(c for s in shares for c in s) == list(range(32))              # disjoint, nothing idle
    # the mask the job was started with is respected (a cgroup / taskset that leaves node 0 only half its cpus)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(0, 4)) | set(range(16, 20)) | set(nodes[1]), raising=False)
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert launch.pin_to_numa_node(1, sysfs=str(tmp_path), pci_of=lambda i: "0000:%02x:00.0" % (0x10 + i), apply=False) == [1, 17]
    # no topology, node -1, switched off: a no-op
    assert launch.pin_to_numa_node(0, sysfs=str(tmp_path / "nowhere"), pci_of=lambda i: "0000:10:00.0", apply=False) is None
    (tmp_path / "bus/pci/devices/0000:10:00.0/numa_node").write_text("-1\n")
    assert launch.pin_to_numa_node(0, sysfs=str(tmp_path), pci_of=lambda i: "0000:10:00.0", apply=False) is None
    monkeypatch.setenv("T2V_CPU_AFFINITY", "0")
    assert launch.pin_to_numa_node(1, sysfs=str(tmp_path), pci_of=lambda i: "0000:11:00.0", apply=False) is None
    assert launch.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_bench_refuses_more_ranks_than_devices():
    """`bench.py --gpus 8` on a node that shows fewer devices ends at once with a message (RCCL would sit in its communicator
    set-up until the timeout); here: no device at all."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "T2V_DIST_BACKEND"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 2 and "--gpus 8 needs 8 visible GPUs" in r.stderr, (r.returncode, r.stderr[-1500:])
