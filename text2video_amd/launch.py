"""One command -> one rank per GPU.

The reference fans out inside one process from one command line: `python train.py ... --gpu_ids 0,1,2,3,4,5,6,7
--batchSize 8` (/root/reference/README.md:171-176) wraps the model in nn.DataParallel, which scatters the batch,
replicates the module and runs one Python thread per listed device
(/root/reference/venv_vid2vid/lib/python3.7/site-packages/torch/nn/parallel/data_parallel.py:116-137).  Here the
fan-out is one PROCESS per device with persistent replicas: when an entry point (vid2vid/train.py, vid2vid/test.py,
bench.py) is started plainly -- no RANK / WORLD_SIZE in the environment -- and asks for more than one device, it
re-executes its own command line once per device with the torchrun-style environment (RANK, LOCAL_RANK,
WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT=<free port>) and waits.  Under
`python -m torch.distributed.run` the environment is already there and nothing is spawned.

T2V_DEVICE_IDS carries the `--gpu_ids` list to the ranks: rank r computes on device ids[r] (as DataParallel's
device_ids[r]).  With T2V_DIST_BACKEND=gloo several ranks may share a device (single-GPU tests), ids wrap around.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def under_launcher():
    """True inside a rank that torchrun (or self_launch) started."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def local_device_index(local_rank):
    """Device ordinal this rank computes on: --gpu_ids[local_rank] when the launcher passed the list on, else
    local_rank; wrapped to the visible devices when ranks share GPUs over gloo (tests)."""
    ids = [int(v) for v in os.environ.get("T2V_DEVICE_IDS", "").split(",") if v.strip() != ""]
    idx = ids[local_rank % len(ids)] if ids else local_rank
    if os.environ.get("T2V_DIST_BACKEND") == "gloo":
        import torch
        if torch.cuda.is_available():
            idx %= torch.cuda.device_count()
    return idx


_PINNED = []


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)"""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def device_pci_address(index):
    """'dddd:bb:dd.f' of HIP device `index`, or None (no GPU / an older torch without the fields)"""
    try:
        import torch
        if not torch.cuda.is_available() or index >= torch.cuda.device_count():
            return None
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:      # noqa: BLE001
        return None


def numa_node_of(pci_address, sysfs="/sys"):
    """NUMA node of a PCI function (the file /sys/class/drm/card*/device/numa_node resolves to), or None when the file
    is missing or says -1 (one node / a VM without topology)"""
    try:
        node = int(open(os.path.join(sysfs, "bus/pci/devices", pci_address, "numa_node")).read().strip())
    except (OSError, ValueError, TypeError):
        return None
    return node if node >= 0 else None


def numa_cpu_share(node, peers, my_index, sysfs="/sys", allowed=None):
    """The CPUs of NUMA node `node` this rank takes when `peers` ranks have their GPU on that node and this one is number
    `my_index` among them: whole physical cores (a core's SMT siblings stay together), contiguous blocks, every rank the
    same count; restricted to `allowed` (the mask the job was started with).  None when the topology cannot be read."""
    try:
        cpus = parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read())
    except (OSError, ValueError):
        return None
    if allowed is not None:
        cpus = [c for c in cpus if c in allowed]
    if not cpus:
        return None
    cores, seen = [], set()
    for c in cpus:
        if c in seen:
            continue
        try:
            sib = parse_cpulist(open(os.path.join(sysfs, "devices/system/cpu/cpu%d/topology/thread_siblings_list" % c)).read())
        except (OSError, ValueError):
            sib = [c]
        grp = [t for t in sib if t in cpus and t not in seen] or [c]
        seen.update(grp)
        cores.append(sorted(grp))
    peers = max(1, peers)
    per = len(cores) // peers
    if per == 0:          # more ranks than cores on the node: share the node
        return sorted(c for g in cores for c in g)
    mine = cores[my_index * per:(my_index + 1) * per]
    return sorted(c for g in mine for c in g)


def pin_to_numa_node(device_index, sysfs="/sys", pci_of=device_pci_address, apply=True):
    """Restrict this rank (and everything it forks later: the pose-rasteriser workers, the JPEG threads) to its share of
    the CPUs of the NUMA node its GPU hangs off -- 8 ranks x (1 launch thread + 4 workers) otherwise migrate across both
    sockets of the host and reach their pinned staging buffers through the inter-socket link.  The ranks of the node
    split its cores evenly (numa_cpu_share).  A no-op (returns None) when T2V_CPU_AFFINITY=0, off Linux, without a GPU,
    or when sysfs does not say where the device is.  Returns the CPU list applied."""
    if os.environ.get("T2V_CPU_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    if apply and _PINNED:          # once per process (a second split would start from the first one's share)
        return _PINNED[0]
    try:
        node = numa_node_of(pci_of(device_index), sysfs)
        if node is None:
            return None
        nlocal = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        me = int(os.environ.get("LOCAL_RANK", "0"))
        on_node = [lr for lr in range(nlocal)
                   if numa_node_of(pci_of(local_device_index(lr)), sysfs) == node]
        if me not in on_node:
            on_node = sorted(set(on_node + [me]))
        allowed = set(os.sched_getaffinity(0))
        share = numa_cpu_share(node, len(on_node), on_node.index(me), sysfs, allowed)
        if not share:
            return None
        if apply:
            os.sched_setaffinity(0, share)
            _PINNED.append(share)
        return share
    except Exception:      # noqa: BLE001 -- placement is an optimisation: never the reason a rank does not start
        return None


def _die_with_parent():
    """child side (preexec): SIGTERM when the launching process dies -- a SIGKILLed parent cannot stop its ranks itself, and
    ranks blocked in a collective would otherwise stay behind.  prctl(PR_SET_PDEATHSIG = 1, SIGTERM)."""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)
    except Exception:      # noqa: BLE001 -- not Linux / no libc: the ranks simply keep the old behaviour
        pass


def _port_taken(port):
    """True when somebody LISTENS on 127.0.0.1:port now (the ranks of a failed attempt are gone by the time this is asked).
    A connect, not a bind: the port of a finished job stays un-bindable for a minute (its connections sit in TIME_WAIT)
    although nobody holds it."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.settimeout(1.0)
        try:
            s.connect(("127.0.0.1", port))
            return True
        except OSError:
            return False


def self_launch(nranks, device_ids=None, argv=None, poll_s=0.05):
    """Run `sys.executable argv` (default: this process's own command line) as `nranks` ranks and return the job's
    exit status: 0 when every rank returned 0, else the first failing rank's status (the remaining ranks are
    terminated by PID -- a rank blocked in a collective whose peer died would otherwise wait for its timeout).
    Ranks inherit stdout / stderr, so rank 0's single JSON / summary line is the job's.
    free_port() can only probe: between its close() and rank 0's bind another process may take the port.  A job
    that fails within its first minute while a stranger holds its rendezvous port is started again on a fresh one (twice at
    most); any other failure is final.  T2V_LAUNCH_PORT pins the first attempt's port (tests)."""
    argv = list(sys.argv if argv is None else argv)
    pinned = os.environ.get("T2V_LAUNCH_PORT")
    rc = 1
    for attempt in range(3):
        port = int(pinned) if (pinned and attempt == 0) else free_port()
        t0 = time.time()
        rc = _launch_once(nranks, device_ids, argv, port, poll_s)
        if rc == 0 or time.time() - t0 > 60.0 or not _port_taken(port):
            return rc
        print("launch: rendezvous port %d was taken by another process -- starting the ranks again on a fresh port"
              % port, file=sys.stderr, flush=True)
    return rc


def _launch_once(nranks, device_ids, argv, port, poll_s):
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), T2V_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
        if device_ids:
            env["T2V_DEVICE_IDS"] = ",".join(str(d) for d in device_ids)
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, preexec_fn=_die_with_parent))

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    old = {}
    for s in (signal.SIGINT, signal.SIGTERM):
        try:
            old[s] = signal.signal(s, lambda signum, frame: (stop_all(signum), sys.exit(128 + signum)))
        except ValueError:      # not the main thread
            pass
    rc = 0
    try:
        live = set(range(nranks))
        while live:
            for r in sorted(live):
                st = procs[r].poll()
                if st is None:
                    continue
                live.discard(r)
                if st != 0 and rc == 0:
                    rc = st if st > 0 else 128 - st
                    print("launch: rank %d exited with status %d -- stopping the other ranks" % (r, st), file=sys.stderr,
                          flush=True)
                    stop_all()
                    t_kill = time.time() + 10.0
                    while any(p.poll() is None for p in procs) and time.time() < t_kill:
                        time.sleep(poll_s)
                    stop_all(signal.SIGKILL)
            if live:
                time.sleep(poll_s)
    finally:
        for s, h in old.items():
            signal.signal(s, h)
    return rc


def fan_out_if_needed(nranks, device_ids=None):
    """Entry-point helper: if this process was started plainly and wants `nranks` > 1, run the ranks and exit with
    the job's status; otherwise return (the caller is a rank, or a single-device run)."""
    if nranks > 1 and not under_launcher():
        sys.exit(self_launch(nranks, device_ids))
