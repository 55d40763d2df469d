"""GPU tests of the multi-GPU code paths that one GPU can execute: the RCCL collectives themselves (a 1-rank `nccl`
process group runs the same RCCL entry points on device tensors: all_gather_into_tensor of the uint8 frames, the
bucketed gradient all-reduce, the stitch pass's tail exchange), and BASELINE configs[2]'s chunk plan with the REAL
HIP generator run over it, chunk by chunk, against the CPU oracle on the same chunks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_its_rccl_path_on_one_rank():
    """bench.py with T2V_BENCH_FORCE_DIST=1: process group "nccl" (= RCCL) of one rank, barrier, the in-region
    all_gather_into_tensor of the chunk's uint8 frames, MAX all-reduce of the elapsed time."""
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", T2V_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--cpu-frames", "0", "--kernel-iters", "2", "--e2e-frames", "0", "--single-variant",
                        "--hires-frames", "0", "--train-steps", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 30.0 and d["config"]["parallelism"] == "sequence-chunk dp1"
    assert d["config"]["collectives"] == "rccl, 1 ranks" and d["config"]["world_size"] == 1
    # the configs[4] block: a whole train step at 512x512 with the bucketed exchange on the 1-rank RCCL group, timed with and
    # without the collectives; all of G's and D's gradient bytes went through them
    t = d["train_step"]
    assert t["steps"] == 2 and 20.0 < t["ms_per_step"] < 2000.0 and t["exchange"]["group"] == "1-rank rccl"
    assert t["exchange"]["bytes"] > 1.3e9 and t["exchange"]["buckets"] >= 20
    assert abs(t["exchange"]["ms"] - (t["exchange"]["ms_per_step_with"] - t["exchange"]["ms_per_step_without"])) < 0.02
    assert len(t["kernels"]) == 3 and all(0.05 < k["frac"] < 1.0 for k in t["kernels"])


def _rccl_worker(tmp):
    """runs in its own process: 1-rank RCCL group; GradientExchange buckets on device gradients, gather_frames of
    uint8 frames, exchange_tails, and a whole Vid2VidTrainer step with the exchange forced on."""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534",
                      T2V_TRAIN_FORCE_DIST="1", T2V_TRAIN_COMM_TIMING="1")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from text2video_amd import distributed as D
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    out = {"backend": dist.get_backend()}
    # frames all-gather (uint8, device)
    fr = torch.randint(0, 255, (5, 32, 32, 4), dtype=torch.uint8, device="cuda:0")
    full = torch.empty(5, 32, 32, 4, dtype=torch.uint8, device="cuda:0")
    dist.all_gather_into_tensor(full, fr)
    out["gather_equal"] = bool(torch.equal(full, fr))
    # bucketed gradient exchange on device tensors, two exchanges in flight
    params = [torch.nn.Parameter(torch.zeros(n, device="cuda:0")) for n in (5, 70000, 3, 1 << 18, 17)]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(i + 1))
    xa = T.allreduce_gradients_begin(params[:3], bucket_mb=1)
    xb = T.allreduce_gradients_begin(params[3:], bucket_mb=1)
    out["bytes"] = xa.finish() + xb.finish()
    out["buckets"] = True
    out["grads_ok"] = all(bool(torch.equal(p.grad, torch.full_like(p, float(i + 1)))) for i, p in enumerate(params))
    # stitch pass's tail exchange
    plan = [[("seq", 0, 10, 2), ("seq", 8, 18, 10)]]
    tails = [torch.randn(16, 16, 8, device="cuda:0") for _ in range(2)]
    known = D.exchange_tails(plan, 0, tails)
    out["tails_ok"] = bool(torch.equal(known[("seq", 10)], tails[0]) and torch.equal(known[("seq", 18)], tails[1]))
    # a whole trainer step with the (1-rank) RCCL gradient exchange in it
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "2", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img"])
    tr = T.Vid2VidTrainer(opt, "cuda:0", seed=5)
    g = torch.Generator().manual_seed(0)
    pose = torch.zeros(2, 64, 64, 12, device="cuda:0")
    pose[..., :9] = (torch.rand(2, 64, 64, 9, generator=g) * 2 - 1).cuda()
    real = torch.zeros(2, 64, 64, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.randn(2, 64, 64, 3, generator=g)).cuda()
    losses, _ = tr.train_step(pose, real, None, None, real_prev=real.flip(0).contiguous())
    out["comm_bytes"] = tr.comm_bytes
    out["comm_ms_exposed"] = tr.comm_ms         # what of the exchange was still running when the backward kernels had drained
    out["buckets"] = [len(tr.bucketsG.bounds), len(tr.bucketsD.bounds)]
    out["n_param_bytes"] = 4 * sum(p.numel() for p in tr.optG.params + tr.optD.params if p.grad is not None)
    out["losses_finite"] = all(np.isfinite(v) for v in losses.values())
    # the same step with the reduce-scatter + all-gather form of the exchange (in place on the flat buckets)
    os.environ["T2V_GRAD_RS_AG"] = "1"
    tr2 = T.Vid2VidTrainer(opt, "cuda:0", seed=5)
    losses2, _ = tr2.train_step(pose, real, None, None, real_prev=real.flip(0).contiguous())
    os.environ["T2V_GRAD_RS_AG"] = "0"
    out["rs_ag_bytes"] = tr2.comm_bytes
    out["rs_ag_same_losses"] = all(losses[k] == losses2[k] for k in losses)
    out["rs_ag_same_weights"] = all(bool(torch.equal(a, b)) for a, b in zip(tr.optG.params, tr2.optG.params))
    # ... and with every bucket's collective issued from the current stream after joining the weight-gradient stream (the form
    # until round 6; default now: issued FROM the weight-gradient stream, the backward pass does not stop)
    os.environ["T2V_EXCHANGE_FROM_SIDE"] = "0"
    tr3 = T.Vid2VidTrainer(opt, "cuda:0", seed=5)
    losses3, _ = tr3.train_step(pose, real, None, None, real_prev=real.flip(0).contiguous())
    del os.environ["T2V_EXCHANGE_FROM_SIDE"]
    out["join_same_losses"] = all(losses[k] == losses3[k] for k in losses) and tr3.comm_bytes == tr.comm_bytes
    out["join_same_weights"] = all(bool(torch.equal(a, b)) for a, b in zip(tr.optG.params + tr.optD.params,
                                                                            tr3.optG.params + tr3.optD.params))
    dist.barrier()
    dist.destroy_process_group()
    with open(tmp, "w") as fh:
        json.dump(out, fh)


def test_rccl_collectives_execute_on_device_tensors(tmp_path):
    """GradientExchange / frame all-gather / tail exchange through RCCL (1-rank "nccl" group) on device tensors."""
    out = str(tmp_path / "o.json")
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from tests.test_gpu_distributed import "
                        "_rccl_worker; _rccl_worker(%r)" % (ROOT, out)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.load(open(out))
    assert d["backend"] == "nccl" and d["gather_equal"] and d["grads_ok"] and d["tails_ok"] and d["losses_finite"]
    assert d["bytes"] == 4 * (5 + 70000 + 3 + (1 << 18) + 17)
    assert d["comm_bytes"] == d["n_param_bytes"] > 0         # every gradient of the step went through the exchange
    assert d["rs_ag_bytes"] == d["comm_bytes"] and d["rs_ag_same_losses"] and d["rs_ag_same_weights"]
    assert d["join_same_losses"] and d["join_same_weights"]
    print("1-rank RCCL gradient exchange: %.1f MB in %s buckets, %.3f ms exposed after the backward pass"
          % (d["comm_bytes"] / 2**20, d["buckets"], d["comm_ms_exposed"]))
    assert 0.0 <= d["comm_ms_exposed"] < 1000.0


def test_config2_chunk_plan_with_the_real_generator_matches_oracle_per_chunk():
    """BASELINE configs[2]: a 514-pose-map sequence (512 output frames) cut by the 8-rank plan into 8 chunks of 64
    frames.  The chunks are run serially on this GPU through the real HIP generator (reduced size: 64x64, ngf 16),
    each as the plan prescribes -- a fresh recurrence, zero previous frames and raw-only first frame -- and compared
    frame by frame with the oracle run on the same chunk (teacher-forced: every frame starts from the oracle's
    FIFO, so differences do not compound through the random-init recurrence).  The oracle runs on the GPU for all eight
    chunks (its modules on cuda, ATen's exact-fp32 kernels) and on the CPU for the first one, where HIP, device oracle and CPU
    oracle must agree pairwise -- 512 frames of the CPU oracle alone were a minute of the suite.  Also: the chunks tile frames 2..513
    exactly once, and a chunk's first frames DIFFER from the unsharded run's (the seam that --stitch_frames closes)."""
    import copy
    from oracle.generator_ref import CompositeGenerator, Vid2VidInferenceRef
    from text2video_amd.distributed import plan_units
    from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict
    spec = GeneratorSpec(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch")
    sd = synthetic_state_dict(spec, 4, "vid2vid", flow_gain=0.1)
    net = CompositeGenerator(9, 3, 6, 16, 2, 2, False, "batch")
    net.load_state_dict(sd, strict=False)
    ref_cpu = Vid2VidInferenceRef([net])
    ref = Vid2VidInferenceRef([copy.deepcopy(net).to("cuda:0")])
    hip = Vid2VidModelG([HipGenerator(spec, "cuda:0").load_state_dict(sd)])
    H = W = 64
    rng = np.random.default_rng(0)
    poses = torch.from_numpy(np.where(rng.random((514, 1, H, W)) < 0.03, rng.uniform(-1, 1, (514, 3, H, W)), -1.0)
                             .astype(np.float32))
    poses_d = poses.to("cuda:0")
    plan = plan_units({"seq": 514}, 8, 3, shard_chunks=True)
    assert all(len(p) == 1 for p in plan)
    covered, worst, worst_cpu, worst_hip_cpu = [], 0.0, 0.0, 0.0
    seam = None
    old = (torch.backends.cudnn.enabled, torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32)
    torch.backends.cudnn.enabled = torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    try:
        for r in range(8):
            (seq, s, e, first_out), = plan[r]
            assert e - first_out == 64 and first_out == s + 2
            ref.reset()
            ref_cpu.reset()
            hip.reset()
            for t in range(first_out, e):
                A = poses_d[t - 2:t + 1].unsqueeze(0)
                if ref.fake_B_prev is not None:
                    hip.load_prev(ref.fake_B_prev)
                    ref_cpu.fake_B_prev = [p.cpu() for p in ref.fake_B_prev]
                want = ref.inference(A)
                got, _ = hip.inference(A)
                worst = max(worst, (got - want).abs().max().item())
                if r == 0:
                    want_cpu = ref_cpu.inference(poses[t - 2:t + 1].unsqueeze(0))
                    worst_cpu = max(worst_cpu, (want_cpu - want.cpu()).abs().max().item())
                    worst_hip_cpu = max(worst_hip_cpu, (got.cpu() - want_cpu).abs().max().item())
                covered.append(t)
                if r == 1 and t == first_out:
                    seam = want.clone()
        assert covered == list(range(2, 514))
        print("configs[2] chunk plan, 8 x 64 frames at 64x64: max|delta| vs oracle per chunk = %.3g (chunk 0: device oracle vs CPU "
              "oracle %.3g, HIP vs CPU oracle %.3g)" % (worst, worst_cpu, worst_hip_cpu))
        # (two fp32 evaluations of the oracle differ by what the HIP path differs from either: measured 3e-5 / 9e-6 / 4e-5)
        assert worst <= 2e-4 and worst_cpu <= 2e-4 and worst_hip_cpu <= 2e-4
        # the unsharded oracle run reaches chunk 1's first frame with a non-zero FIFO: the seam is real
        ref.reset()
        first_out = plan[1][0][3]
        for t in range(2, first_out + 1):
            full = ref.inference(poses_d[t - 2:t + 1].unsqueeze(0))
        assert (full - seam).abs().max().item() > 1e-2
    finally:
        torch.backends.cudnn.enabled, torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = old


def _run_test_py(work, extra, env, nproc=1, port=29541):
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "vid2vid", "test.py"), "--name", "fadg0", "--dataroot", "datasets/fadg0", "--dataset_mode", "pose",
            "--input_nc", "3", "--resize_or_crop", "scaleHeight", "--loadSize", "128", "--openpose_only", "--how_many", "1200",
            "--no_first_img", "--random_drop_prob", "0", "--synthetic_weights", "1", "--ngf", "16", "--n_blocks", "2",
            "--n_downsample_G", "2", "--pose_workers", "2"] + extra
    r = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r


def test_two_rank_frame_loop_whole_sequences_chunks_and_stitch(tmp_path):
    """The drop-in frame loop (vid2vid/test.py) under torchrun with TWO ranks, both on this GPU (gloo transport,
    T2V_DIST_BACKEND=gloo: RCCL refuses two ranks per device), against the single-process run:
      * default sharding (whole sequences): identical JPEG files, every frame written exactly once;
      * --shard_chunks: the one long sequence is cut in two -- same files, but frames after the cut differ;
      * --shard_chunks --stitch_frames <chunk length>: the tail exchange + re-generation reproduces the single-process
        files bit for bit."""
    import glob
    import shutil
    from PIL import Image
    from text2video_amd.keypoints import read_keypoints
    src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", T2V_DIST_BACKEND="gloo")

    def dataset(work, seqs):
        root = os.path.join(work, "datasets", "fadg0")
        for seq, n in seqs.items():
            os.makedirs(os.path.join(root, "test_openpose", seq))
            os.makedirs(os.path.join(root, "test_img", seq))
            img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), (128, 96)))
            for i in range(n):
                shutil.copyfile(os.path.join(src, files[(i * 5 + len(seq)) % len(files)]),
                                os.path.join(root, "test_openpose", seq, "%05d.json" % i))
                img.save(os.path.join(root, "test_img", seq, "%04d.jpg" % i))

    def frames(work):
        out = {}
        for p in sorted(glob.glob(os.path.join(work, "results", "fadg0", "test_latest", "*", "fake_B_*.jpg"))):
            out[os.path.relpath(p, work)] = open(p, "rb").read()
        return out

    # (1) two sequences, whole-sequence sharding
    w1, w2 = str(tmp_path / "one"), str(tmp_path / "two")
    for w in (w1, w2):
        os.makedirs(w)
        dataset(w, {"tmp": 9, "tmp_smooth": 7})
    _run_test_py(w1, [], env)
    _run_test_py(w2, [], env, nproc=2)
    a, b = frames(w1), frames(w2)
    assert len(a) == (9 - 2) + (7 - 2) and a.keys() == b.keys() and all(a[k] == b[k] for k in a)
    # (2) one sequence of 14 pose maps -> 12 frames, cut in two chunks of 6
    ws = {k: str(tmp_path / k) for k in ("single", "chunks", "stitched")}
    for w in ws.values():
        os.makedirs(w)
        dataset(w, {"tmp": 14})
    _run_test_py(ws["single"], [], env)
    _run_test_py(ws["chunks"], ["--shard_chunks"], env, nproc=2, port=29542)
    r = _run_test_py(ws["stitched"], ["--shard_chunks", "--stitch_frames", "100"], env, nproc=2, port=29543)
    one, ch, st = frames(ws["single"]), frames(ws["chunks"]), frames(ws["stitched"])
    assert len(one) == 12 and one.keys() == ch.keys() == st.keys()
    names = sorted(one)
    assert all(one[k] == ch[k] for k in names[:6])            # the first chunk is the sequence's start
    assert any(one[k] != ch[k] for k in names[6:])            # the second restarts the recurrence: a seam
    assert all(one[k] == st[k] for k in names), [k for k in names if one[k] != st[k]]


def test_two_rank_train_py_keeps_replicas_in_sync(tmp_path):
    """vid2vid/train.py under torchrun with two ranks sharing this GPU (gloo transport): every rank trains on its own
    synthetic clip, the bucketed gradient exchange averages G's and the discriminators' gradients, and after the run the
    replicas' weights are still identical (checked inside run_train, which raises otherwise)."""
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", T2V_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(ROOT, "vid2vid", "train.py"), "--name", "dp", "--dataset_mode", "pose",
           "--input_nc", "3", "--openpose_only", "--no_first_img", "--ngf", "16", "--n_blocks", "2", "--n_downsample_G", "2",
           "--num_D", "2", "--ndf", "16", "--fineSize", "64", "--batchSize", "2", "--max_frames_per_gpu", "2", "--niter", "3",
           "--niter_decay", "0", "--add_face_disc", "--no_vgg", "--synthetic_data", "--checkpoints_dir", str(tmp_path / "ck")]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "replicas in sync after 3 steps" in r.stdout and "on all 2 ranks" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("(iter")][-1]
    mb = float(line.split("all-reduce")[1].split("MB")[0])
    assert mb > 1.0, line                                        # the gradients of G and the discriminators went through it
    assert os.path.exists(tmp_path / "ck" / "dp" / "latest_net_G0.pth")


def test_bench_two_ranks_on_one_gpu_contract():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per "GPU"), here with both ranks on this
    GPU over gloo: ONE JSON line from rank 0, n_gpus = 2, value = the frames of BOTH ranks over the slowest rank's time,
    weak scaling, no cpu_baseline / e2e legs (those are N = 1 only)."""
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", T2V_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--kernel-iters", "2", "--single-variant", "--train-steps", "1", "--train-ngf", "32"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["cpu_baseline"] is None and d["e2e"] is None
    assert "hires" not in d                                          # (N = 1 only)
    # configs[4] with N = 2: every rank its own clip, the real gradient all-reduce between them, replicas still equal
    t = d["train_step"]
    assert t["exchange"]["group"] == "2-rank gloo" and t["exchange"]["replicas_in_sync"] is True
    assert t["exchange"]["bytes"] > 5e7 and "ngf 32" in t["workload"]      # (a narrow generator here: the gloo exchange of the
                                                                           # full 1.5 GB through the host takes a minute)
    assert d["config"]["parallelism"] == "sequence-chunk dp2" and d["config"]["collectives"] == "gloo, 2 ranks"
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]      # 2 ranks x K frames / max-over-ranks time
    assert d["value"] > 30.0


def _plain_env():
    """no torchrun variables: the entry points have to fan out themselves"""
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", T2V_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_two_ranks_agree_on_the_face_terms_when_one_clip_shows_no_face(tmp_path):
    """--add_face_disc with two ranks whose clips differ: clipA shows the nose-neck limb (a face region exists), clipB has
    the nose key point at confidence 0 (no face region -> that rank alone would skip D_f's terms, leave D_f without
    gradients and put a different parameter list into its gradient buckets).  The ranks settle it with a MIN all-reduce of
    the has-face flag per chunk: the face terms are skipped on both, the exchange stays aligned, replicas stay in sync."""
    import json as js
    import shutil
    from PIL import Image
    from text2video_amd.keypoints import read_keypoints
    root = tmp_path / "datasets" / "fadg0"
    src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
    for seq in ("clipA", "clipB"):
        os.makedirs(root / "train_openpose" / seq)
        os.makedirs(root / "train_img" / seq)
        for i, f in enumerate(files + files[::-1]):
            d = js.load(open(os.path.join(src, f)))
            if seq == "clipB":
                d["people"][0]["pose_keypoints_2d"][2] = 0.0          # nose confidence 0: the nose-neck limb is not drawn
            dst = root / "train_openpose" / seq / ("%04d_keypoints.json" % i)
            js.dump(d, open(dst, "w"))
            Image.fromarray(read_keypoints(str(dst), (256, 192))).save(root / "train_img" / seq / ("%04d.jpg" % i))
    cmd = [sys.executable, os.path.join(ROOT, "vid2vid", "train.py"), "--name", "fadg0", "--dataroot", "datasets/fadg0",
           "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2", "--resize_or_crop",
           "randomScaleHeight_and_scaledCrop", "--loadSize", "136", "--fineSize", "128", "--gpu_ids", "0,1", "--batchSize", "2",
           "--max_frames_per_gpu", "2", "--no_first_img", "--n_frames_total", "5", "--max_t_step", "2", "--niter_step", "100",
           "--add_face_disc", "--random_drop_prob", "0", "--ngf", "16", "--n_blocks", "2", "--n_downsample_G", "2",
           "--ndf", "16", "--no_vgg", "--niter", "2", "--niter_decay", "0", "--checkpoints_dir", str(tmp_path / "ck")]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=_plain_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "replicas in sync after 2 steps" in r.stdout and "on all 2 ranks" in r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("(iter")]
    assert lines and not any("D_f" in l for l in lines), lines      # rank 0's clip HAS a face: skipped together all the same


def test_chunk_plan_on_one_gpu_equals_the_two_rank_chunk_plan(tmp_path):
    """BASELINE configs[2] on fewer GPUs than chunks: `test.py --shard_chunks --chunks_per_rank 2` in ONE process cuts the
    sequence exactly as two ranks would and advances the two chunks in lock-step (batch 2) -- the JPEG files equal the
    two-rank `--shard_chunks` run's; with the stitch pass over the whole chunk length they equal the unsharded run's."""
    import glob
    import shutil
    from PIL import Image
    from text2video_amd.keypoints import read_keypoints
    src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
    env = _plain_env()

    def work(name):
        w = str(tmp_path / name)
        root = os.path.join(w, "datasets", "fadg0")
        os.makedirs(os.path.join(root, "test_openpose", "tmp"))
        os.makedirs(os.path.join(root, "test_img", "tmp"))
        img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), (128, 96)))
        for i in range(14):
            shutil.copyfile(os.path.join(src, files[(i * 5 + 3) % len(files)]), os.path.join(root, "test_openpose", "tmp", "%05d.json" % i))
            img.save(os.path.join(root, "test_img", "tmp", "%04d.jpg" % i))
        return w

    def frames(w):
        return {os.path.relpath(p, w): open(p, "rb").read()
                for p in sorted(glob.glob(os.path.join(w, "results", "fadg0", "test_latest", "*", "fake_B_*.jpg")))}

    ws = {k: work(k) for k in ("single", "two_ranks", "one_rank_two_chunks", "one_rank_stitched")}
    _run_test_py(ws["single"], [], env)
    _run_test_py(ws["two_ranks"], ["--shard_chunks", "--gpu_ids", "0,1"], env)
    _run_test_py(ws["one_rank_two_chunks"], ["--shard_chunks", "--chunks_per_rank", "2"], env)
    _run_test_py(ws["one_rank_stitched"], ["--shard_chunks", "--chunks_per_rank", "2", "--stitch_frames", "100"], env)
    one, two, cpr, st = (frames(ws[k]) for k in ("single", "two_ranks", "one_rank_two_chunks", "one_rank_stitched"))
    assert len(one) == 12 and one.keys() == two.keys() == cpr.keys() == st.keys()
    assert all(two[k] == cpr[k] for k in two)                       # the same chunks, the same frames
    assert any(one[k] != cpr[k] for k in one)                       # ... which restart the recurrence at the cut
    assert all(one[k] == st[k] for k in one)                        # stitched over the whole chunk: the unsharded sequence


def test_stitch_pass_with_a_two_scale_generator_reproduces_the_unsharded_run(tmp_path):
    """configs[2] x configs[3]: `test.py --n_scales_spatial 2 --shard_chunks --chunks_per_rank 2 --stitch_frames
    <chunk length>` -- the generator keeps one FIFO per pyramid level, the stitch pass's tail carries both
    (distributed.pack_state) -- writes the files of the unsharded two-scale run, byte for byte; without the stitch pass the
    frames after the cut differ."""
    import glob
    import shutil
    from PIL import Image
    from text2video_amd.keypoints import read_keypoints
    src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
    env = _plain_env()

    def work(name):
        w = str(tmp_path / name)
        root = os.path.join(w, "datasets", "fadg0")
        os.makedirs(os.path.join(root, "test_openpose", "tmp"))
        os.makedirs(os.path.join(root, "test_img", "tmp"))
        img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), (128, 96)))
        for i in range(12):
            shutil.copyfile(os.path.join(src, files[(i * 5 + 3) % len(files)]), os.path.join(root, "test_openpose", "tmp", "%05d.json" % i))
            img.save(os.path.join(root, "test_img", "tmp", "%04d.jpg" % i))
        return w

    def frames(w):
        return {os.path.relpath(p, w): open(p, "rb").read()
                for p in sorted(glob.glob(os.path.join(w, "results", "fadg0", "test_latest", "*", "fake_B_*.jpg")))}
    two = ["--n_scales_spatial", "2", "--n_blocks_local", "1"]
    ws = {k: work(k) for k in ("single", "chunks", "stitched")}
    _run_test_py(ws["single"], two, env)
    _run_test_py(ws["chunks"], two + ["--shard_chunks", "--chunks_per_rank", "2"], env)
    _run_test_py(ws["stitched"], two + ["--shard_chunks", "--chunks_per_rank", "2", "--stitch_frames", "100"], env)
    one, ch, st = (frames(ws[k]) for k in ("single", "chunks", "stitched"))
    assert len(one) == 10 and one.keys() == ch.keys() == st.keys()
    assert any(one[k] != ch[k] for k in one)          # the chunk restarts both levels' recurrences at the cut
    assert all(one[k] == st[k] for k in one)          # ... the stitch pass carries both levels' tails across it
