"""GPU end-to-end test of the drop-in boundary B1: the reference's own test.py command line
(text2video_audio.sh:42) against a dataset in the layout its L2 driver writes, producing
results/<name>/test_latest/<seq>/fake_B_*.jpg."""
import glob
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _make_dataset(tmp):
    """datasets/fadg0 as interp_landmarks_motion_phoneme_VidTIMIT_smooth.py leaves it: pose JSONs
    (fixtures captured from the reference driver) + the skeleton jpgs it renders (:220,:266)."""
    from text2video_amd.keypoints import read_keypoints
    root = os.path.join(tmp, "vid2vid", "datasets", "fadg0")
    for seq, pat in (("tmp", "%04d.jpg"), ("tmp_smooth", "smooth_%04d.jpg")):
        src = os.path.join(GOLD, "dataset_fadg0_l2", "test_openpose", seq)
        os.makedirs(os.path.join(root, "test_openpose", seq))
        os.makedirs(os.path.join(root, "test_img", seq))
        for i, f in enumerate(sorted(os.listdir(src))):
            shutil.copyfile(os.path.join(src, f), os.path.join(root, "test_openpose", seq, f))
            Image.fromarray(read_keypoints(os.path.join(src, f), (512, 384))).save(
                os.path.join(root, "test_img", seq, pat % i))
    return os.path.join(tmp, "vid2vid")


def test_reference_command_line_end_to_end(tmp_path):
    work = _make_dataset(str(tmp_path))
    cmd = [sys.executable, os.path.join(ROOT, "vid2vid", "test.py"), "--name", "fadg0", "--dataroot", "datasets/fadg0",
           "--dataset_mode", "pose", "--input_nc", "3", "--resize_or_crop", "scaleHeight", "--loadSize", "512",
           "--openpose_only", "--how_many", "1200", "--no_first_img", "--random_drop_prob", "0",
           # no checkpoint ships with the reference: explicit opt-in to seeded random weights, small net
           "--synthetic_weights", "1", "--ngf", "32", "--n_blocks", "3", "--timing_json", "timing.json", "--write_video"]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0")
    r = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = os.path.join(work, "results", "fadg0", "test_latest")
    assert sorted(os.path.basename(p) for p in glob.glob(os.path.join(res, "tmp", "fake_B_*.jpg"))) == \
        ["fake_B_%04d.jpg" % i for i in range(2, 6)]
    assert sorted(os.path.basename(p) for p in glob.glob(os.path.join(res, "tmp_smooth", "fake_B_*.jpg"))) == \
        ["fake_B_smooth_%04d.jpg" % i for i in range(2, 6)]
    assert len(glob.glob(os.path.join(res, "*", "real_A_*.jpg"))) == 8
    im = np.asarray(Image.open(os.path.join(res, "tmp", "fake_B_0003.jpg")))
    assert im.shape == (512, 320, 3) and im.std() > 1.0          # 512 x 680 scaleHeight, central-width crop
    assert "process image..." in r.stdout
    # --write_video: the mux stage (image2video*.py) on the frames just written, 4 JPEG samples per track
    for seq in ("tmp", "tmp_smooth"):
        mp4 = os.path.join(work, "results", "fadg0", "fadg0_%s.mp4" % seq)
        blob = open(mp4, "rb").read()
        assert blob[4:8] == b"ftyp" and b"moov" in blob and b"mp4v" in blob
        first = open(os.path.join(res, seq, sorted(os.listdir(os.path.join(res, seq)))[0]), "rb").read()
        assert first in blob                                      # the frame file itself is the video sample
    # without --synthetic_weights the missing checkpoint is a hard error (no silent fallback)
    r2 = subprocess.run(cmd[:cmd.index("--synthetic_weights")] + ["--ngf", "32"], cwd=work, env=env, capture_output=True,
                        text=True, timeout=600)
    assert r2.returncode != 0 and "not found" in (r2.stderr + r2.stdout)


def test_lean_command_writes_the_same_files(tmp_path):
    """A plain single-device `test.py` run never imports torch: its frame loop takes buffers, stream and events from the
    library's host-plumbing entry points (text2video_amd/leantorch.py, include/t2v.h ABI 14) and reads the checkpoint with
    its own reader.  Same files, byte for byte, as the run on torch (T2V_LEAN=0) -- from a zip-container checkpoint and
    from the legacy stream torch 0.4.1 wrote."""
    import json
    import torch
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    work = _make_dataset(str(tmp_path))
    os.makedirs(os.path.join(work, "checkpoints", "fadg0"))
    ckpt = os.path.join(work, "checkpoints", "fadg0", "latest_net_G0.pth")
    sd = synthetic_state_dict(GeneratorSpec(ngf=32, n_downsample=3, n_blocks=3, no_flow=False, norm="batch"), 5, flow_gain=0.1)
    cmd = [sys.executable, os.path.join(ROOT, "vid2vid", "test.py"), "--name", "fadg0", "--dataroot", "datasets/fadg0",
           "--dataset_mode", "pose", "--input_nc", "3", "--resize_or_crop", "scaleHeight", "--loadSize", "512",
           "--openpose_only", "--how_many", "1200", "--no_first_img", "--random_drop_prob", "0", "--ngf", "32", "--n_blocks", "3",
           "--timing_json", "timing.json"]
    res = os.path.join(work, "results", "fadg0", "test_latest")

    def run(lean, legacy):
        shutil.rmtree(os.path.join(work, "results"), ignore_errors=True)
        torch.save(sd, ckpt, _use_new_zipfile_serialization=not legacy)
        # (T2V_UPLOAD_MIN_SPAN: this small net's checkpoint also goes up as one file span, as the 1.46 GB one does)
        env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", T2V_LEAN="1" if lean else "0", T2V_UPLOAD_MIN_SPAN=str(1 << 20))
        r = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        cs = json.load(open(os.path.join(work, "timing.json")))["cold_start"]
        assert cs["torch_imported"] is (not lean), (lean, cs)
        if lean:      # zip records are mirrored chunk by chunk, a legacy stream's tensors placed one by one; no fallback
            up = cs["upload"]
            assert up["mirrored"] is (not legacy) and up["as_views"] == up["tensors"] > 0 and "error" not in up, up
        assert "carries a flow branch" in r.stdout          # the architecture followed the checkpoint
        files = sorted(glob.glob(os.path.join(res, "*", "*.jpg")))
        return {os.path.relpath(f, res): open(f, "rb").read() for f in files}

    with_torch = run(False, False)
    assert len(with_torch) == 16
    assert run(True, False) == with_torch
    assert run(True, True) == with_torch


def test_resident_server_serves_the_one_shot_command(tmp_path):
    """test.py --resident: the command is a thin client of a server process that keeps the model on the GPU between calls
    (text2video_amd/resident.py).  Same command line, same files as the plain run, byte for byte; the second call reuses the
    server (its pid in the closing line); a checkpoint / weight change is noticed (another --synthetic_weights seed gives
    other frames); --resident_stop ends it."""
    work = _make_dataset(str(tmp_path))
    base = [sys.executable, os.path.join(ROOT, "vid2vid", "test.py"), "--name", "fadg0", "--dataroot", "datasets/fadg0",
            "--dataset_mode", "pose", "--input_nc", "3", "--resize_or_crop", "scaleHeight", "--loadSize", "512",
            "--openpose_only", "--how_many", "1200", "--no_first_img", "--random_drop_prob", "0", "--ngf", "32", "--n_blocks", "3"]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", T2V_RESIDENT_KEY=str(tmp_path))      # this test's own server
    res = os.path.join(work, "results", "fadg0", "test_latest")

    def run(extra, seed="1"):
        shutil.rmtree(os.path.join(work, "results"), ignore_errors=True)
        r = subprocess.run(base + ["--synthetic_weights", seed] + extra, cwd=work, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        files = sorted(glob.glob(os.path.join(res, "*", "*.jpg")))
        return r.stdout, {os.path.relpath(f, res): open(f, "rb").read() for f in files}

    try:
        _, plain = run([])
        out1, first = run(["--resident", "--resident_idle_s", "120"])
        out2, second = run(["--resident"])
        assert len(plain) == 16 and first == plain and second == plain
        assert "process image..." in out1 and "resident server, pid" in out1 and "resident server, pid" in out2
        pid = lambda o: o.rsplit("pid", 1)[1].split(")")[0].strip()      # noqa: E731
        assert pid(out1) == pid(out2)
        out3, other = run(["--resident"], seed="2")
        assert pid(out3) == pid(out1) and other.keys() == plain.keys() and other != plain
        # a failing request is reported with its status, and the server lives on
        r = subprocess.run(base + ["--resident"], cwd=work, env=env, capture_output=True, text=True, timeout=600)     # no checkpoint, no seed
        assert r.returncode != 0 and "not found" in (r.stdout + r.stderr)
        out4, again = run(["--resident"])
        assert pid(out4) == pid(out1) and again == plain
    finally:
        subprocess.run(base + ["--resident_stop"], cwd=work, env=env, capture_output=True, text=True, timeout=120)


def test_train_py_then_test_py_roundtrip(tmp_path):
    """train.py (reference flag surface, synthetic sequences: G + multiscale D + face D, Adam) writes
    upstream-format checkpoints that test.py loads and renders frames from."""
    work = _make_dataset(str(tmp_path))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0")
    common = ["--name", "fadg0", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--no_first_img",
              "--ngf", "32", "--n_blocks", "3"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "vid2vid", "train.py")] + common +
                       ["--num_D", "2", "--fineSize", "128", "--batchSize", "1", "--max_frames_per_gpu", "2", "--niter", "2",
                        "--add_face_disc", "--synthetic_data"], cwd=work, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "G_GAN" in r.stdout and "D_f" in r.stdout
    for f in ("latest_net_G0.pth", "latest_net_D.pth", "latest_net_D_f.pth"):
        assert os.path.exists(os.path.join(work, "checkpoints", "fadg0", f))
    # the files load STRICTLY into the reference-architecture modules (the oracle's stock torch.nn restatement):
    # parameters + BatchNorm buffers under upstream's key names
    import torch
    from oracle.generator_ref import CompositeGenerator, MultiscaleDiscriminator
    sd = torch.load(os.path.join(work, "checkpoints", "fadg0", "latest_net_G0.pth"), map_location="cpu")
    CompositeGenerator(9, 3, 6, 32, 3, 3, False, "batch").load_state_dict(sd, strict=True)
    sd = torch.load(os.path.join(work, "checkpoints", "fadg0", "latest_net_D.pth"), map_location="cpu")
    MultiscaleDiscriminator(6, 64, 3, 2, "batch").load_state_dict(sd, strict=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "vid2vid", "test.py")] + common +
                       ["--dataroot", "datasets/fadg0", "--resize_or_crop", "scaleHeight", "--loadSize", "512",
                        "--how_many", "3", "--random_drop_prob", "0"], cwd=work, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert len(glob.glob(os.path.join(work, "results", "fadg0", "test_latest", "tmp", "fake_B_*.jpg"))) == 3


def test_train_py_on_a_real_layout_dataset(tmp_path):
    """The reference's training recipe (README.md:171-176) on <dataroot>/train_openpose + train_img: random scale +
    scaled crop, clips of n_frames_total frames walked in chunks of max_frames_per_gpu with the recurrence carried."""
    from text2video_amd.keypoints import read_keypoints
    root = tmp_path / "vid2vid" / "datasets" / "fadg0"
    src = os.path.join(GOLD, "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
    for seq in ("clipA", "clipB"):
        os.makedirs(root / "train_openpose" / seq)
        os.makedirs(root / "train_img" / seq)
        for i, f in enumerate(files + files[::-1]):
            shutil.copyfile(os.path.join(src, f), root / "train_openpose" / seq / ("%04d_keypoints.json" % i))
            Image.fromarray(read_keypoints(os.path.join(src, f), (256, 192))).save(root / "train_img" / seq / ("%04d.jpg" % i))
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0")
    cmd = [sys.executable, os.path.join(ROOT, "vid2vid", "train.py"), "--name", "fadg0", "--dataroot",
           "datasets/fadg0", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2",
           "--resize_or_crop", "randomScaleHeight_and_scaledCrop", "--loadSize", "136", "--fineSize", "128",
           "--batchSize", "1", "--max_frames_per_gpu", "2", "--no_first_img",
           "--n_frames_total", "5", "--max_t_step", "2", "--niter_step", "100", "--add_face_disc",
           "--random_drop_prob", "0", "--ngf", "16", "--n_blocks", "2"]
    # 2 epochs at the initial learning rate + 1 of decay, one clip per sequence and epoch: 3 x 2 iterations
    r = subprocess.run(cmd + ["--niter", "2", "--niter_decay", "1"], cwd=tmp_path / "vid2vid", env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("(iter")]
    # (the face terms appear in the chunks whose pose maps show the nose-neck colour after the random crop)
    assert len(lines) == 6 and "5 frames 128x160" in lines[0] and any("D_f" in l for l in lines), r.stdout[-1500:]
    assert all(k in lines[0] for k in ("F_Flow", "F_Warp", "G_Warp")), lines[0]     # the recipe trains the flow branch
    assert [l.split(",")[1].strip() for l in lines] == ["epoch 1"] * 2 + ["epoch 2"] * 2 + ["epoch 3"] * 2
    assert r.stdout.count("update learning rate") == 1        # after epoch 3 (> niter)
    ck = tmp_path / "vid2vid" / "checkpoints" / "fadg0"
    for f in ("latest_net_G0.pth", "3_net_G0.pth", "3_net_D.pth", "3_net_D_f.pth", "iter.txt"):
        assert os.path.exists(ck / f), f
    assert open(ck / "iter.txt").read().split() == ["4", "0"]
    # --continue_train: reloads the latest nets, reads iter.txt, runs the one epoch that a longer schedule adds
    r = subprocess.run(cmd + ["--niter", "2", "--niter_decay", "2", "--continue_train"], cwd=tmp_path / "vid2vid", env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Resuming from epoch 4 at iteration 0" in r.stdout and "not found" not in r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("(iter")]
    assert len(lines) == 2 and all("epoch 4" in l for l in lines)
    assert os.path.exists(ck / "4_net_G0.pth")


def test_fifo_server_serves_requests(tmp_path):
    """test_fifo.py: model resident, one utterance per line written to the named pipe."""
    import time
    work = _make_dataset(str(tmp_path))
    fifo = os.path.join(str(tmp_path), "t2v.fifo")
    os.mkfifo(fifo)
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", PYTHONUNBUFFERED="1")
    cmd = [sys.executable, os.path.join(ROOT, "vid2vid", "test_fifo.py"), "--fifo", fifo, "--name", "fadg0", "--dataroot",
           "datasets/fadg0", "--dataset_mode", "pose", "--input_nc", "3", "--resize_or_crop", "scaleHeight", "--loadSize",
           "512", "--openpose_only", "--how_many", "1200", "--no_first_img", "--random_drop_prob", "0",
           "--synthetic_weights", "1", "--ngf", "32", "--n_blocks", "3"]
    proc = subprocess.Popen(cmd, cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        done = os.path.join(work, "results", "fadg0", "test_latest", ".done")
        for req, expect in (("run", "8"), ("how_many=3", "3")):
            if os.path.exists(done):
                os.remove(done)
            with open(fifo, "w") as fh:          # blocks until the server opens the pipe for reading
                fh.write(req + "\n")
            t0 = time.time()
            while not os.path.exists(done):
                assert proc.poll() is None, proc.stdout.read()[-2000:]
                assert time.time() - t0 < 300
                time.sleep(0.2)
            time.sleep(0.2)
            assert open(done).read().strip() == expect
        with open(fifo, "w") as fh:
            fh.write("quit\n")
        assert proc.wait(timeout=60) == 0
    finally:
        if proc.poll() is None:
            proc.kill()


def test_checkpoint_decides_flow_branch_and_legacy_keys(tmp_path):
    """A checkpoint that carries model_final_flow.* enables the flow-warp compositor even under
    --openpose_only (the architecture follows the checkpoint); `module.` prefixes and
    BatchNorm running statistics of old checkpoints are dropped."""
    import torch
    from text2video_amd import model as M
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    from text2video_amd.options import TestOptions
    ck = tmp_path / "ckpt" / "p"
    ck.mkdir(parents=True)
    spec = GeneratorSpec(ngf=32, n_blocks=2, no_flow=False)
    sd = {"module." + k: v for k, v in synthetic_state_dict(spec, 1).items()}
    sd["module.model_down_seg.2.running_mean"] = torch.zeros(32)
    sd["module.model_down_seg.2.num_batches_tracked"] = torch.tensor(5)
    torch.save(sd, str(ck / "latest_net_G0.pth"))
    base = ["--name", "p", "--checkpoints_dir", str(tmp_path / "ckpt"), "--ngf", "32", "--n_blocks", "2", "--openpose_only",
            "--no_first_img"]
    m = M.create_model(TestOptions().parse(base))
    assert m.nets[0].spec.no_flow is False
    m = M.create_model(TestOptions().parse(base + ["--no_flow"]))       # explicit flag wins
    assert m.nets[0].spec.no_flow is True
    spec_nf = GeneratorSpec(ngf=32, n_blocks=2, no_flow=True)
    torch.save(synthetic_state_dict(spec_nf, 1), str(ck / "latest_net_G0.pth"))
    m = M.create_model(TestOptions().parse(["--name", "p", "--checkpoints_dir", str(tmp_path / "ckpt"), "--ngf", "32",
                                            "--n_blocks", "2", "--no_first_img"]))
    assert m.nets[0].spec.no_flow is True                                # no flow keys -> no flow branch


def test_in_memory_pipeline_equals_file_pipeline(tmp_path):
    """The whole L0 recipe on the reference's data: drop-in L2 driver script (files) + test.py, against the
    in-memory pipeline (no JSON / skeleton-JPEG round trip).  Same frames, same names."""
    t2v = tmp_path / "Text2Video"
    shutil.copytree(os.path.join(GOLD, "l2_inputs"), t2v)                     # the reference's data layout
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0")
    text, person = "She had your dark suit in greasy wash water all year.", "fadg0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "Text2Video", "interp_landmarks_motion_phoneme_VidTIMIT_smooth.py"),
                        text, person], cwd=t2v, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ds = tmp_path / "vid2vid" / "datasets" / person
    assert len(os.listdir(ds / "test_openpose" / "tmp")) == 87 == len(os.listdir(ds / "test_img" / "tmp_smooth"))
    flags = ["--name", person, "--dataroot", "datasets/" + person, "--dataset_mode", "pose", "--input_nc", "3",
             "--resize_or_crop", "scaleHeight", "--loadSize", "512", "--openpose_only", "--how_many", "12",
             "--no_first_img", "--random_drop_prob", "0", "--synthetic_weights", "1", "--ngf", "32", "--n_blocks", "3"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "vid2vid", "test.py")] + flags, cwd=tmp_path / "vid2vid",
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted(glob.glob(str(tmp_path / "vid2vid" / "results" / person / "test_latest" / "tmp" / "fake_B_*.jpg")))
    assert [os.path.basename(f) for f in files] == ["fake_B_%04d.jpg" % i for i in range(2, 14)]
    # in-memory: same utterance, results into another directory
    r = subprocess.run([sys.executable, "-m", "text2video_amd.pipeline", text, person] + flags[flags.index("--how_many"):] +
                       ["--results_dir", str(tmp_path / "mem")], cwd=t2v, env=dict(env, PYTHONPATH=ROOT),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    mem = sorted(glob.glob(str(tmp_path / "mem" / person / "test_latest" / "tmp" / "fake_B_*.jpg")))
    assert [os.path.basename(f) for f in mem] == [os.path.basename(f) for f in files]
    for a, b in zip(files, mem):
        assert np.array_equal(np.asarray(Image.open(a)), np.asarray(Image.open(b))), (a, b)


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's contract fields plus roofline and cpu_baseline."""
    import json
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--cpu-frames", "1", "--kernel-iters", "4", "--e2e-frames", "12", "--hires-frames", "3", "--train-steps", "2"],
                       cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["dtype"] == "f32" and d["unit"] == "frames/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 0.01 * d["value"] and d["value"] > 30.0      # north_star: >= 30 fps
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] <= 1.0
    assert abs(rf["achieved"] - rf["gflop_per_launch"] / rf["ms_per_launch"]) <= 0.02 * rf["achieved"]
    assert 0.5 * rf["ms_per_launch"] < rf["ms_per_launch_bracketed"] < 1.15 * rf["ms_per_launch"]      # per-launch event pairs
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["cores"] >= 1 and 0 < cb["value"] < d["value"]
    assert cb["parity"]["frames"] == 2 and 0 < cb["parity"]["max_abs_delta_vs_oracle"] <= cb["parity"]["tolerance"] == 1e-3
    assert cb["value_8_threads"] is None or 0 < cb["value_8_threads"] < d["value"]
    assert cb["value"] == max(cb["by_threads"].values()) and str(cb["cores"]) in cb["by_threads"]
    # the headline is the generator with the flow-warp compositor; the no-flow variant is timed in the same run
    v = d["config"]["variants"]
    assert v["headline"] == "flow" and v["flow_fps"] == d["value"] and v["noflow_fps"] > v["flow_fps"] > 30.0
    assert "flow-warp compositor ON" in d["config"]["workload"]
    # end to end through the drop-in test.py frame loop (pose JSONs -> JPEG files)
    runs = d["e2e"]["runs"]
    assert {r["geometry"].split(" ")[0].rstrip(",") for r in runs} == {"512x512", "512x680", "512x320"}
    assert all(r["fps"] > 0 for r in runs) and d["e2e"]["frames_per_run"] == 12 and d["e2e"]["pose_workers"] >= 1
    # the two-sequence dataset (tmp + tmp_smooth, as the reference's L2 driver writes it), one at a time and in lock-step
    assert sorted(r["batch_sequences"] for r in runs if r["sequences"] == 2) == [1, 1, 2, 2]      # at 512x512 and at the reference's 512x320
    # N independent sequences per GPU in lock-step: aggregate rates beside the single-sequence headline
    assert v["batch2_fps"] > 30.0 and v["batch4_fps"] > 30.0 and d["value"] == v["flow_fps"]
    # the box the line was measured on and the core clock during the timed region (None where sysfs has no such file)
    box = d["box"]
    assert box["host"] and "sclk_mhz_mean" in box and (box["sclk_mhz_mean"] is None or 500 < box["sclk_mhz_mean"] < 3500)
    assert rf["traffic"] is None or "pmc_summary.json" in rf["traffic_source"]
    # the one-shot command as the reference starts it (a process per utterance), timed from outside, with its own split
    cs = d["e2e"]["cold_start"]
    assert cs["frames"] == 170 and len(cs["wall_s"]) == 2 and cs["wall_s"][-1] > cs["split_of_last_run"]["loop_s"] > 0
    sp = cs["split_of_last_run"]
    assert {"process_to_run_test_s", "weights_to_device_s", "first_step_s", "loop_s", "to_last_jpeg_s", "loop_split"} <= set(sp)
    # the plain command runs without torch (text2video_amd/leantorch.py), its checkpoint goes up as one file span; the same
    # command with torch is timed beside it
    assert sp["torch_imported"] is False and sp["upload"]["mirrored"] is True and sp["upload"]["threads"] >= 1
    assert set(sp["loop_split"]) == {"wait_pose_s", "upload_s", "enqueue_s", "finish_s"}
    assert cs["with_torch"]["wall_s"] > 0 and cs["with_torch"]["loop_s"] > 0
    assert cs["resident"]["warm_call_wall_s"] < cs["wall_s"][-1] and cs["resident"]["warm_loop_s"] > 0
    # the whole post-alignment utterance (text2video_audio.sh:24-44): L2 driver, test.py and the mux as three processes
    ut = d["e2e"]["utterance"]
    assert "error" not in ut and ut["frames"] == 170 and ut["videos"] == 2 and len(ut["chain_runs"]) == 2
    last = ut["chain_runs"][-1]
    assert set(last) == {"l2_driver_s", "test_py_s", "mux_s", "wall_s"} and all(v > 0 for v in last.values())
    assert abs(last["wall_s"] - (last["l2_driver_s"] + last["test_py_s"] + last["mux_s"])) < 0.2
    assert ut["l2_plus_mux_s"] < ut["reference_l2_driver_s"] and ut["l2_plus_mux_below_test_py"] is True
    assert len(ut["in_memory_pipeline_wall_s"]) == 2 and ut["in_memory_pipeline_wall_s"][-1] < ut["chain_wall_s"]
    assert ut["chain_resident_warm"]["test_py_s"] < last["test_py_s"]
    assert d["cpu_baseline"]["host_logical_cpus"] >= d["cpu_baseline"]["host_cpus_available"] >= 1
    # BASELINE configs[3]: 1024x1024 frames, single-scale and two-scale generator, both variants, + the GEMM stage at that size
    hi = d["hires"]
    for name in ("single_scale", "two_scale"):
        assert hi[name]["noflow_fps"] > hi[name]["flow_fps"] > 5.0, hi
    assert hi["two_scale"]["flow_fps"] > hi["single_scale"]["flow_fps"]
    g = hi["single_scale"]["gemm_stage"]
    assert g["kernel"].endswith("@128x128") and 0.3 < g["frac"] <= 1.0 and abs(g["achieved"] - g["gflop_per_launch"] / g["ms_per_launch"]) <= 0.02 * g["achieved"]
    # ... and the HBM-bound kernels of that config (north_star: "HBM-bound conv tiles"): algorithmic bytes / launch time / 8 TB/s
    hb = hi["hbm_bound"]["rows"]
    assert len(hb) == 8 and all(len(r) == 6 and r[1] > 0 and 0.005 < r[4] < 1.0 and abs(r[4] - r[3] / 8000.0) < 2e-3 for r in hb), hb
    assert {r[0].split(" ")[0] for r in hb} == {"G0", "G1", "norm"}
    # the core clock during the roofline kernel's loop and during the train block (round 6): a slow box and a slow kernel
    # read differently -- frac is against the 2.4 GHz spec peak, frac_at_sclk against the peak at the sampled clock
    assert "sclk_mhz" in rf and "frac_at_sclk" in rf
    if rf["sclk_mhz"]:
        assert 500 < rf["sclk_mhz"] < 3500 and abs(rf["frac_at_sclk"] - rf["frac"] * 2400.0 / rf["sclk_mhz"]) < 5e-3
        assert rf["frac_at_sclk"] <= 1.0 and rf["sclk_samples"] >= 3
    # BASELINE configs[4], one GPU's work: ms per optimiser step, the exchange as step_with - step_without + bytes + buckets,
    # the step's heaviest kernels against the fp32 MFMA peak
    t = d["train_step"]
    assert "configs[4]" in t["workload"] and t["steps"] == 2 and 20.0 < t["ms_per_step"] < 2000.0
    assert "sclk_mhz" in t and "ms_per_step_without_at_2360mhz" in t
    if t["sclk_mhz"]:
        assert abs(t["ms_per_step_without_at_2360mhz"] - t["exchange"]["ms_per_step_without"] * t["sclk_mhz"] / 2360.0) < 0.02
    ex = t["exchange"]
    assert ex["group"] == "1-rank rccl" and ex["bytes"] > 1.3e9 and ex["buckets"] >= 20 and ex["replicas_in_sync"] is None
    assert abs(ex["ms"] - (ex["ms_per_step_with"] - ex["ms_per_step_without"])) < 0.02
    assert len(t["kernels"]) == 3 and all(0.05 < k["frac"] < 1.0 for k in t["kernels"])
    assert all(np.isfinite(v) for v in t["losses"].values()) and "G_GAN" in t["losses"] and "D_f" in t["losses"]
    assert len(lines[0]) < 7200, len(lines[0])       # the driver keeps a bounded tail of stdout: the line stays compact


def test_lockstep_sequences_write_the_same_files_as_one_at_a_time(tmp_path):
    """test.py --batch_sequences 2 (the default: tmp and tmp_smooth advance in lock-step, one batched generator call per
    frame) against --batch_sequences 1, flow branch on: identical JPEG files, the same file set; unequal sequence
    lengths (the longer one finishes alone) and --how_many cutting into the second sequence."""
    from text2video_amd.keypoints import read_keypoints
    src = os.path.join(GOLD, "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))

    def dataset(work):
        root = os.path.join(work, "datasets", "fadg0")
        for seq, n in {"tmp": 9, "tmp_smooth": 6, "third": 5}.items():
            os.makedirs(os.path.join(root, "test_openpose", seq))
            os.makedirs(os.path.join(root, "test_img", seq))
            img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), (128, 96)))
            for i in range(n):
                shutil.copyfile(os.path.join(src, files[(i * 5 + len(seq)) % len(files)]),
                                os.path.join(root, "test_openpose", seq, "%05d.json" % i))
                img.save(os.path.join(root, "test_img", seq, "%04d.jpg" % i))

    def run(work, extra):
        cmd = [sys.executable, os.path.join(ROOT, "vid2vid", "test.py"), "--name", "fadg0", "--dataroot", "datasets/fadg0",
               "--dataset_mode", "pose", "--input_nc", "3", "--resize_or_crop", "scaleHeight", "--loadSize", "128",
               "--openpose_only", "--no_first_img", "--random_drop_prob", "0", "--synthetic_weights", "1", "--ngf", "16",
               "--n_blocks", "2", "--n_downsample_G", "2", "--pose_workers", "2", "--no_pose_crop"] + extra
        r = subprocess.run(cmd, cwd=work, env=dict(os.environ, CUDA_VISIBLE_DEVICES="0"), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        return {os.path.relpath(p, work): open(p, "rb").read()
                for p in sorted(glob.glob(os.path.join(work, "results", "fadg0", "test_latest", "*", "fake_B_*.jpg")))}

    outs = {}
    for name, extra in (("one", ["--batch_sequences", "1", "--how_many", "1200"]), ("two", ["--how_many", "1200"]),
                        ("three", ["--batch_sequences", "3", "--how_many", "1200"]),
                        ("one_cut", ["--batch_sequences", "1", "--how_many", "9"]), ("two_cut", ["--how_many", "9"])):
        w = str(tmp_path / name)
        os.makedirs(w)
        dataset(w)
        # a checkpoint-less run with the flow branch: --synthetic_weights builds it unless --openpose_only implies no_flow;
        # both forms are covered by the generator-level tests, here the file-level equality is the point
        outs[name] = run(w, extra)
    assert len(outs["one"]) == 7 + 4 + 3
    for other in ("two", "three"):
        assert outs["one"].keys() == outs[other].keys()
        assert all(outs["one"][k] == outs[other][k] for k in outs["one"]), other
    assert len(outs["one_cut"]) == 9 and outs["one_cut"].keys() == outs["two_cut"].keys()
    assert all(outs["one_cut"][k] == outs["two_cut"][k] for k in outs["one_cut"])
