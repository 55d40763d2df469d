"""CPU tests: the oracle's anchors, the host-side mirror of the reference interface (options, pose
rasteriser vs golden vectors captured from the reference's own keypoint2img.py, dataset geometry,
output layout)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_CMD = ("--name fadg0 --dataroot %s --dataset_mode pose --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 "
           "--openpose_only --how_many 1200 --no_first_img --random_drop_prob 0")  # text2video_audio.sh:42


# ---------------------------------------------------------------- oracle anchors
def test_trainmode_batchnorm_equals_instancenorm_affine():
    x = torch.randn(1, 16, 9, 11)
    bn = torch.nn.BatchNorm2d(16, affine=True).train()
    with torch.no_grad():
        bn.weight.normal_(1, 0.1)
        bn.bias.normal_(0, 0.1)
        ref = F.instance_norm(x, eps=1e-5) * bn.weight.view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)
        assert (bn(x) - ref).abs().max().item() < 1e-5


def test_oracle_resample_identity_and_corner_alignment():
    from oracle.generator_ref import resample
    img = torch.randn(1, 3, 7, 13)
    assert (resample(img, torch.zeros(1, 2, 7, 13)) - img).abs().max().item() < 1e-5
    # +1 px flow in x shifts the image by exactly one pixel (pixel units, not normalised units)
    flow = torch.zeros(1, 2, 7, 13)
    flow[:, 0] = 1.0
    out = resample(img, flow)
    assert (out[..., :-1] - img[..., 1:]).abs().max().item() < 1e-5
    assert (out[..., -1] - img[..., -1]).abs().max().item() < 1e-5   # border clamp


def test_oracle_first_frame_raw_only_and_fifo():
    from oracle.generator_ref import CompositeGenerator, Vid2VidInferenceRef
    torch.manual_seed(0)
    net = CompositeGenerator(9, 3, 6, ngf=8, n_downsampling=2, n_blocks=2, no_flow=False, norm="batch")
    ref = Vid2VidInferenceRef([net])
    A = torch.randn(1, 3, 3, 16, 16).clamp(-1, 1)
    f0 = ref.inference(A)
    with torch.no_grad():
        raw = net.train()(A.reshape(1, 9, 16, 16), torch.zeros(1, 6, 16, 16), True)[0]
    assert torch.equal(f0, raw)
    assert torch.equal(ref.fake_B_prev[0][1], f0[0]) and ref.fake_B_prev[0][0].abs().max() == 0
    f1 = ref.inference(A)
    assert torch.equal(ref.fake_B_prev[0][0], f0[0]) and torch.equal(ref.fake_B_prev[0][1], f1[0])


def test_oracle_two_scale_shapes():
    from oracle.generator_ref import CompositeGenerator, CompositeLocalGenerator, Vid2VidInferenceRef
    g0 = CompositeGenerator(9, 3, 6, ngf=16, n_downsampling=2, n_blocks=2, no_flow=False)
    g1 = CompositeLocalGenerator(9, 3, 6, ngf_global=16, n_blocks_local=1, scale=1, no_flow=False)
    ref = Vid2VidInferenceRef([g0, g1])
    out = ref.inference(torch.randn(1, 3, 3, 32, 32).clamp(-1, 1))
    assert out.shape == (1, 3, 32, 32) and ref.fake_B_prev[1].shape == (2, 3, 16, 16)


def test_synthetic_weights_load_into_oracle_strictly(lib_built):
    from oracle.generator_ref import CompositeGenerator
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    spec = GeneratorSpec(ngf=8, n_blocks=3)
    sd = synthetic_state_dict(spec, 3)
    net = CompositeGenerator(9, 3, 6, 8, 3, 3, False, "batch")
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(("running" in k or "num_batches" in k) for k in missing)
    sd2 = synthetic_state_dict(spec, 3)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)   # deterministic


# ---------------------------------------------------------------- host mirror vs reference goldens
def test_rasteriser_matches_reference_golden_maps():
    """Bit-exact against maps captured from the reference's keypoint2img.read_keypoints
    (tests/golden/make_host_goldens.py; cv2.circle stubbed => discs excluded)."""
    from text2video_amd.keypoints import read_keypoints
    g = np.load(os.path.join(GOLD, "pose_maps_fadg0.npz"))
    for name, want in zip(g["names"], g["maps"]):
        got = read_keypoints(os.path.join(GOLD, "keypoints_fadg0", str(name)), (512, 384), hand_discs=False)
        assert got.dtype == np.uint8 and got.shape == (384, 512, 3)
        assert np.array_equal(got, want), name
        assert (want != 0).any(2).sum() > 4000
        fast = read_keypoints(os.path.join(GOLD, "keypoints_fadg0", str(name)), (512, 384), hand_discs=False,
                              exact_fit=False)
        assert (fast != want).any(2).sum() <= 25     # closed form flips <= 25 px per frame


def test_rasteriser_training_jitter_and_drops_match_reference_golden_maps():
    """random_drop_prob > 0 (limb / hand / face drops) and remove_face_labels (head and face jitter,
    /root/reference/keypoint2img.py:113-123): a RandomState seeded like the reference's global np.random stream
    reproduces the reference's maps bit for bit (tests/golden/make_jitter_golden.py); with random_drop_prob == 0 the
    jitter is off, as there."""
    from text2video_amd.keypoints import read_keypoints
    g = np.load(os.path.join(GOLD, "pose_maps_jitter.npz"))
    plain = {}
    for name, seed, prob, remove, want in zip(g["names"], g["seeds"], g["probs"], g["remove"], g["maps"]):
        path = os.path.join(GOLD, "keypoints_fadg0", str(name))
        got = read_keypoints(path, (512, 384), float(prob), bool(remove), hand_discs=False,
                             rng=np.random.RandomState(int(seed)))
        assert np.array_equal(got, want), (name, seed, prob, remove)
        if prob == 0:
            plain[str(name)] = want
        else:
            assert str(name) not in plain or not np.array_equal(want, plain[str(name)])
    # the jitter really moves something: same seed, same drops, with and without remove_face_labels
    a = read_keypoints(path, (512, 384), 0.3, True, hand_discs=False, rng=np.random.RandomState(5))
    b = read_keypoints(path, (512, 384), 0.3, False, hand_discs=False, rng=np.random.RandomState(5))
    assert not np.array_equal(a, b)


def test_rasteriser_options_match_reference_golden_maps():
    """basic_point_only (pose limbs only) and canvas sizes other than the L2 driver's 512x384 -- key points are not
    rescaled, the canvas clips or leaves room (/root/reference/keypoint2img.py:70-90) -- bit-exact against maps captured
    from the reference (tests/golden/make_jitter_golden.py)."""
    from text2video_amd.keypoints import read_keypoints
    g = np.load(os.path.join(GOLD, "pose_maps_options.npz"))
    for i, (name, size, basic) in enumerate(zip(g["names"], g["sizes"], g["basic"])):
        want = g["map%d" % i]
        got = read_keypoints(os.path.join(GOLD, "keypoints_fadg0", str(name)), (int(size[0]), int(size[1])), 0, False, bool(basic),
                             hand_discs=False)
        assert got.shape == want.shape == (int(size[1]), int(size[0]), 3)
        assert np.array_equal(got, want), (name, tuple(size), basic)
        assert (want != 0).any(2).sum() > 500


def test_rasteriser_hand_discs_and_colour_key():
    from text2video_amd.keypoints import NOSE_NECK_RGB, read_keypoints
    p = os.path.join(GOLD, "keypoints_fadg0", "sa1_000_keypoints.json")
    with_d = read_keypoints(p, (512, 384))
    without = read_keypoints(p, (512, 384), hand_discs=False)
    diff = (with_d != without).any(2)
    ys, xs = np.nonzero(diff)
    assert 40 < diff.sum() < 120 and ys.max() <= 8 and xs.max() <= 8      # quarter disc of radius 8 at (0,0)
    assert tuple(with_d[0, 0]) == (255, 0, 0)                              # second (right-hand) disc wins
    assert (without == np.array(NOSE_NECK_RGB)).all(2).any()               # nose-neck limb colour present


def test_l2_driver_goldens_shape_and_monotone_names():
    g = np.load(os.path.join(GOLD, "l2_driver_Shehadyour.npz"))
    assert g["tmp"].shape == (87, 285) and g["tmp_smooth"].shape == (87, 285)
    assert list(g["tmp_names"]) == ["%05d.json" % i for i in range(87)]
    assert np.isfinite(g["tmp_smooth"]).all()


def test_reference_command_line_parses_and_sets_no_flow():
    from text2video_amd.options import TestOptions
    opt = TestOptions().parse((REF_CMD % "datasets/fadg0").split())
    assert opt.name == "fadg0" and opt.loadSize == 512 and opt.how_many == 1200 and opt.no_first_img
    assert opt.openpose_only and opt.no_flow and opt.batchSize == 1 and opt.n_frames_G == 3
    assert opt.ngf == 128 and opt.n_blocks == 9 and opt.n_downsample_G == 3 and opt.norm == "batch"
    opt2 = TestOptions().parse(["--name", "x", "--some_fork_flag", "3"])   # unknown flags do not abort
    assert opt2.name == "x"


def test_readme_train_command_parses():
    from text2video_amd.options import TrainOptions
    cmd = ("--name pose2body_256p --dataroot datasets/pose --dataset_mode pose --input_nc 3 --num_D 2 "
           "--resize_or_crop randomScaleHeight_and_scaledCrop --loadSize 544 --fineSize 512 "
           "--gpu_ids 0,1,2,3,4,5,6,7 --batchSize 8 --max_frames_per_gpu 2 --niter 500 --niter_decay 5 "
           "--no_first_img --n_frames_total 12 --max_t_step 4 --niter_step 100 --save_epoch_freq 100 "
           "--add_face_disc --openpose_only")   # README.md:171-176
    opt = TrainOptions().parse(cmd.split())
    assert opt.gpu_ids == list(range(8)) and opt.batchSize == 8 and opt.max_frames_per_gpu == 2 and opt.add_face_disc
    # the recipe passes --openpose_only and no --no_flow: training keeps the flow branch (inference lets the checkpoint decide)
    assert opt.openpose_only and not opt.no_flow
    assert TrainOptions().parse(cmd.split() + ["--no_flow"]).no_flow


def test_scale_height_geometry_and_central_crop():
    from text2video_amd.options import TestOptions
    from text2video_amd.pose_dataset import central_crop_cols, get_img_params
    opt = TestOptions().parse((REF_CMD % "x").split())
    assert get_img_params(opt, (512, 384)) == (680, 512)
    assert get_img_params(opt, (1280, 720)) == (912, 512)
    assert central_crop_cols(680) == (180, 500) and central_crop_cols(912) == (232, 680)


def test_pose_dataset_windows_names_and_change_seq():
    from text2video_amd.options import TestOptions
    from text2video_amd.pose_dataset import PoseDataset
    opt = TestOptions().parse((REF_CMD % os.path.join(GOLD, "dataset_fadg0_l2")).split())
    ds = PoseDataset(opt)
    assert len(ds) == 2 * (6 - 3 + 1)
    items = list(ds)
    assert [it["change_seq"] for it in items] == [True, False, False, False] * 2
    assert items[0]["A"].shape == (3, 512, 320, 3) and items[0]["A"].dtype == np.uint8
    assert np.array_equal(items[1]["A"][0], items[0]["A"][1])      # sliding window
    assert os.path.basename(items[0]["A_path"]).startswith("00002") and items[4]["seq"] == "tmp_smooth"
    # the process-pool prefetcher yields identical items (bit for bit, same order)
    pre = list(ds.iter_prefetch(workers=2))
    assert len(pre) == len(items)
    for a, b in zip(items, pre):
        assert np.array_equal(a["A"], b["A"]) and a["A_path"] == b["A_path"] and a["change_seq"] == b["change_seq"]
    assert len(list(ds.iter_prefetch(workers=2, limit=3))) == 3
    opt.no_pose_crop = True
    assert PoseDataset(opt)[0]["A"].shape == (3, 512, 680, 3)


def test_visualizer_output_layout(tmp_path):
    from text2video_amd.options import TestOptions
    from text2video_amd.visualizer import Visualizer, tensor2im_np
    opt = TestOptions().parse(["--name", "fadg0", "--results_dir", str(tmp_path)])
    vis = Visualizer(opt)
    img = np.zeros((8, 8, 3), np.uint8)
    paths = vis.save_images({"real_A": img, "fake_B": img}, "datasets/fadg0/test_img/tmp_smooth/smooth_0002.jpg")
    vis.flush()
    want = os.path.join(str(tmp_path), "fadg0", "test_latest", "tmp_smooth", "fake_B_smooth_0002.jpg")
    assert want in paths and os.path.exists(want)                  # image2video*.py globs fake_B_*.jpg
    x = np.array([[[-1.0, 0.2, 1.0]]], np.float32).transpose(2, 0, 1)
    assert tensor2im_np(x).ravel().tolist() == [0, 153, 255] or tensor2im_np(x).ravel().tolist() == [0, 152, 255]


def test_test_py_fails_loudly_without_checkpoint(tmp_path):
    """No checkpoint and no --synthetic_weights: the product path refuses to run (and never falls
    back to a CPU implementation)."""
    from text2video_amd import model as M
    from text2video_amd.options import TestOptions
    opt = TestOptions().parse((REF_CMD % os.path.join(GOLD, "dataset_fadg0_l2")).split()
                              + ["--checkpoints_dir", str(tmp_path)])
    with pytest.raises((FileNotFoundError, RuntimeError)):
        M.create_model(opt)


def test_train_dataset_sampling(tmp_path):
    """TrainPoseDataset: clip sampling (length, stride, start), one augmentation per clip, seeded determinism."""
    import shutil
    from PIL import Image
    from text2video_amd.keypoints import read_keypoints
    from text2video_amd.options import TrainOptions
    from text2video_amd.pose_dataset import TrainPoseDataset, get_train_img_params, get_video_params
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(gold) if f.startswith("sa1_"))
    root = tmp_path / "ds"
    for seq, reps in (("a", 3), ("b", 2)):
        os.makedirs(root / "train_openpose" / seq)
        os.makedirs(root / "train_img" / seq)
        for i, f in enumerate(files * reps):
            shutil.copyfile(os.path.join(gold, f), root / "train_openpose" / seq / ("%04d.json" % i))
            Image.fromarray(read_keypoints(os.path.join(gold, f), (256, 192))).save(root / "train_img" / seq / ("%04d.jpg" % i))
    opt = TrainOptions().parse(["--name", "x", "--dataroot", str(root), "--dataset_mode", "pose", "--input_nc", "3",
                                "--resize_or_crop", "randomScaleHeight_and_scaledCrop", "--loadSize", "136", "--fineSize", "128",
                                "--n_frames_total", "6", "--max_t_step", "3", "--random_drop_prob", "0", "--fast_pose"])
    ds = TrainPoseDataset(opt, seed=5)
    assert len(ds) == 2
    c = ds.sample(0)
    # 6 output frames + tG-1 = 2 warm-up frames; crop: height fineSize, width fineSize*w/h -> multiple of 32
    assert c["A"].shape == c["B"].shape == (8, 128, 160, 3) and c["A"].dtype == np.uint8
    assert 1 <= c["t_step"] <= 2 and c["start"] + 7 * c["t_step"] < 18            # 18-frame sequence: step <= (18-1)//7
    assert 128 <= c["params"]["new_size"][1] <= 136 and c["params"]["crop_size"] == (160, 128)
    assert c["A"].any() and c["B"].any()
    c2 = TrainPoseDataset(opt, seed=5).sample(0)
    assert np.array_equal(c["A"], c2["A"]) and np.array_equal(c["B"], c2["B"])     # seeded
    assert ds.sample(1)["seq"] == "b"
    # short sequence: the clip shrinks to what exists
    rng = np.random.default_rng(0)
    n, start, step = get_video_params(opt, 30, 12, rng)
    assert n == 12 and step == 1 and start == 0
    ds.update_training_batch(1)
    assert ds.n_frames_total == 12
    # plain `scaleHeight` (no crop) keeps the inference geometry
    opt.resize_or_crop = "scaleHeight"
    p = get_train_img_params(opt, (512, 384), rng)
    assert p["new_size"] == p["crop_size"] and p["crop_pos"] == (0, 0)


def test_oracle_vgg19_taps_match_the_references_torchvision():
    """oracle.generator_ref.VGG19Features against taps produced by the reference's own vendored torchvision
    vgg19 (tests/golden/make_vgg_golden.py imports it from /root/reference): same seeded weights and input."""
    import numpy as np
    from oracle.generator_ref import VGG19Features
    from text2video_amd.train import vgg19_random_state_dict
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vgg19_taps.npz"))
    net = VGG19Features().eval()
    net.load_state_dict(vgg19_random_state_dict(int(gold["seed"])))
    with torch.no_grad():
        taps = net(torch.from_numpy(gold["x"]))
    assert len(taps) == 5
    for i, t in enumerate(taps):
        ref = gold["tap%d" % i]
        assert tuple(t.shape) == ref.shape
        assert np.abs(t.numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), i


def test_oracle_adam_matches_the_references_torch041_optimizer():
    """oracle.optim_ref.adam_041_step against parameters produced by the reference's vendored torch-0.4.1 adam.py
    (tests/golden/make_adam_golden.py): five steps with gradients from 1e-6 to 30 and a few exact zeros."""
    import numpy as np
    from oracle.optim_ref import adam_041_step
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adam041.npz"))
    p = torch.from_numpy(gold["p0"].copy())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(gold["grads"].shape[0]):
        adam_041_step(p, torch.from_numpy(gold["grads"][s]), m, v, float(gold["lr"]), float(gold["beta1"]),
                      float(gold["beta2"]), float(gold["eps"]), s + 1)
        assert np.array_equal(p.numpy(), gold["p_after"][s]), s        # same fp32 operations in the same order
    assert np.array_equal(m.numpy(), gold["exp_avg"]) and np.array_equal(v.numpy(), gold["exp_avg_sq"])


def test_oracle_resample_restatement_matches_modern_grid_sample():
    """oracle.resample writes torch 0.4.1's grid sampler out (unclipped bilinear weights, clipped corner indices);
    modern F.grid_sample(align_corners=True, padding_mode='border') must give the same values everywhere and the
    same gradients wherever the sampling position is strictly inside the image and off the integer lattice."""
    import torch
    from oracle.generator_ref import resample, resample_modern
    torch.manual_seed(0)
    for (h, w, scale) in [(17, 23, 6.0), (32, 32, 0.8), (9, 40, 30.0)]:
        img = torch.randn(2, 3, h, w, requires_grad=True)
        flow = (torch.randn(2, 2, h, w) * scale).requires_grad_()
        a, b = resample(img, flow), resample_modern(img, flow)
        assert (a - b).abs().max().item() <= 2e-6
        g = torch.randn_like(a)
        ga = torch.autograd.grad((a * g).sum(), [img, flow])
        gb = torch.autograd.grad((b * g).sum(), [img, flow])
        assert (ga[0] - gb[0]).abs().max().item() <= 1e-5
        xs = torch.arange(w).view(1, 1, w) + flow.detach()[:, 0]
        ys = torch.arange(h).view(1, h, 1) + flow.detach()[:, 1]
        safe = ((xs > 0.01) & (xs < w - 1.01) & (ys > 0.01) & (ys < h - 1.01)
                & ((xs - xs.round()).abs() > 1e-3) & ((ys - ys.round()).abs() > 1e-3)).unsqueeze(1)
        assert ((ga[1] - gb[1]).abs() * safe).max().item() <= 1e-4 * max(1.0, gb[1].abs().max().item())
        # outside the image the coordinate's gradient is exactly zero (both clipped corners are the same pixel)
        outx = (xs < 0) | (xs > w - 1)
        if outx.any():
            assert ga[1][:, 0][outx].abs().max().item() == 0.0
    # identity flow returns the image (corner-aligned base grid), the anchor SURVEY 8c names
    img = torch.randn(1, 3, 12, 20)
    assert (resample(img, torch.zeros(1, 2, 12, 20)) - img).abs().max().item() <= 1e-5


def test_transforms_and_conv_init_pinned_to_the_vendored_sources(lib_built):
    """tests/golden/transforms.npz was produced by the reference's own vendored torchvision/transforms/functional.py
    (to_tensor, normalize, resize NEAREST) and torch/nn/modules/conv.py (reset_parameters), imported where they lie
    (make_transforms_golden.py).  The host side of row a2 and the seeded default init must reproduce it."""
    import torch
    from PIL import Image
    from text2video_amd.generator import GeneratorSpec, default_init_bound, layer_shapes, synthetic_state_dict
    from text2video_amd.options import TestOptions
    from text2video_amd.pose_dataset import get_img_params
    g = np.load(os.path.join(GOLD, "transforms.npz"))
    # ToTensor + Normalize(.5,.5): the formula the GPU kernel and `_real_A_u8` restate, bit for bit
    img = torch.from_numpy(g["img"])
    want = torch.from_numpy(g["img_norm"])
    assert torch.equal(((img.float() / 255.0 - 0.5) / 0.5).permute(2, 0, 1), want)
    allv = torch.arange(256, dtype=torch.uint8).float()
    assert torch.equal((allv / 255.0 - 0.5) / 0.5, torch.from_numpy(g["all_values_norm"])[0].reshape(-1))
    # NEAREST resize as the dataset applies it: scaleHeight 512 on the L2 driver's 512x384 canvas -> 680x512
    opt = TestOptions().parse(["--name", "x", "--resize_or_crop", "scaleHeight", "--loadSize", "512"])
    assert get_img_params(opt, (512, 384)) == (680, 512)
    got = np.asarray(Image.fromarray(g["sk"]).resize(get_img_params(opt, (512, 384)), Image.NEAREST))
    assert got.shape == (512, 680, 3) and np.array_equal(got, g["sk_680x512"])
    assert np.array_equal(np.asarray(Image.fromarray(g["small_src"]).resize((85, 64), Image.NEAREST)), g["small_85x64"])
    assert np.array_equal(np.asarray(Image.fromarray(g["small_src"]).resize((40, 30), Image.NEAREST)), g["down_40x30"])
    # torch-0.4.1 default conv init: stdv = 1/sqrt(in_channels*k*k), in_channels = dim 0 of a ConvTranspose2d weight
    for transposed, cin, cout, k, s0, s1, s2, s3, stdv in g["conv_init"]:
        assert abs(default_init_bound((int(s0), int(s1), int(s2), int(s3)), bool(transposed)) - stdv) <= 1e-15
    spec = GeneratorSpec(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch")
    sd = synthetic_state_dict(spec, 1, "uniform_fan_in")
    seen_t = 0
    for key, shape, role in layer_shapes(spec):
        if role in ("conv_w", "convT_w") and not key.startswith("model_final_flow"):
            b = default_init_bound(shape, role == "convT_w")
            mx = sd[key].abs().max().item()
            assert 0.9 * b <= mx <= b, (key, mx, b)
            seen_t += role == "convT_w"
    assert seen_t == 4      # the 2 + 2 transposed convs of the two decoders


def test_checkpoint_loading_rules(tmp_path):
    """`load_checkpoint` takes an upstream `latest_net_G0.pth` as is: `module.` prefixes of a DataParallel save are
    stripped, BatchNorm running statistics are dropped (the generator runs its norm layers on batch statistics at test
    time) -- which is also what torch 0.4.1's InstanceNorm does with pre-0.4 running stats on load
    ($SP/torch/nn/modules/instancenorm.py:15-38 removes `running_mean` / `running_var` when track_running_stats=False) --
    and a wrapped {'state_dict': ...} file is unwrapped."""
    import torch
    from text2video_amd.model import load_checkpoint
    sd = {"module.model_down_seg.1.weight": torch.randn(4, 3, 7, 7), "module.model_down_seg.2.running_mean": torch.zeros(4),
          "module.model_down_seg.2.running_var": torch.ones(4), "module.model_down_seg.2.num_batches_tracked": torch.tensor(5),
          "module.model_down_seg.2.weight": torch.ones(4).double()}
    p = str(tmp_path / "latest_net_G0.pth")
    torch.save(sd, p)
    out = load_checkpoint(p)
    assert sorted(out) == ["model_down_seg.1.weight", "model_down_seg.2.weight"]
    assert out["model_down_seg.2.weight"].dtype == torch.float32
    torch.save({"state_dict": {"a.weight": torch.ones(2)}}, p)
    assert list(load_checkpoint(p)) == ["a.weight"]


def test_direct_minpack_fit_equals_curve_fit_bit_for_bit():
    """The bit-exact rasteriser mode calls MINPACK's lmdif directly instead of going through scipy.optimize.curve_fit:
    same routine, same defaults, so the same bits -- on random two-point segments, incl. horizontal ones."""
    import warnings
    from scipy.optimize import curve_fit
    from text2video_amd import keypoints as K
    assert K._minpack() is not None
    rng = np.random.default_rng(0)
    n = 0
    for i in range(400):
        u, v = rng.uniform(0, 512, 2), rng.uniform(0, 384, 2)
        if i % 7 == 0:
            v[1] = v[0]
        if abs(u[0] - u[1]) < 1:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            (a, b), _ = curve_fit(K._affine, u, v)
        a2, b2 = K._fit_line(u, v, True)
        assert a == a2 and b == b2, (u, v)
        n += 1
    assert n > 300


def test_c_stamping_loops_equal_the_numpy_form(lib_built, monkeypatch):
    """csrc/raster_host.c (lib/libt2v_host.so) against keypoints.stamp's numpy form: random segments that overlap each
    other, run off the image (clamped coordinates -> repeated pixels), long and short, with and without end caps."""
    import importlib
    from text2video_amd import keypoints as K
    assert K._host_lib(), "libt2v_host.so not built (make -C text2video_amd/csrc)"
    rng = np.random.default_rng(5)
    segs = []
    for i in range(60):
        x = rng.uniform(-30, 230, 2)
        y = rng.uniform(-30, 180, 2)
        if i % 9 == 0:
            x = np.array([5.0, 195.0 + 900 * (i % 2)])      # a long one (1100 points: beyond the stack buffer)
        xs, ys = K.trace_segment(x, y, exact_fit=False)
        segs.append((xs, ys, int(rng.integers(1, 4)), tuple(int(v) for v in rng.integers(0, 256, 3)), bool(i % 2)))
    a = np.zeros((150, 200, 3), np.uint8)
    for xs, ys, bw, rgb, caps in segs:
        K.stamp(a, xs, ys, bw, rgb, caps)
    monkeypatch.setattr(K, "_HOST", False)                      # the numpy form
    b = np.zeros((150, 200, 3), np.uint8)
    for xs, ys, bw, rgb, caps in segs:
        K.stamp(b, xs, ys, bw, rgb, caps)
    assert a.any() and np.array_equal(a, b)


def test_host_rasteriser_loops_are_clean_under_the_sanitizers(tmp_path):
    """SURVEY section 5 (race detection / sanitizers): the host-side C of the frame path (csrc/raster_host.c) built with
    -fsanitize=address,undefined and driven over segments that hang over every border, pens wider than the image,
    key points far outside it and point counts past the stack buffer."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "text2video_amd", "csrc", "raster_host.c")
    drv = tmp_path / "drive.c"
    drv.write_text(r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
void t2v_raster_stamp(uint8_t* img, int h, int w, const long* xs, const long* ys, int n, int bw, const double* rgb, int caps);
int main(void) {
    const int dims[4][2] = {{1, 1}, {7, 5}, {64, 48}, {33, 130}};
    unsigned s = 12345u;
    for (int d = 0; d < 4; ++d) {
        const int h = dims[d][0], w = dims[d][1];
        uint8_t* img = (uint8_t*)calloc((size_t)h * w * 3, 1);
        for (int rep = 0; rep < 40; ++rep) {
            const int n = rep % 5 == 4 ? 3000 : 1 + (int)(s % 97u);
            long* xs = (long*)malloc(sizeof(long) * n);
            long* ys = (long*)malloc(sizeof(long) * n);
            for (int i = 0; i < n; ++i) {
                s = s * 1664525u + 1013904223u;
                xs[i] = (long)(s >> 8) % (3 * w + 1) - w;
                s = s * 1664525u + 1013904223u;
                ys[i] = (long)(s >> 8) % (3 * h + 1) - h;
            }
            if (rep % 7 == 0) { xs[0] = 4000000000000L; ys[n - 1] = -4000000000000L; }
            const double rgb[3] = {(double)(s & 255u), 170.0, 0.0};
            t2v_raster_stamp(img, h, w, xs, ys, n, 1 + rep % 4, rgb, rep & 1);
            free(xs);
            free(ys);
        }
        unsigned long sum = 0;
        for (long i = 0; i < (long)h * w * 3; ++i) sum += img[i];
        printf("%d x %d: %lu\n", h, w, sum);
        free(img);
    }
    return 0;
}
''')
    exe = tmp_path / "drive"
    r = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", str(drv), src, "-o", str(exe)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", LD_PRELOAD=""))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert r.stdout.count("\n") == 4
