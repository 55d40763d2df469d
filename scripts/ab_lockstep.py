"""A/B of --batch_sequences 2 as one batched call per step (T2V_LOCKSTEP=1) against one call per sequence (=0) on the test.py
path end to end, at the reference's 512x320 / 512x680 and at 512x512: two sequence folders, full-size generator, seeded weights."""
import contextlib, io, os, shutil, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image
from text2video_amd.keypoints import read_keypoints
from text2video_amd.model import run_test, create_model
from text2video_amd.options import TestOptions

src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 85
model = None
for canvas, extra, geom in [((512, 384), [], "512x320"), ((512, 384), ["--no_pose_crop"], "512x680"), ((512, 512), ["--no_pose_crop"], "512x512")]:
    tmp = tempfile.mkdtemp(prefix="t2v_ls_")
    try:
        root = os.path.join(tmp, "datasets", "fadg0")
        img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), canvas))
        for q, seq in enumerate(("tmp", "tmp_smooth")):
            os.makedirs(os.path.join(root, "test_openpose", seq)); os.makedirs(os.path.join(root, "test_img", seq))
            for i in range(n + 2):
                shutil.copyfile(os.path.join(src, files[(i + 7 * q) % len(files)]), os.path.join(root, "test_openpose", seq, "%05d.json" % i))
                img.save(os.path.join(root, "test_img", seq, "%04d.jpg" % i))
        for rep in range(3):
            for bs, force in (("1", None), ("2", "0"), ("2", "1")):
                argv = ("--name fadg0 --dataroot %s --dataset_mode pose --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 "
                        "--openpose_only --how_many 1200 --no_first_img --random_drop_prob 0 --synthetic_weights 1 --results_dir %s "
                        "--checkpoints_dir %s --batch_sequences %s" % (root, os.path.join(tmp, "res"), os.path.join(tmp, "ckpt"), bs)).split() + extra
                opt = TestOptions().parse(argv)
                if model is None:
                    import bench
                    model = bench.build_models(torch.device("cuda:0"), os.environ.get("AB_FLOW", "1") == "1", 1)[0]
                model.reset()
                os.environ.pop("T2V_LOCKSTEP", None)
                if force is not None:
                    os.environ["T2V_LOCKSTEP"] = force
                with contextlib.redirect_stdout(io.StringIO()):
                    st = run_test(opt, model=model, device="cuda:0")
                print("%s  batch_sequences %s  lockstep %-4s  %.2f fps (%d frames)" % (geom, bs, force, st["fps_loop"], st["frames"]), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
