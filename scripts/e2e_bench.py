"""End-to-end throughput of the drop-in test.py path (rasterise -> H2D -> generator -> D2H -> JPEG) at
full model size with seeded random weights, on a dataset laid out like the reference's L2 driver
output.  Usage: python scripts/e2e_bench.py [--frames 90] [--workers N] [extra test.py flags]"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image
from text2video_amd.keypoints import read_keypoints
from text2video_amd.model import run_test
from text2video_amd.options import TestOptions

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=90)
ap.add_argument("--workers", type=int, default=None)
args, extra = ap.parse_known_args()
src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
tmp = tempfile.mkdtemp()
root = os.path.join(tmp, "datasets", "fadg0")
os.makedirs(os.path.join(root, "test_openpose", "tmp")); os.makedirs(os.path.join(root, "test_img", "tmp"))
img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), (512, 384)))
for i in range(args.frames):
    shutil.copyfile(os.path.join(src, files[i % len(files)]), os.path.join(root, "test_openpose", "tmp", "%05d.json" % i))
    img.save(os.path.join(root, "test_img", "tmp", "%04d.jpg" % i))
argv = ("--name fadg0 --dataroot %s --dataset_mode pose --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 "
        "--openpose_only --how_many 1200 --no_first_img --random_drop_prob 0 --synthetic_weights 1 "
        "--results_dir %s --checkpoints_dir %s" % (root, os.path.join(tmp, "results"), os.path.join(tmp, "ckpt"))).split()
opt = TestOptions().parse(argv + extra)
opt.pose_workers = args.workers
t0 = time.perf_counter()
stats = run_test(opt)
print(json.dumps({"frames": stats["frames"], "fps_frame_loop": round(stats["fps_loop"], 2),
                  "seconds_total_incl_model_load": round(time.perf_counter() - t0, 2), "pose_workers": args.workers,
                  "geometry": "512x320 (scaleHeight 512, central crop)" if not opt.no_pose_crop else "512x680"}))
shutil.rmtree(tmp)
