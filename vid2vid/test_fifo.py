#!/usr/bin/env python
"""Long-lived variant of test.py (north_star names test_fifo.py; its source is not in the reference
tree, SURVEY R8 -- ASSUMPTION: same options as test.py, model loaded once, one utterance per request
read from a named pipe).

    python test_fifo.py --fifo /tmp/t2v.fifo --name fadg0 --dataroot datasets/fadg0 <test.py flags>

Each line written to the FIFO is a request: empty / "run" re-runs the configured dataroot; otherwise
whitespace-separated `key=value` overrides (name=, dataroot=, how_many=, results_dir=); "quit" ends.
After every request `<results dir>/.done` is (re)written with the frame count.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from text2video_amd.model import create_model, run_test  # noqa: E402
from text2video_amd.options import TestOptions            # noqa: E402


def main():
    opt = TestOptions().parse()
    if not opt.fifo:
        sys.exit("test_fifo.py: --fifo PATH is required")
    if not os.path.exists(opt.fifo):
        os.mkfifo(opt.fifo)
    model = create_model(opt)
    print("test_fifo: model resident, waiting on %s" % opt.fifo, flush=True)
    while True:
        with open(opt.fifo) as fh:
            for line in fh:
                line = line.strip()
                if line == "quit":
                    return
                for kv in line.split():
                    if "=" in kv:
                        k, v = kv.split("=", 1)
                        if k in ("name", "dataroot", "results_dir"):
                            setattr(opt, k, v)
                        elif k == "how_many":
                            opt.how_many = int(v)
                stats = run_test(opt, model)
                with open(os.path.join(stats["results_dir"], ".done"), "w") as done:
                    done.write("%d\n" % stats["frames"])
                print("test_fifo: %d frames, %.2f fps" % (stats["frames"], stats["fps_loop"]), flush=True)


if __name__ == "__main__":
    main()
