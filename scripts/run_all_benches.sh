cd "$GRAFT_REPO_ROOT"
bash scripts/run_prof.sh > gpurun_out/prof_stdout.txt 2>&1
python bench.py --flow --cpu-frames 0 2>&1 | tail -1 > gpurun_out/bench_flow.json
python bench.py --height 1024 --width 1024 --steps 16 --warmup 4 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/bench_1024_single.json
python bench.py --height 1024 --width 1024 --scales 2 --steps 16 --warmup 4 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/bench_1024_2scale.json
python bench.py --height 1024 --width 1024 --scales 2 --flow --steps 16 --warmup 4 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/bench_1024_2scale_flow.json
for f in gpurun_out/bench_*.json; do echo $f; python -c "
import json,sys; d=json.loads(open('$f').read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['layer'])"; done
