cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python bench.py --single-variant --batch-variants "" --steps 2 --warmup 1 --cpu-frames 0 --e2e-frames 0 --hires-frames 0 --train-steps 0 > $O/bench_profiled_single.json 2> $O/bench_prof.err
cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_single.csv
rm -rf $O/bench_prof
head -4 $O/bench_kernel_stats_single.csv | cut -c1-160
python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['box']['sclk_mhz_mean'], d['train_step']['ms_per_step'], d['train_step']['exchange']['ms_per_step_without'])
d=json.load(open('$O/bench_profiled_single.json')); print(d['value'], d['roofline']['ms_per_launch'])
"
