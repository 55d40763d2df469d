cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06p; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python bench.py --single-variant --batch-variants "" --steps 2 --warmup 1 --cpu-frames 0 --e2e-frames 0 --hires-frames 0 --train-steps 0 > $O/bench_profiled_single.json 2> $O/bench_prof.err
cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_single.csv
rm -rf $O/bench_prof
head -3 $O/bench_kernel_stats_single.csv | cut -c1-200
bash scripts/r06_suite.sh
