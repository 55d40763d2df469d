#!/bin/bash
# Round-2 measurement session: everything profiles/r02_* is made from.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; rm -rf $O; mkdir -p $O
# 1. the default bench command under rocprofv3 (--kernel-trace --stats) and without the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python bench.py > $O/bench_prof.log 2>&1
grep '^{"metric' $O/bench_prof.log | tail -1 > $O/r02_bench_profiled.json
cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
python bench.py > $O/bench.log 2>&1; grep '^{"metric' $O/bench.log | tail -1 > $O/r02_bench.json
python bench.py --variant noflow --cpu-frames 0 --e2e-frames 0 2>/dev/null | grep '^{"metric' | tail -1 > $O/r02_bench_noflow_headline.json
# 2. single-stream per-kernel tables of the frame loop (no cross-stream overlap in the averages)
T2V_STREAMS=1 bash scripts/prof_frames.sh r02_flow > /dev/null 2>&1; cp gpurun_out/prof_frames_r02_flow/summary.txt $O/r02_frames_flow_1stream.txt
T2V_STREAMS=1 bash scripts/prof_frames.sh r02_noflow --noflow > /dev/null 2>&1; cp gpurun_out/prof_frames_r02_noflow/summary.txt $O/r02_frames_noflow_1stream.txt
# 3. other geometries (generator only)
for g in "1024 1024 1" "1024 1024 2" "512 680 1" "512 320 1"; do set -- $g
  python bench.py --height $1 --width $2 --scales $3 --cpu-frames 0 --e2e-frames 0 --steps 30 2>/dev/null | grep '^{"metric' | tail -1 > $O/r02_bench_$1x$2_s$3.json
done
# 4. train step (config-5 size per GPU)
python scripts/train_bench.py > $O/r02_train_bench.txt 2>&1
python scripts/train_bench.py --no_flow >> $O/r02_train_bench.txt 2>&1
python scripts/train_bench.py --no_flow --no_face >> $O/r02_train_bench.txt 2>&1
python scripts/train_bench.py --vgg >> $O/r02_train_bench.txt 2>&1
bash scripts/prof_train.sh > $O/r02_train_step_kernel_summary.txt 2>&1
cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) $O/r02_train_step_kernel_stats.csv
# 5. PMC passes on the dominant kernel (Winograd F(4x4) GEMM stage) and on the weight-gradient kernel
bash scripts/run_pmc.sh r02_wino4 --shapes rb1024 --winograd 2 --stages 2 > $O/pmc_wino4.txt 2>&1
for i in 1 2 3; do cp gpurun_out/pmc_r02_wino4/p$i/summary.txt $O/r02_wino4_gemm_p$i.txt; done
bash scripts/run_pmc_wgrad.sh rb1024 > $O/pmc_wgrad.txt 2>&1
for i in 1 2 3 4; do cp gpurun_out/pmc_wgrad_rb1024/p$i/summary.txt $O/r02_wgrad_rb1024_p$i.txt; done
ls -la $O | head -40
