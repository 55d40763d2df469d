#!/bin/bash
# bench.py under rocprofv3 (--kernel-trace --stats), outputs under gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py "$@" > gpurun_out/prof_bench/bench.log 2>&1
grep "^{\"metric" gpurun_out/prof_bench/bench.log | tail -1 > gpurun_out/prof_bench/bench.json
cut -c1-2400 gpurun_out/prof_bench/bench.json
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-230 {} | head -30'
