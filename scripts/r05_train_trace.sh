#!/bin/bash
# train step under rocprofv3 --kernel-trace (one stream): per-shape table (scripts/trace_shapes.py) + the usual per-kernel table
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
tag=${1:-a}; shift
rm -rf gpurun_out/prof_train_$tag; mkdir -p gpurun_out/prof_train_$tag
T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$tag -o train -- python scripts/train_bench.py --iters 2 "$@" > gpurun_out/prof_train_$tag/log.txt 2>&1
tail -1 gpurun_out/prof_train_$tag/log.txt
tr=$(find gpurun_out/prof_train_$tag -name "*kernel_trace.csv" | head -1)
python scripts/trace_shapes.py $tr 4 > $O/train_shapes_$tag.txt
cp $(find gpurun_out/prof_train_$tag -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_$tag.csv
rm -rf gpurun_out/prof_train_$tag
python scripts/train_bench.py --iters 5 | tail -1
head -50 $O/train_shapes_$tag.txt
