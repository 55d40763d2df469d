#!/bin/bash
# Per-kernel stats of the reference's one-shot command itself (vid2vid/test.py on the configs[0] utterance: 2 x 85 frames,
# 512x320, lock-step, no torch in the process) under rocprofv3.   gpurun -- 'bash scripts/prof_lean_command.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lean_prof
python -c "
import bench, json, os
out = os.path.abspath('gpurun_out/lean_prof')
r = bench.cold_start_block(wrap=['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', out, '-o', 'lean', '--'])
print(json.dumps(r['cold_start']))
" > gpurun_out/lean_command_split.json
f=$(find gpurun_out/lean_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/lean_command_kernel_stats.csv
rm -rf gpurun_out/lean_prof
head -12 gpurun_out/lean_command_kernel_stats.csv | cut -c1-150
