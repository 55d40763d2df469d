#!/bin/bash
# Train-step part of the round-3 profile set (scripts/r03_profiles.sh step 5); usage: r03_train_profiles.sh <outdir>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/r03}; mkdir -p $O
# 5. train step: everything on, then the round's changes switched off one after the other
( python scripts/train_bench.py --iters 5
  T2V_PACK_PREFETCH=0 python scripts/train_bench.py --iters 5
  T2V_PACK_PREFETCH=0 T2V_WGRAD_STREAM=0 python scripts/train_bench.py --iters 5
  T2V_PACK_PREFETCH=0 T2V_WGRAD_STREAM=0 T2V_WGRAD_SK=0 python scripts/train_bench.py --iters 5
  T2V_PACK_PREFETCH=0 T2V_WGRAD_STREAM=0 T2V_WGRAD_SK=0 T2V_WINO_GEMM_SK=0 python scripts/train_bench.py --iters 5
  T2V_PACK_PREFETCH=0 T2V_WGRAD_STREAM=0 T2V_WGRAD_SK=0 T2V_WINO_GEMM_SK=0 T2V_DGRAD_TRANSPOSED=0 python scripts/train_bench.py --iters 5
  T2V_PACK_PREFETCH=0 T2V_WGRAD_STREAM=0 T2V_WGRAD_SK=0 T2V_WINO_GEMM_SK=0 T2V_DGRAD_TRANSPOSED=0 T2V_D_SHARED_FWD=0 python scripts/train_bench.py --iters 5
  T2V_PACK_PREFETCH=0 T2V_WGRAD_STREAM=0 T2V_WGRAD_SK=0 T2V_WINO_GEMM_SK=0 T2V_DGRAD_TRANSPOSED=0 T2V_D_SHARED_FWD=0 T2V_GRAD_DIRECT=0 T2V_WGRAD_COMBINE=0 python scripts/train_bench.py --iters 5
  python scripts/train_bench.py --iters 5 --no_flow; python scripts/train_bench.py --iters 5 --no_flow --no_face; python scripts/train_bench.py --iters 5 --vgg ) 2>&1 | grep -v "amdgpu.ids\|^warning" > $O/train_bench.txt
bash scripts/prof_train.sh > $O/train_step_kernel_summary.txt 2>&1
cp gpurun_out/prof_train/train_kernel_stats.csv $O/train_step_kernel_stats.csv
rm -rf gpurun_out/prof_train
