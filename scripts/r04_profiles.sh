#!/bin/bash
# Round-4 profile set, one GPU box.  Outputs under gpurun_out/r04/ (copied into profiles/r04_* afterwards).
#   usage: r04_profiles.sh [bench] [frames] [pmc] [train] [wgradpmc]      (default: all)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
what="${*:-bench frames pmc train wgradpmc}"
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }

if has bench; then
  # 1. the default bench line (what the driver runs), and the same command under the profiler
  python bench.py > $O/bench.json 2> $O/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python bench.py --cpu-frames 0 --e2e-frames 0 --hires-frames 0 --train-steps 0 > $O/bench_profiled.json 2> $O/bench_prof.err
  cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
  rm -rf $O/bench_prof
fi

if has frames; then
  # 2. per-frame kernel tables on ONE stream: 512x512, the reference's 512x320 (one sequence and two in lock-step) and 512x680
  T2V_STREAMS=1 bash scripts/prof_frames.sh r04a --frames 40 > $O/frames_flow_512x512_1stream.txt 2>&1
  T2V_STREAMS=1 bash scripts/prof_frames.sh r04b --frames 40 --width 320 > $O/frames_flow_512x320_1stream.txt 2>&1
  T2V_STREAMS=1 bash scripts/prof_frames.sh r04c --frames 20 --width 320 --batch 2 > $O/frames_flow_512x320_batch2_1stream.txt 2>&1
  T2V_STREAMS=1 bash scripts/prof_frames.sh r04d --frames 30 --width 680 > $O/frames_flow_512x680_1stream.txt 2>&1
  T2V_STREAMS=1 T2V_WINO_GEMM_SK_RAGGED=0 bash scripts/prof_frames.sh r04e --frames 30 --width 680 > $O/frames_flow_512x680_whole_tiles_1stream.txt 2>&1
  T2V_STREAMS=1 T2V_WINO_GEMM_SK_RAGGED=0 bash scripts/prof_frames.sh r04f --frames 20 --width 320 --batch 2 > $O/frames_flow_512x320_batch2_whole_tiles_1stream.txt 2>&1
  rm -rf gpurun_out/prof_frames_r04*
fi

if has pmc; then
  # 3. PMC on the GEMM stage inside frames at the reference's geometries, one counter group per run (kernel trace only)
  # (512 1: the 256x128 / one-block-per-CU form the two-stream frames run under the overlap hint, forced here on one stream)
  for cfg in "320 1 wino_gemm_sk" "320 2 wino_gemm_skt" "680 1 wino_gemm_skt" "512 1 wino_gemm_sk"; do
    set -- $cfg; w=$1; nb=$2; pat=$3
    if [ $w = 512 ]; then export T2V_WINO_GEMM_SK_TALL=2; else unset T2V_WINO_GEMM_SK_TALL; fi
    i=0
    for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1)); out=$O/pmc_gemm_${w}_b$nb/p$i; mkdir -p $out
      T2V_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/frame_prof.py --frames 6 --width $w --batch $nb > $out/log.txt 2>&1
      f=$(find $out -name "*counter_collection.csv" | head -1)
      python scripts/pmc_summary.py ${f%_counter_collection.csv} "$pat" > $O/pmc_wino4_gemm_512x${w}_batch${nb}_p$i.txt 2>&1
    done
    rm -rf $O/pmc_gemm_${w}_b$nb
  done
  unset T2V_WINO_GEMM_SK_TALL
fi

if has train; then
  # 4. train step: the default (weight gradients on the side stream) and everything on ONE stream, each with its kernel table
  ( python scripts/train_bench.py --iters 5
    T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 python scripts/train_bench.py --iters 5
    python scripts/train_bench.py --iters 5 --force_dist ) 2>&1 | grep -v "amdgpu.ids\|^warning" > $O/train_bench.txt
  bash scripts/prof_train.sh > $O/train_step_kernel_summary.txt 2>&1
  cp gpurun_out/prof_train/train_kernel_stats.csv $O/train_step_kernel_stats.csv
  rm -rf gpurun_out/prof_train
  T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 bash scripts/prof_train.sh > $O/train_step_kernel_summary_1stream.txt 2>&1
  cp gpurun_out/prof_train/train_kernel_stats.csv $O/train_step_kernel_stats_1stream.csv
  rm -rf gpurun_out/prof_train
fi

if has wgradpmc; then
  # 5. PMC on the direct weight gradient of the stride-2 / transposed layers
  for shape in down512 up1024; do
    bash scripts/run_pmc_wgrad.sh $shape > $O/pmc_wgrad_s2_$shape.txt 2>&1
    rm -rf gpurun_out/pmc_wgrad_$shape
  done
fi
ls -la $O
