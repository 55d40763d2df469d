#!/bin/bash
# Round-5 profile set, one GPU box.  Outputs under gpurun_out/r05/ (copied into profiles/r05_* afterwards).
#   usage: r05_profiles.sh [bench] [frames] [pmc] [train] [head]      (default: all)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
what="${*:-bench frames pmc train head}"
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }

if has bench; then
  # 1. the default bench line (what the driver runs); then the headline variant ALONE under the profiler
  #    (--single-variant --batch-variants "": the dominant kernel's AVERAGE in the CSV is the line's roofline.ms_per_launch)
  python bench.py > $O/bench.json 2> $O/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python bench.py --single-variant --batch-variants "" --cpu-frames 0 --e2e-frames 0 --hires-frames 0 --train-steps 0 > $O/bench_profiled_single.json 2> $O/bench_prof.err
  cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_single.csv
  rm -rf $O/bench_prof
fi

if has frames; then
  # 2. per-frame kernel tables on ONE stream: 512x512, the reference's 512x320 and 512x680; and the two-stream 512x512 frames
  T2V_STREAMS=1 bash scripts/prof_frames.sh r05a --frames 40 > $O/frames_flow_512x512_1stream.txt 2>&1
  T2V_STREAMS=1 bash scripts/prof_frames.sh r05b --frames 40 --width 320 > $O/frames_flow_512x320_1stream.txt 2>&1
  T2V_STREAMS=1 bash scripts/prof_frames.sh r05d --frames 30 --width 680 > $O/frames_flow_512x680_1stream.txt 2>&1
  bash scripts/prof_frames.sh r05e --frames 40 > $O/frames_flow_512x512_2streams.txt 2>&1
  rm -rf gpurun_out/prof_frames_r05*
fi

if has pmc; then
  # 3. PMC on the GEMM stage inside two-stream 512x512 frames -- the 256x128 / one-block-per-CU form the bench line names --
  #    one counter group per run (kernel trace only)
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); out=$O/pmc_gemm_512/p$i; mkdir -p $out
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/frame_prof.py --frames 6 > $out/log.txt 2>&1
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python scripts/pmc_summary.py ${f%_counter_collection.csv} wino_gemm_sk > $O/pmc_wino4_gemm_512x512_2streams_p$i.txt 2>&1
  done
  rm -rf $O/pmc_gemm_512
fi

if has train; then
  # 4. train step: the default (weight gradients on the side stream) and everything on ONE stream with its per-kernel / per-shape tables
  ( python scripts/train_bench.py --iters 5
    T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 python scripts/train_bench.py --iters 5
    python scripts/train_bench.py --iters 5 --force_dist ) 2>&1 | grep -v "amdgpu.ids\|^warning" > $O/train_bench.txt
  T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 bash scripts/prof_train.sh > $O/train_step_kernel_summary_1stream.txt 2>&1
  cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) $O/train_step_kernel_stats_1stream.csv
  tr=$(find gpurun_out/prof_train -name "*kernel_trace.csv" | head -1)
  python scripts/trace_shapes.py $tr 4 > $O/train_step_shapes_1stream.txt
  rm -rf gpurun_out/prof_train
fi

if has head; then
  # 5. the 7x7 head alone + its PMC passes
  python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids > $O/head_bench.txt
  bash scripts/run_pmc_head.sh > /dev/null 2>&1
  cat gpurun_out/pmc_head_p1.txt gpurun_out/pmc_head_p2.txt gpurun_out/pmc_head_p3.txt > $O/pmc_head.txt
fi
ls -la $O
