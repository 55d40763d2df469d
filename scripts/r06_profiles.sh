#!/bin/bash
# Round-6 profile set, one GPU box.  Outputs under gpurun_out/r06p/ (copied into profiles/r06_* afterwards).
#   usage: r06_profiles.sh [bench] [frames] [train] [pmc] [pmcgemm]      (default: the first four)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06p; mkdir -p $O
what="${*:-bench frames train pmc}"
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }

if has bench; then
  # the default bench line (what the driver runs), then the headline variant ALONE under the profiler with 2 + 1 frames only:
  # the dominant kernel's rows in the CSV are then the roofline loops' own launches (in two-stream frames a launch shares its
  # CUs with the other stream's kernels and takes 170-190 us), their average is the line's roofline.ms_per_launch
  python bench.py > $O/bench.json 2> $O/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python bench.py --single-variant --batch-variants "" --steps 2 --warmup 1 --cpu-frames 0 --e2e-frames 0 --hires-frames 0 --train-steps 0 > $O/bench_profiled_single.json 2> $O/bench_prof.err
  cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_single.csv
  rm -rf $O/bench_prof
fi

if has frames; then
  # configs[3] at the round's end state: per-frame kernel tables on ONE stream, 1024x1024 single-scale and two-scale, + 512x512
  T2V_STREAMS=1 bash scripts/prof_frames.sh r06a --frames 16 --size 1024 > $O/frames_flow_1024x1024_single_scale_1stream.txt 2>&1
  T2V_STREAMS=1 bash scripts/prof_frames.sh r06b --frames 24 --size 1024 --scales 2 > $O/frames_flow_1024x1024_two_scale_1stream.txt 2>&1
  T2V_STREAMS=1 bash scripts/prof_frames.sh r06c --frames 40 > $O/frames_flow_512x512_1stream.txt 2>&1
  rm -rf gpurun_out/prof_frames_r06*
fi

if has train; then
  # train step: default timing, then everything on ONE stream with its per-kernel / per-shape tables
  ( python scripts/train_bench.py --iters 12
    T2V_D_BATCHED=0 python scripts/train_bench.py --iters 12
    python scripts/train_bench.py --iters 12
    python scripts/train_bench.py --iters 5 --force_dist ) 2>&1 | grep -v "amdgpu.ids\|^warning" > $O/train_bench.txt
  T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 T2V_D_BWD_STREAM=0 bash scripts/prof_train.sh > $O/train_step_kernel_summary_1stream.txt 2>&1
  cp $(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) $O/train_step_kernel_stats_1stream.csv
  tr=$(find gpurun_out/prof_train -name "*kernel_trace.csv" | head -1)
  python scripts/trace_shapes.py $tr 4 > $O/train_step_shapes_1stream.txt
  rm -rf gpurun_out/prof_train
  python scripts/train_bench.py --iters 2 --aten_kernels 2>&1 | grep -v "amdgpu.ids\|^warning\|Warning\|_warn_once" | cut -c1-200 > $O/train_step_aten_kernels.txt
fi

if has pmc; then
  # PMC on the direct weight gradient at the step's heaviest shapes, two frames per launch as the step runs them
  for shape in down512 up1024; do
    bash scripts/run_pmc_wgrad.sh $shape 2 > /dev/null 2>&1
    cat gpurun_out/pmc_wgrad_$shape/p*/summary.txt > $O/pmc_wgrad_${shape}_2frames.txt
  done
  rm -rf gpurun_out/pmc_wgrad_*
fi
if has pmcgemm; then
  # PMC on the bench line's dominant kernel -- the Winograd GEMM stage on 128x128 tiles, fixed grid of two blocks per CU --
  # inside two-stream 512x512 frames, one counter group per run (kernel trace only)
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); out=$O/pmc_gemm_512/p$i; mkdir -p $out
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/frame_prof.py --frames 6 > $out/log.txt 2>&1
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python scripts/pmc_summary.py ${f%_counter_collection.csv} wino_gemm_sk > $O/pmc_wino4_gemm_512x512_2streams_p$i.txt 2>&1
  done
  rm -rf $O/pmc_gemm_512
fi
ls -la $O
