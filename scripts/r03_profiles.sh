#!/bin/bash
# Round-3 profile set, one GPU box.  Outputs under gpurun_out/r03/ (copied into profiles/r03_* afterwards).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
# 1. default bench line, unprofiled and under the profiler
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python bench.py --cpu-frames 0 --e2e-frames 0 > $O/bench_profiled.json 2> $O/bench_prof.err
cp $(find $O/bench_prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
# 2. per-frame kernel tables on ONE stream: 1 / 2 / 4 sequences in lock-step
for nb in 1 2 4; do
  T2V_STREAMS=1 bash scripts/prof_frames.sh r03b$nb --frames $((40 / nb)) --batch $nb > $O/frames_flow_batch${nb}_1stream.txt 2>&1
done
T2V_STREAMS=1 T2V_NORM_TICKET=1 bash scripts/prof_frames.sh r03t --frames 20 --batch 2 > $O/frames_flow_batch2_ticket_1stream.txt 2>&1
# 3. PMC on the batched Winograd GEMM stage inside frames (1 / 2 sequences: the fixed-grid kernel; 4: one block per tile):
#    one counter group per run
for nb in 1 2 4; do
  i=0
  pat="wino_gemm_sk"; [ $nb = 4 ] && pat="false, false, 2"
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); out=$O/pmc_gemm_b$nb/p$i; mkdir -p $out
    T2V_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/frame_prof.py --frames 6 --batch $nb > $out/log.txt 2>&1
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python scripts/pmc_summary.py ${f%_counter_collection.csv} "$pat" > $O/pmc_wino4_gemm_batch${nb}_p$i.txt 2>&1
  done
done
# 4. two-stream A/B of the knobs tried this round (alternating runs on this one box)
for i in 1 2; do
  python scripts/batch_probe.py --flow 1 --batches 1,2,4
  T2V_WINO_GEMM_SK=0 python scripts/batch_probe.py --flow 1 --batches 1,2,4
  T2V_NORM_TICKET=1 python scripts/batch_probe.py --flow 1 --batches 1,2
  T2V_WINO_GEMM_SK=0 T2V_WINO_GEMM_TILE=2 python scripts/batch_probe.py --flow 1 --batches 2,4
done > $O/ab_batch_ticket_tile.txt 2>&1
python scripts/sk_probe.py > $O/sk_probe.txt 2>&1
# 5. train step
bash scripts/r03_train_profiles.sh $O
# the raw traces are tens of MB each: keep the summaries only (gpurun merges at most 64 MiB back)
rm -rf $O/bench_prof $O/pmc_gemm_b2 $O/pmc_gemm_b4 gpurun_out/prof_frames_r03* gpurun_out/prof_train
ls -la $O
