#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) on the Winograd-domain weight-gradient reduction of the
# 1024 -> 1024 ResnetBlock conv over two 512x512 frames: fixed grid (wino_wgrad_sk_kernel) and one block per tile
# (conv_wgrad_kernel with 36 "taps"); both run in scripts/wgrad_sk_probe.py.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_wgrad_sk; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); out=$O/p$i; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/wgrad_sk_probe.py 1 > $out/log.txt 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  ( python scripts/pmc_summary.py ${f%_counter_collection.csv} wino_wgrad_sk; python scripts/pmc_summary.py ${f%_counter_collection.csv} conv_wgrad_kernel ) > $O/r03_wgrad_winograd_p$i.txt 2>&1
  rm -rf $out
done
cat $O/*.txt
