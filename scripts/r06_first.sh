set -x
mkdir -p gpurun_out/r06
python scripts/train_bench.py --iters 2 --aten_kernels 2>&1 | tail -90 > gpurun_out/r06/aten_kernels_base.txt
D="d0,d0_dg,e0,e1,e2,e3,d3h_dg,d2h_dg,d1h_dg,f0,f1,f2,f3"
for b in 2 4 6; do
  echo "== tile 0 batch $b" >> gpurun_out/r06/dshape_sweep2.txt
  python scripts/kernel_bench.py --shapes $D --batch $b --iters 30 --warmup 40 >> gpurun_out/r06/dshape_sweep2.txt 2>&1
done
