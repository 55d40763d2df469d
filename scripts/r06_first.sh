mkdir -p gpurun_out/r06
python scripts/train_bench.py --iters 2 --aten_kernels 2>&1 | tail -75 | cut -c1-220 > gpurun_out/r06/aten_kernels_batched2.txt
