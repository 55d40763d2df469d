set -x
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_train_step.py -x -q 2>&1 | tail -25 > gpurun_out/r06/pytest_train_step.txt
tail -5 gpurun_out/r06/pytest_train_step.txt
python scripts/train_bench.py --iters 12 2>&1 | tail -1 > gpurun_out/r06/train_bench_batched.txt
T2V_D_BATCHED=0 python scripts/train_bench.py --iters 12 2>&1 | tail -1 >> gpurun_out/r06/train_bench_batched.txt
python scripts/train_bench.py --iters 12 2>&1 | tail -1 >> gpurun_out/r06/train_bench_batched.txt
cat gpurun_out/r06/train_bench_batched.txt
python scripts/train_bench.py --iters 2 --aten_kernels 2>&1 | tail -60 > gpurun_out/r06/aten_kernels_batched.txt
