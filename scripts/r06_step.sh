mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_pieces.py tests/test_gpu_backward.py -x -q 2>&1 | tail -25 > gpurun_out/r06/pytest_train_step.txt
tail -5 gpurun_out/r06/pytest_train_step.txt
for i in 1 2 3; do for mode in 1 0; do
  echo -n "D_BATCHED=$mode " >> gpurun_out/r06/train_ab2.txt
  T2V_D_BATCHED=$mode python scripts/train_bench.py --iters 16 2>&1 | tail -1 | cut -c1-110 >> gpurun_out/r06/train_ab2.txt
done; done
echo -n "SKIP_FUSED=0 " >> gpurun_out/r06/train_ab2.txt
T2V_SKIP_GRAD_FUSED=0 python scripts/train_bench.py --iters 16 2>&1 | tail -1 | cut -c1-110 >> gpurun_out/r06/train_ab2.txt
echo -n "default " >> gpurun_out/r06/train_ab2.txt
python scripts/train_bench.py --iters 16 2>&1 | tail -1 | cut -c1-110 >> gpurun_out/r06/train_ab2.txt
cat gpurun_out/r06/train_ab2.txt
