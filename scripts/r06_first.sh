mkdir -p gpurun_out/r06
python scripts/wgrad_placement_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/wgrad_placement.txt
timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q -k "two_scale" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "bench_contract" 2>&1 | tail -15
