mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=40 ) > gpurun_out/r06/gpu_suite.txt 2>&1
tail -60 gpurun_out/r06/gpu_suite.txt
