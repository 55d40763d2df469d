mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_device_oracle.py -q -s -k "face_discriminator" 2>&1 | grep "seed \|D_f on\|passed\|failed" > gpurun_out/r06/df_bound_seeds.txt
cat gpurun_out/r06/df_bound_seeds.txt
