#!/bin/bash
# what the driver runs at the end of the round: smoke(), then the bench command line of BENCH_r*.json
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06/bench_driver_cmd.json 2> gpurun_out/r06/bench_driver_cmd.err
tail -4 gpurun_out/r06/bench_driver_cmd.err
python - <<'PY'
import json
t = open("gpurun_out/r06/bench_driver_cmd.json").read()
lines = [l for l in t.splitlines() if l.startswith("{")]
j = json.loads(lines[-1])
print("line %d chars; stdout %d chars; value %.2f; roofline frac %.3f at %s MHz -> %.3f; train %.2f / %.2f ms at %s MHz; kernels %s"
      % (len(lines[-1]), len(t), j["value"], j["roofline"]["frac"], j["roofline"]["sclk_mhz"], j["roofline"]["frac_at_sclk"] or 0,
         j["train_step"]["exchange"]["ms_per_step_without"], j["train_step"]["exchange"]["ms_per_step_with"], j["train_step"]["sclk_mhz"],
         [(k["frac"], k["sclk_mhz"]) for k in j["train_step"]["kernels"]]))
PY
