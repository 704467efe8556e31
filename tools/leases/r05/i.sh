#!/bin/bash
# r05 lease i: the whole GPU suite, then bench.py (default flags)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_i
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python3 - <<'PY'
import json,os
r=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r05_i/bench.json").read().strip().splitlines()[-1])
print("value",r["value"],"ms_per_step",r["ms_per_step"])
print("roofline",{k:v for k,v in r["roofline"].items() if not isinstance(v,(dict,list))})
fd=r.get("roofline_flash_decode",{})
print("flash_decode",{k:v for k,v in fd.items() if k in ("achieved","frac","kernel_ms_avg","tflops","bf16_ceiling_tflops_1_wave_per_simd","frac_of_bf16_ceiling_measured")})
for k,v in sorted(r.get("roofline_extra",{}).items()): print(" ",k,v.get("achieved"),v.get("unit"),v.get("frac"),v.get("frac_of_ceiling_measured"))
print("cpu",r.get("cpu_baseline"))
PY
