#!/bin/bash
# round 4: the profile sets (tools/gpu_profiles_r04.sh) + quant_extra parity
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
mkdir -p $R/gpurun_out/r04_q
timeout 600 python3 -m pytest tests/test_quant_extra_gpu.py -m gpu -q > $R/gpurun_out/r04_q/pytest.log 2>&1
tail -3 $R/gpurun_out/r04_q/pytest.log
bash tools/gpu_profiles_r04.sh "bench mla prefill qserve moe attn"
for f in bench mla qserve moe attn; do echo "== $f"; grep -v "^    " $R/gpurun_out/r04/prof/$f.summary.txt | head -14 | cut -c1-180; done
