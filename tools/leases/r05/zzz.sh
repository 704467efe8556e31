#!/bin/bash
# r05 lease zzz: every sweep mode of tools/row_sweep.py on the final tree (regression look; logs under gpurun_out/r05_zzz)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zzz
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for m in gemm fwd moe mla fwdbs elem elem2 prefill1 sample elem3 gemmbw moe2; do
  timeout 600 python3 tools/row_sweep.py $m > $OUT/$m.log 2>&1
  echo "== $m: $(wc -l < $OUT/$m.log) lines, rc $?"
done
