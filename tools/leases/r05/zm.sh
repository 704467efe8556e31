#!/bin/bash
# r05 lease zm: fused_experts int4 at other models' shapes (many small experts) across token counts
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_zm
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
for shp in 128,2048,768,8 256,7168,2048,8 64,4096,1536,6; do
  MOE_SHAPE=$shp timeout 900 python3 tools/row_sweep.py moe 2>&1 | grep "fused_experts" | tee $OUT/moe_$shp.log
done
