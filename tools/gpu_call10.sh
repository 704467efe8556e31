#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_c10
mkdir -p $OUT
cd $R
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
timeout 1200 python3 -m pytest tests/test_moe_gpu.py -x -q -m gpu -k "mxfp4 or golden or errors" > $OUT/pytest_mxfp4.log 2>&1
timeout 600 python3 tools/attn_decode_sweep.py > $OUT/decode_sweep.log 2>&1
cd /tmp && export TMPDIR=/tmp
A="$R/tools/attn_bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/attn_trace -- python3 $A > $OUT/attn_trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/attn_sq -- python3 $A > $OUT/attn_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/attn_sq2 -- python3 $A > $OUT/attn_sq2.log 2>&1
cd $R
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -8 $OUT/pytest_mxfp4.log; cat $OUT/decode_sweep.log; cat $OUT/summary.txt | cut -c1-200 | head -80
