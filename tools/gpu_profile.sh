#!/bin/bash
# rocprofv3 kernel-trace stats + HBM traffic PMC passes of the headline bench
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -20
find $OUT -name "*kernel_stats.csv" -exec head -12 {} \;
du -sh $OUT
