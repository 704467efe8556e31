#!/bin/bash
# One profile set on the GPU box: kernel trace + two SQ counter passes (+ FETCH / WRITE with PROF_MEM=1) of ONE program,
# digested by tools/summarize_prof.py into gpurun_out/<round>/prof/digest/<name>_{kernel_stats.csv,pmc.json}.
#   tools/gpu_prof.sh <name> <program> [args...]     (the program directly after "--": no shell in between; counters
#   are never combined with trace flags)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${PROF_ROUND:-r04}/prof
name=$1; shift
mkdir -p $OUT/$name
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- "$@" > $OUT/$name.trace.log 2>&1
timeout 900 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/$name/pmc_sq1 -- "$@" > $OUT/$name.sq1.log 2>&1
timeout 900 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/$name/pmc_sq2 -- "$@" > $OUT/$name.sq2.log 2>&1
if [ -n "$PROF_MEM" ]; then
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/pmc_fetch -- "$@" > $OUT/$name.fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/pmc_write -- "$@" > $OUT/$name.write.log 2>&1
fi
cd $R
python3 tools/summarize_prof.py $OUT/$name --to $OUT/digest --tag $name > $OUT/$name.summary.txt 2>&1
find $OUT/$name -name "*_kernel_trace.csv" -delete
find $OUT/$name -name "*counter_collection.csv" -delete
tail -60 $OUT/$name.summary.txt
