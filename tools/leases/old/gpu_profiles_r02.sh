#!/bin/bash
# Round-2 profile set (gpurun_out/r02_prof -> digested into profiles/r02 by tools/summarize_prof.py):
#   bench      : bench.py (headline GEMM + quant + the extra legs): kernel trace, two SQ counter passes, FETCH / WRITE
#   attn       : tools/attn_bench.py (fwd decode + causal / full prefill at BASELINE configs[2])
#   moe        : tools/moe_bench.py 64 2048 (fused_experts int4 W4A16 and bf16 at BASELINE configs[4])
#   mla        : kbench mla 128 8192 128 (flash_mla_decode at BASELINE configs[3])
#   qserve     : tools/qserve_bench.py 16 4096
# One rocprofv3 invocation per pass, the program directly after "--" (no shell in between), counters never combined
# with trace flags.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_prof
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
run_set() {  # name, program + args...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- "$@" > $OUT/$name.trace.log 2>&1
  timeout 900 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/$name/pmc_sq1 -- "$@" > $OUT/$name.sq1.log 2>&1
  timeout 900 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/$name/pmc_sq2 -- "$@" > $OUT/$name.sq2.log 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/pmc_fetch -- "$@" > $OUT/$name.fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/pmc_write -- "$@" > $OUT/$name.write.log 2>&1
}
SETS=${1:-"bench_full bench attn moe mla qserve"}   # optional argument: the sets to (re)collect
mkdir -p $OUT/bench $OUT/attn $OUT/moe $OUT/mla $OUT/qserve
for s in $SETS; do
  case $s in
    bench_full) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_full/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_full.trace.log 2>&1 ;;
    bench) run_set bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra ;;
    attn) run_set attn python3 $R/tools/attn_bench.py ;;
    moe) run_set moe python3 $R/tools/moe_bench.py 64 2048 ;;
    mla) run_set mla $K mla 128 8192 128 ;;
    qserve) run_set qserve python3 $R/tools/qserve_bench.py 16 4096 ;;
  esac
done
cd $R
for s in $SETS; do
  python3 tools/summarize_prof.py $OUT/$s --to $OUT/digest --tag $s > $OUT/$s.summary.txt 2>&1
done
# drop the raw per-dispatch tables (tens of MB): the digests and logs are what is kept
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT; ls $OUT/digest
