#!/bin/bash
# Round-5 profile sets (gpurun_out/r05/prof -> digests copied into profiles/r05), one rocprofv3 invocation per pass
# (tools/gpu_prof.sh), the program directly after "--", counters never with trace flags:
#   bench    : bench.py (headline GEMM + quant + flash_mla_decode roofline leg; no extra legs): trace, SQ counters, FETCH / WRITE
#   mla      : kbench mla 128 8192 128 (q x 100 gaussian logits as the reference benchmark): ONE kernel per call since round 5
#   qserve   : tools/qserve_bench.py 1 16 64 (the decode weight-stream kernel as of round 5)
#   attn     : tools/attn_decode_sweep.py (fwd decode d = 64 / 128 / 256 / fp8 KV), WITH the FETCH / WRITE passes this time
#   prefill_*: tools/attn_prefill_bench.py, one run per shape (ATTN_PREFILL_ONLY = 128 / 64 / chunk / softcap): the kernel
#              stats rows are per shape, not an average over three workloads
#   moe      : tools/moe_bench.py 64 512 2048 (fused_experts int4 W4A16)
#   gemm_slices / sampling / prefill_long : tools/row_sweep.py gemm (256 / 512 / 1024 rows: the K-slice units and their sum kernels),
#              sample (vocab 128256), prefill1 (chunks of one long sequence: KV splits of the 128-row-block kernel) - kernel trace only
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PROF_ROUND=r05
mkdir -p $R/gpurun_out/r05
SETS=${1:-"bench mla qserve attn prefill_128 prefill_64 prefill_chunk prefill_softcap"}
for s in $SETS; do
  case $s in
    bench) PROF_MEM=1 tools/gpu_prof.sh bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra ;;
    mla) MLA_GAUSS=100 PROF_MEM=1 tools/gpu_prof.sh mla $R/sgl-kernel-xpu_amd/build/kbench mla 128 8192 128 ;;
    qserve) PROF_MEM=1 tools/gpu_prof.sh qserve python3 $R/tools/qserve_bench.py 1 16 64 ;;
    attn) PROF_MEM=1 tools/gpu_prof.sh attn python3 $R/tools/attn_decode_sweep.py ;;
    prefill_128) ATTN_PREFILL_ONLY=128 tools/gpu_prof.sh attn_prefill_d128 python3 $R/tools/attn_prefill_bench.py ;;
    prefill_64) ATTN_PREFILL_ONLY=64 tools/gpu_prof.sh attn_prefill_d64 python3 $R/tools/attn_prefill_bench.py ;;
    prefill_chunk) ATTN_PREFILL_ONLY=chunk tools/gpu_prof.sh attn_prefill_chunk128 python3 $R/tools/attn_prefill_bench.py ;;
    prefill_softcap) ATTN_PREFILL_ONLY=softcap tools/gpu_prof.sh attn_prefill_softcap python3 $R/tools/attn_prefill_bench.py ;;
    moe) MOE_BENCH_INT4_ONLY=1 tools/gpu_prof.sh moe python3 $R/tools/moe_bench.py 64 512 2048 ;;
    gemm_slices) GEMM_MS=256,512,1024 tools/gpu_prof.sh gemm_slices python3 $R/tools/row_sweep.py gemm ;;
    sampling) tools/gpu_prof.sh sampling python3 $R/tools/row_sweep.py sample ;;
    prefill_long) tools/gpu_prof.sh attn_prefill_one_sequence python3 $R/tools/row_sweep.py prefill1 ;;
  esac
done > $R/gpurun_out/r05/prof_all.log 2>&1
ls $R/gpurun_out/r05/prof/digest
