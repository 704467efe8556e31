#!/bin/bash
# Round-4 profile sets (gpurun_out/r04/prof -> digests copied into profiles/r04), one rocprofv3 invocation per pass
# (tools/gpu_prof.sh), the program directly after "--", counters never with trace flags:
#   bench   : bench.py (headline GEMM + quant + flash_mla_decode roofline leg; no extra legs): trace, SQ counters, FETCH / WRITE
#   mla     : kbench mla 128 8192 128 (q x 100 gaussian logits as the reference benchmark)
#   prefill : tools/attn_prefill_bench.py (attn_prefill_kernel: causal d = 128 / 64, q = 128 chunk)
#   attn    : tools/attn_decode_sweep.py (fwd decode d = 64 / 128 / 256 / fp8 KV)
#   moe     : tools/moe_bench.py 64 512 2048 (fused_experts int4 W4A16)
#   qserve  : tools/qserve_bench.py 1 16 64 (the decode weight-stream kernel)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PROF_ROUND=r04
mkdir -p $R/gpurun_out/r04
SETS=${1:-"bench mla prefill qserve"}
for s in $SETS; do
  case $s in
    bench) PROF_MEM=1 tools/gpu_prof.sh bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra ;;
    mla) MLA_GAUSS=100 PROF_MEM=1 tools/gpu_prof.sh mla $R/sgl-kernel-xpu_amd/build/kbench mla 128 8192 128 2 ;;
    prefill) PROF_MEM=1 tools/gpu_prof.sh attn_prefill python3 $R/tools/attn_prefill_bench.py ;;
    attn) tools/gpu_prof.sh attn python3 $R/tools/attn_decode_sweep.py ;;
    moe) tools/gpu_prof.sh moe python3 $R/tools/moe_bench.py 64 512 2048 ;;
    qserve) PROF_MEM=1 tools/gpu_prof.sh qserve python3 $R/tools/qserve_bench.py 1 16 64 ;;
  esac
done > $R/gpurun_out/r04/prof_all.log 2>&1
ls $R/gpurun_out/r04/prof/digest
