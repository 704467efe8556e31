#!/bin/bash
# Round-3 profile set (gpurun_out/r03/prof -> digests copied into profiles/r03):
#   bench  : bench.py (headline GEMM + quant + flash_mla_decode roofline leg; no extra legs): trace, SQ counters, FETCH / WRITE
#   mla    : kbench mla 128 8192 128 (q x 100 gaussian logits as the reference benchmark; rows128x kernel vs the round-2 kernel)
#   attn   : tools/attn_decode_sweep.py (fwd decode d = 64 / 128 / 256 / fp8 KV)
#   moe    : tools/moe_bench.py 64 2048 (fused_experts int4 W4A16: streaming kernels at 64 tokens, moe_persist.hip at 2048)
#   qserve : tools/qserve_bench.py 16 4096 (decode split kernel, W4A8 modes of the persistent int8 pipeline)
#   swiglu : tools/swiglu_bench.py (silu_and_mul, silu_and_mul_clamp, swiglu_gpt_oss_sigmoid_alpha streams)
# One rocprofv3 invocation per pass (tools/gpu_prof.sh), the program directly after "--", counters never with trace flags.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
SETS=${1:-"bench mla attn moe qserve"}
for s in $SETS; do
  case $s in
    bench) PROF_MEM=1 tools/gpu_prof.sh bench python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra ;;
    bench_full) tools/gpu_prof.sh bench_full python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline ;;
    mla) MLA_AB=1 MLA_GAUSS=100 PROF_MEM=1 tools/gpu_prof.sh mla $R/sgl-kernel-xpu_amd/build/kbench mla 128 8192 128 2 ;;
    attn) tools/gpu_prof.sh attn python3 $R/tools/attn_decode_sweep.py ;;
    moe) tools/gpu_prof.sh moe python3 $R/tools/moe_bench.py 64 2048 ;;
    qserve) tools/gpu_prof.sh qserve python3 $R/tools/qserve_bench.py 16 4096 ;;
    swiglu) PROF_MEM=1 tools/gpu_prof.sh swiglu python3 $R/tools/swiglu_bench.py ;;
  esac
done > $R/gpurun_out/r03/prof_all.log 2>&1
ls $R/gpurun_out/r03/prof/digest
