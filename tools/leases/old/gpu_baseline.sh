#!/bin/bash
# Round-start measurements on the GPU box: MFMA ceilings, GEMM loss budget + in-kernel timeline, SQ counter pass of the
# headline GEMM ("before"), kernel stats + counters of fwd (attn_fwd_kernel) at BASELINE configs[2].
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_base
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
K=$R/sgl-kernel-xpu_amd/build/kbench
cd /tmp && export TMPDIR=/tmp
{
  timeout 120 $K peak 512 256 2000
  timeout 120 $K peak 256 256 2000
  timeout 300 $K gemm 4096 14336 4096 4 14 15 16 17
  timeout 120 $K stamps 4096 14336 4096
} > $OUT/kbench.log 2>&1
ARGS="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
A="$R/tools/attn_bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/attn_trace -- python3 $A > $OUT/attn_trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/attn_sq -- python3 $A > $OUT/attn_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/attn_fetch -- python3 $A > $OUT/attn_fetch.log 2>&1
cd $R && timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r02_base"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(os.path.basename(os.path.dirname(d)), k, {c: round(sum(v)/len(v)/1e6, 3) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
    for f in glob.glob(d + "**/*kernel_stats.csv", recursive=True):
        print(open(f).read()[:3000])
PY
cat $OUT/kbench.log; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
