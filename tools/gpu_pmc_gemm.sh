#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K=$R/sgl-kernel-xpu_amd/build/kbench
for v in 8 11 1; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/v$v -- $K gemm 4096 14336 4096 $v > $OUT/v$v.log 2>&1
rocprofv3 --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/w$v -- $K gemm 4096 14336 4096 $v > $OUT/w$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_gemm"
for d in sorted(glob.glob(out + "/[vw]*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_8bit" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(os.path.dirname(d)), {k: round(sum(v)/len(v)/1e6, 3) for k, v in agg.items()})
PY
