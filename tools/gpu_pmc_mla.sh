#!/bin/bash
# PMC passes for flash_mla_decode (kbench mla B S H); never combined with tracing
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_mla
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K=$R/sgl-kernel-xpu_amd/build/kbench
H=${1:-128}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a$H -- $K mla 128 8192 $H > $OUT/a$H.log 2>&1
rocprofv3 --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/b$H -- $K mla 128 8192 $H > $OUT/b$H.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c$H -- $K mla 128 8192 $H > $OUT/c$H.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_mla"
for d in sorted(glob.glob(out + "/[abc]*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(os.path.basename(os.path.dirname(d)), k, {c: round(sum(v)/len(v)/1e6, 3) for c, v in cs.items()})
PY
