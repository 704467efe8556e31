#!/bin/bash
# r05 lease ze: fp8 block-scale GEMM, odd units of a workgroup walk K backwards (probe 16 = variant 38): outputs against the default
# variant, interleaved timing (kbench gemmab), FETCH_SIZE of both
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_ze
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
LD_PRELOAD=$R/sgl-kernel-xpu_amd/build/libsglk_probes.so timeout 300 python3 tools/dbg/gemm_variant_check.py 38 2>&1 | grep -v amdgpu | tee $OUT/check.log
KB=$R/sgl-kernel-xpu_amd/build/kbench
timeout 300 $KB gemmab 4096 14336 4096 4:1 38:1 2>&1 | tee $OUT/ab.log
timeout 300 $KB gemmab 8192 8192 8192 4:1 38:1 2>&1 | tee -a $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
for v in 4 38; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -- $KB gemm 4096 14336 4096 $v > $OUT/fetch_$v.log 2>&1
  f=$(ls $OUT/fetch_$v/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" $v <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm_fp8bw" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(f"variant {sys.argv[2]}: FETCH_SIZE avg {sum(vals)/max(1,len(vals)):.0f} KiB over {len(vals)} dispatches (x2 per the gfx950 note = {2*sum(vals)/max(1,len(vals))*1024/1e6:.1f} MB)")
PY
  rm -rf $OUT/fetch_$v
done 2>&1 | tee $OUT/fetch.log
