#!/usr/bin/env python3
"""Digest gpurun_out/prof (tools/gpu_profile.sh) into profiles/<round>/<tag>_{kernel_stats.csv,pmc.json}."""
import collections, csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof")
rnd, tag = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("r01", "bench_fp8_gemm")
dst = os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    print(open(stats[0]).read()[:1500])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"dispatches": len(v), "avg": sum(v) / len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(agg.items())}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
for k, cs in out.items():
    print(k[:90], {c: round(v["avg"], 1) for c, v in cs.items()})
