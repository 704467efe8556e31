#!/usr/bin/env python3
"""Digest a directory of rocprofv3 outputs (one sub-directory per pass: --kernel-trace --stats, or --pmc) into a kernel
stats table and per-kernel counter averages. `summarize_prof.py DIR` prints them; with `--to profiles/rNN --tag NAME` it
also writes NAME_kernel_stats.csv (our kernels' rows) and NAME_pmc.json (counter averages per kernel, all passes)."""
import argparse, collections, csv, glob, json, os

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("--to")
ap.add_argument("--tag", default="prof")
ap.add_argument("--match", default="sglk", help="substring a kernel name must contain")
args = ap.parse_args()

stats_rows, header = [], None
for f in sorted(glob.glob(os.path.join(args.src, "**", "*kernel_stats.csv"), recursive=True)):
    rd = csv.reader(open(f))
    header = next(rd)
    for r in rd:
        if args.match in r[0]:
            stats_rows.append(r)
if header:
    print(",".join(header))
    for r in stats_rows:
        print(",".join([r[0][:70]] + r[1:]))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(args.src, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if args.match in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# a kernel launched with several shapes (decode / prefill) is split by its dispatch order being stable across passes:
# report the average and, when the values cluster in two groups, min / max as well
out = {}
for k, cs in sorted(agg.items()):
    out[k] = {c: {"dispatches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in sorted(cs.items())}
    print(k[:70])
    for c, v in out[k].items():
        print(f"    {c:36s} n={v['dispatches']:4d} avg={v['avg']:.4g} min={v['min']:.4g} max={v['max']:.4g}")
if args.to:
    os.makedirs(args.to, exist_ok=True)
    if header:
        with open(os.path.join(args.to, f"{args.tag}_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(header)
            w.writerows(stats_rows)
    if out:
        json.dump(out, open(os.path.join(args.to, f"{args.tag}_pmc.json"), "w"), indent=1)
