"""Summarise rocprofv3 csv output of tools/profile_r.sh into one JSON (per-kernel averages)."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    res["kernel_stats"] = [r for r in csv.DictReader(open(f))][:8]
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r.get("Kernel_Name", "?")[:60]][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
    res[sub] = {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
