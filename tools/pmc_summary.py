#!/usr/bin/env python3
"""Fold rocprofv3 counter_collection / kernel_stats CSVs into small per-kernel summaries for profiles/.

  python tools/pmc_summary.py pmc  <dir with one sub-directory per --pmc pass>  out.json   (mean per launch, last launches)
  python tools/pmc_summary.py stats <kernel_stats.csv>                            out.csv    (short kernel names)
"""
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([\w:]+(?:<[^()]*?>)?)\(", name)
    return m.group(1) if m else name[:60]


def pmc(root: str, out: str):
    acc = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = {}
        for r in rows:
            per.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in per.items():
            v = v[len(v) // 2:]                      # drop warm-up launches (buffers growing, first-touch)
            acc.setdefault(k, {})[c] = sum(v) / len(v)
    json.dump(acc, open(out, "w"), indent=1, sort_keys=True)


def stats(path: str, out: str):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


if __name__ == "__main__":
    {"pmc": pmc, "stats": stats}[sys.argv[1]](sys.argv[2], sys.argv[3])
