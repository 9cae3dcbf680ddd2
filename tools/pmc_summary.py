#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per dispatch, per kernel."""
import csv
import sys
from collections import defaultdict

def main(path, filt=""):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if filt and filt not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k[:100])
        for c, v in sorted(d.items()):
            print(f"   {c:32s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
