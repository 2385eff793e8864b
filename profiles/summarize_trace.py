#!/usr/bin/env python3
"""Per kernel and grid size: dispatches, median / min / mean duration from a rocprofv3 --kernel-trace directory.
usage: python profiles/summarize_trace.py <tag> <trace dir> [kernel substring ...]   ->  profiles/<tag>.json"""
import collections
import csv
import glob
import json
import os
import statistics
import sys


def main():
    tag, root = sys.argv[1:3]
    wanted = sys.argv[3:]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    d = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            if wanted and not any(w in k for w in wanted):
                continue
            d["%s grid=%s" % (k, r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {k: {"dispatches": len(v), "median_us": round(statistics.median(v), 1), "min_us": round(min(v), 1), "mean_us": round(statistics.mean(v), 1)}
           for k, v in sorted(d.items())}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), tag + ".json")
    with open(path, "w") as f:
        json.dump({"source": root, "unit": "microseconds per dispatch (rocprofv3 --kernel-trace)", "kernels": out}, f, indent=1)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    main()
