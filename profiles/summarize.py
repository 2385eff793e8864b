#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/<tag>_{kt,fetch,write,sq}) into the small
tracked files under profiles/.

usage: python profiles/summarize.py <round-tag> <gpurun_out prefix> <kernel substring>
e.g.   python profiles/summarize.py r01_forward gpurun_out/p2 k_forward_fused_strip

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are collected in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B per lane) coalesced read stream, so it is doubled.
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import statistics
import sys


def counters(path, kernel):
    out = {}
    files = glob.glob(os.path.join(path, "*", "*_counter_collection.csv"))
    if not files:
        return out
    for row in csv.DictReader(open(files[0])):
        if kernel in row["Kernel_Name"]:
            out.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {"dispatches": len(v), "median": statistics.median(v), "mean": sum(v) / len(v)} for k, v in out.items()}


def main():
    tag, prefix, kernel = sys.argv[1:4]
    here = os.path.dirname(os.path.abspath(__file__))
    summary = {"kernel": kernel, "source": prefix}
    kt = glob.glob(prefix + "_kt/*/*_kernel_stats.csv")
    if kt:
        shutil.copy(kt[0], os.path.join(here, tag + "_kernel_stats.csv"))
        for row in csv.DictReader(open(kt[0])):
            if kernel in row["Name"]:
                summary["kernel_stats"] = {k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}
        tr = glob.glob(prefix + "_kt/*/*_kernel_trace.csv")
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0]))
             if kernel in r["Kernel_Name"]]
        # the bench launches this kernel at several sizes (the 1024-plane batch, and 16 planes inside the
        # configs[3] leg): the roofline figure is about the largest one, so group the trace by grid size
        by_grid = {}
        for r in csv.DictReader(open(tr[0])):
            if kernel in r["Kernel_Name"]:
                by_grid.setdefault(int(r["Grid_Size_X"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        summary["kernel_trace_by_grid"] = {str(g): {"dispatches": len(v), "mean_ns": sum(v) / len(v), "median_ns": statistics.median(v),
                                                    "min_ns": min(v), "max_ns": max(v)} for g, v in sorted(by_grid.items())}
        first = next(r for r in csv.DictReader(open(tr[0])) if kernel in r["Kernel_Name"])
        summary["kernel_trace"] = {"dispatches": len(d), "median_ns": statistics.median(d), "mean_ns": sum(d) / len(d),
                                   "min_ns": min(d), "vgpr": first["VGPR_Count"], "sgpr": first["SGPR_Count"],
                                   "lds_bytes": first["LDS_Block_Size"], "workgroup": first["Workgroup_Size_X"],
                                   "grid": first["Grid_Size_X"]}
    # which kernel sources the numbers belong to: bench.py accepts the traffic figure only while this matches
    repo = os.path.dirname(here)
    sys.path.insert(0, repo)
    try:
        from bench import FORWARD_SOURCES
        h = hashlib.sha256()
        for rel in FORWARD_SOURCES:
            with open(os.path.join(repo, rel), "rb") as fsrc:
                h.update(fsrc.read())
        summary["forward_source_sha16"] = h.hexdigest()[:16]
    except Exception as exc:   # pragma: no cover
        summary["forward_source_sha16"] = None
        summary["forward_source_sha16_error"] = str(exc)
    f = counters(prefix + "_fetch", kernel).get("FETCH_SIZE")
    w = counters(prefix + "_write", kernel).get("WRITE_SIZE")
    if f and w:
        rd = 2.0 * f["median"] * 1024.0          # gfx950: FETCH_SIZE counts 64 B per 128 B request
        wr = w["median"] * 1024.0
        blocks = None
        pmc_files = glob.glob(os.path.join(prefix + "_fetch", "*", "*_counter_collection.csv"))
        for row in csv.DictReader(open(pmc_files[0])):
            if kernel in row["Kernel_Name"]:
                blocks = int(row["Grid_Size"])              # 64 threads = 64 blocks per workgroup
                break
        summary["hbm_traffic"] = {"FETCH_SIZE_KiB_median": f["median"], "WRITE_SIZE_KiB_median": w["median"],
                                  "read_bytes_corrected": rd, "write_bytes": wr, "total_bytes_per_launch": rd + wr,
                                  "blocks_per_launch": blocks, "bytes_per_block": (rd + wr) / blocks if blocks else None,
                                  "correction": "read = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM); write = WRITE_SIZE x 1024"}
    sq = counters(prefix + "_sq", kernel)
    if sq:
        summary["sq_counters_median"] = {k: v["median"] for k, v in sq.items()}
    bench = prefix + "_kt.json"
    if os.path.exists(bench):
        summary["bench_line_under_profiler"] = json.loads(open(bench).read())
    with open(os.path.join(here, tag + "_summary.json"), "w") as fo:
        json.dump(summary, fo, indent=1)
    print(json.dumps({k: summary[k] for k in summary if k != "bench_line_under_profiler"}, indent=1))


if __name__ == "__main__":
    main()
