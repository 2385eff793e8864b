#!/usr/bin/env python3
"""Condense the counter passes of profiles/collect_pmc_r03.sh into one tracked JSON.

usage: python profiles/summarize_pmc.py <tag> <dir with one sub-directory per pass> [kernel substring ...]
Every pass directory <run>_<A|B|C> holds rocprofv3's *_counter_collection.csv; per run and kernel the median over
the dispatches of every counter is kept (the first dispatches of a kernel include cold caches: medians, not means),
plus a few ratios that are independent of the units the counters tick in:
  valu_per_wave        SQ_INSTS_VALU / SQ_WAVES
  wave_life_quads      SQ_WAVE_CYCLES / SQ_WAVES            (quad-cycles a wave is resident)
  wait_share           SQ_WAIT_ANY / SQ_WAVE_CYCLES         (parked on s_waitcnt / barrier)
  valu_quads_per_inst  SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
  valu_busy_share      SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES or SQ_BUSY_CYCLES)
"""
import csv
import glob
import json
import os
import statistics
import sys


def main():
    tag, root = sys.argv[1:3]
    wanted = sys.argv[3:]
    runs = {}
    for d in sorted(glob.glob(os.path.join(root, "*_[A-Z]"))):
        run = os.path.basename(d)[:-2]
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv"))
        if not files:
            continue
        per = {}
        for row in csv.DictReader(open(files[0])):
            k = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            k = k.split("(")[0].strip() or row["Kernel_Name"][:60]
            if wanted and not any(w in k for w in wanted):
                continue
            per.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            per[k].setdefault("_grid", []).append(float(row.get("Grid_Size", 0) or 0))
            per[k].setdefault("_vgpr", []).append(float(row.get("VGPR_Count", 0) or 0))
            per[k].setdefault("_lds", []).append(float(row.get("LDS_Block_Size", 0) or 0))
        for k, ctrs in per.items():
            dst = runs.setdefault(run, {}).setdefault(k, {})
            for c, v in ctrs.items():
                dst[c if not c.startswith("_") else c[1:]] = statistics.median(v)
            dst["dispatches"] = len(next(iter(ctrs.values())))
    for run, kernels in runs.items():
        for k, c in kernels.items():
            g = lambda n: c.get(n)
            r = {}
            if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
                r["valu_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
            if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
                r["wave_life_quads"] = g("SQ_WAVE_CYCLES") / g("SQ_WAVES")
            if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
                r["wait_share"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
            if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
                r["issue_stall_share"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
            if g("SQ_ACTIVE_INST_VALU") and g("SQ_INSTS_VALU"):
                r["valu_quads_per_inst"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
            busy = g("SQ_BUSY_CU_CYCLES") or g("SQ_BUSY_CYCLES")
            if g("SQ_VALU_MFMA_BUSY_CYCLES") and busy:
                r["valu_busy_over_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / busy
            c["derived"] = {a: round(b, 4) for a, b in r.items()}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), tag + "_pmc_summary.json")
    with open(out, "w") as f:
        json.dump({"source": root, "runs": runs}, f, indent=1, sort_keys=True)
    for run, kernels in runs.items():
        for k, c in kernels.items():
            print(run, k[:60], json.dumps(c.get("derived")))


if __name__ == "__main__":
    main()
