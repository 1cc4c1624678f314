#!/usr/bin/env python3
"""
One table for EVERY kernel on the hot path: words per cell, average launch time, bytes moved by construction, PMC-measured memory-side
traffic, GB/s and fraction of the 8 TB/s HBM peak (MI355X_MICROARCH.md).

Input: one directory per workload group, produced by tools/kernel_roofline.sh (= three rocprofv3 passes over tools/path_workload.py):
    <dir>/manifest.json                      kernel-name pattern -> bytes moved per launch (written by the workload itself)
    <dir>/stats/**/*kernel_trace.csv         rocprofv3 --kernel-trace --stats           -> launch durations
    <dir>/FETCH_SIZE/**/*counter_collection.csv, <dir>/WRITE_SIZE/**/...   rocprofv3 --kernel-trace --pmc <counter> (separate passes)
PMC units as the guide prescribes: FETCH_SIZE x 2048 B (gfx950 reports 1/2 of a wide streaming read at 1 KiB units), WRITE_SIZE x 1024 B;
both are additionally calibrated on the 512 MiB copy the workload runs first, and the calibrated unit is reported.

    python tools/kernel_roofline.py gpurun_out/roofline/f32_256 gpurun_out/roofline/f32_512 gpurun_out/roofline/f64_384 > profiles/r03_kernel_roofline.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

HBM_PEAK = 8.0e12
csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("phihip::", "")


def read_durations(d):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)     # us
    return per


def read_counter(d, counter):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def calibrate(per_counter):
    """ counter units per byte from the 512 MiB copy (elementwise copy kernel with the largest count) """
    known = 512.0 * 1024 * 1024
    best = 0.0
    for name, vals in per_counter.items():
        low = name.lower()
        if ("elementwise" in low or "copy" in low) and "march" not in low:
            best = max(best, max(vals))
    return known / best if best > 1e4 else None


def group_table(d):
    man = json.load(open(os.path.join(d, "manifest.json")))
    dur = read_durations(d)
    fetch, write = read_counter(d, "FETCH_SIZE"), read_counter(d, "WRITE_SIZE")
    unit_f, unit_w = calibrate(fetch), calibrate(write)
    rows = []
    for e in man["kernels"]:
        pat = re.compile(e["kernel"])
        names = [n for n in dur if pat.search(short(n))]
        if not names:
            rows.append(dict(label=e["label"], kernel_pattern=e["kernel"], launches=0, note="kernel did not run in this group"))
            continue
        if short(names[0]).startswith("march_kernel") and not man.get("pinned_plans"):
            # (runs without pinned plans only) the first-call autotune launches every (tile, chunk) candidate a few times: the plan the solve ran
            # is the variant with the most launches
            names = [max(names, key=lambda n: len(dur[n]))]
        t = [x for n in names for x in dur[n]]
        t_sorted = sorted(t)
        avg = sum(t) / len(t)
        med = t_sorted[len(t) // 2]
        moved = e["bytes_per_launch"]
        f = [x for n in names for x in fetch.get(n, [])]
        w = [x for n in names for x in write.get(n, [])]
        row = dict(label=e["label"], kernels=sorted({short(n) for n in names}), launches=len(t), words_per_cell=e["words_per_cell"],
                   basis=e["basis"], avg_us=round(avg, 2), median_us=round(med, 2), min_us=round(t_sorted[0], 2),
                   moved_bytes_per_launch=moved, moved_GBs=round(moved / avg / 1e3, 1), frac_of_8TBs=round(moved / (avg * 1e-6) / HBM_PEAK, 4))
        if moved <= 0:           # (a launch that moves nothing by construction: the empty fix-up kernels -- their time is the point)
            rows.append(row)
            continue
        if f and w:
            rb, wb = sum(f) / len(f) * 2048.0, sum(w) / len(w) * 1024.0
            row.update(pmc_read_bytes=int(rb), pmc_write_bytes=int(wb), pmc_bytes_per_launch=int(rb + wb),
                       pmc_GBs=round((rb + wb) / avg / 1e3, 1), pmc_over_moved=round((rb + wb) / moved, 3))
            if unit_f and unit_w:
                row["pmc_bytes_calibrated"] = int(sum(f) / len(f) * unit_f + sum(w) / len(w) * unit_w)
        rows.append(row)
    out = dict(group=man["group"], size=man["size"], build_id=man.get("build_id"), source_matches_tree=man.get("source_matches_tree"),
               fetch_unit_calibrated_B=unit_f, write_unit_calibrated_B=unit_w)
    pinned = man.get("pinned_plans")
    if pinned:
        # consistency of the evidence: the traced CG iteration (MATVEC + the mean of the two UPDATE forms, kernel durations only) against the wall
        # time of an UNTRACED iteration of the same process image and launch plans (which additionally contains two kernel boundaries)
        def avg_of(mode):
            r = [x for x in rows if x.get("kernels") and re.search(r"march_kernel<\w+, \d, \d, \d+, %d," % mode, x["kernels"][0])]
            return r[0]["avg_us"] if r else None
        mv, ur, ux = avg_of(2), avg_of(6), avg_of(7)
        out["pinned_plans"] = pinned.get("plans")
        if mv and ur and ux:
            traced = mv + 0.5 * (ur + ux)
            other = pinned["untraced_ms_per_cg_iteration"] * 1e3
            same = (man.get("same_process") or {}).get("ms_per_cg_iteration")
            wall = same * 1e3 if same else other
            out["cg_iteration_check"] = dict(traced_kernel_sum_us=round(traced, 2), wall_us=round(wall, 2), ratio=round(traced / wall, 4),
                                             within_3_percent=abs(traced / wall - 1.0) <= 0.03,
                                             wall_measured_in="the traced process (same workspace allocations as the traced launches)" if same else "the untraced plans process",
                                             untraced_process_wall_us=round(other, 2),
                                             workspace_placement=dict(traced_process=(man.get("same_process") or {}).get("workspace_placement"),
                                                                      untraced_process=pinned.get("workspace_placement")),
                                             note="traced = avg MATVEC + (avg UPDATE_R + avg UPDATE_X2) / 2 under rocprofv3 --kernel-trace; wall = 100 iterations "
                                                  "(hipEvents) / 100 incl. the two kernel boundaries of an iteration; same build, same pinned plans. r6: between "
                                                  "PROCESSES an iteration moves with the allocations that hold the workspace (cg.hip place_workspace), so the check "
                                                  "uses the traced process's own wall time; the untraced process's is listed beside it")
    out["kernels"] = rows
    return out


def main():
    out = dict(peak_GBs=HBM_PEAK / 1e9,
               method="durations: rocprofv3 --kernel-trace --stats (End - Start per dispatch, all launches of the workload); bytes moved: by "
                      "construction (tools/path_workload.py manifest); pmc_*: rocprofv3 --pmc FETCH_SIZE x 2048 B + WRITE_SIZE x 1024 B, separate "
                      "passes (MI355X_MICROARCH.md); frac = moved bytes / avg time / 8 TB/s. 256^3 fp32 arrays (67 MB) fit the 256 MiB Infinity "
                      "Cache: that group is cache-assisted, the 512^3 / 384^3 fp64 groups are HBM-resident.",
               groups=[group_table(d) for d in sys.argv[1:]])
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
