#!/usr/bin/env python3
"""
Collects the per-launch HBM traffic of the CG kernels from tools/pmc_summary.py outputs (one per problem size) into the small
table bench.py reads (profiles/pmc_traffic.json):  python tools/pmc_traffic.py pmc_summary_256.json pmc_summary_512.json
"""
import json
import re
import sys


def main():
    out = {}
    for path in sys.argv[1:]:
        size = int(re.search(r"(\d+)\.json$", path).group(1))
        data = json.load(open(path))["kernels"]
        for mode in ("cg_update", "cg_update_r", "cg_update_x2", "cg_matvec_dot", "cg_residual"):
            # the launch plan the solve ran is the variant with the most launches (the first-call autotune also times a few launches of
            # every other candidate)
            cands = [(e.get("launches", 0), key, e) for key, e in data.items()
                     if key.startswith(mode + "<") and "read_bytes_prescribed" in e and "write_bytes_prescribed" in e]
            if cands:
                _, key, e = max(cands, key=lambda c: c[0])
                out[f"{mode}_{size}"] = int(round(e["read_bytes_prescribed"] + e["write_bytes_prescribed"]))
                out[f"{mode}_{size}_kernel"] = key
    for size in sorted({int(k.rsplit("_", 1)[1]) for k in out if re.search(r"_\d+$", k)}):
        # the default 'CG' alternates UPDATE_R (r only) and UPDATE_X2 (x for two steps + r): bench.py's "cg_update" is their mean
        if f"cg_update_r_{size}" in out and f"cg_update_x2_{size}" in out:
            out[f"cg_update_{size}"] = int(round(0.5 * (out[f"cg_update_r_{size}"] + out[f"cg_update_x2_{size}"])))
            out[f"cg_update_{size}_kernel"] = "mean of " + out[f"cg_update_r_{size}_kernel"] + " and " + out[f"cg_update_x2_{size}_kernel"]
    out["_note"] = ("HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, one process per size), "
                    "FETCH_SIZE x 2048 B, WRITE_SIZE x 1024 B (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 1/2 of a wide streaming "
                    "read); calibrated against a 512 MiB streaming copy in the same run (profiles/r02_pmc_summary_<size>.json)")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
