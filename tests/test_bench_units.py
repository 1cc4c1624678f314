"""
bench.py pieces that need no device: the roofline block's traffic cross-check compares the PMC ratio of this invocation with the committed per-kernel table only
where both were measured with the same launch plan (round 5: the first-call autotune settles on one of three MATVEC plans at 512^3, and halo re-reads belong to the plan).
"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.argv = argv


def test_traffic_cross_check_names_both_plans():
    b = _bench()
    name = next(n for n in ("r06_kernel_roofline.json", "r05_kernel_roofline.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))      # (bench.py's order)
    with open(os.path.join(ROOT, "profiles", name)) as f:
        table = json.load(f)
    # r6: the block carries no traffic unless the PMC passes ran in the invocation (no committed fallback file any more): stand in for them here
    b.live_pmc_traffic = lambda n, kernel_key, plans=None: (int(1.01 * 3 * 4 * n ** 3), "unit test")
    grp = next(g for g in table["groups"] if g["group"] == "f32_512")
    table_plan = [int(grp["pinned_plans"]["1"][k]) for k in ("rows", "tpr", "chunk")]
    per = {"cg_matvec_dot": (0.30, 100, 30.0), "cg_update": (0.5, 50, 25.0), "cg_update_r": (0.29, 50, 14.5)}
    same = {"0": [4, 64, 64], "1": table_plan, "2": [4, 64, 64], "3": [1, 64, 128]}
    other = dict(same, **{"1": [table_plan[0] * 2, table_plan[1], table_plan[2]]})
    blk, it = b.roofline_block(512, per, 1, 1, False, "unit test", plans=same)
    chk = blk["traffic_cross_check"]
    assert chk["same_launch_plan"] is True and chk["plan_here"] == table_plan == chk["plan_table"] and isinstance(chk["agree_within_3_percent"], bool)
    blk2, _ = b.roofline_block(512, per, 1, 1, False, "unit test", plans=other)
    chk2 = blk2["traffic_cross_check"]
    assert chk2["same_launch_plan"] is False and chk2["agree_within_3_percent"] is None and chk2["plan_table"] == table_plan
    # the block itself: bytes moved by construction over the launch time, against 8 TB/s
    assert blk["bound"] == "hbm" and abs(blk["frac"] - 3 * 4 * 512 ** 3 / 0.30e-3 / 1e9 / 8000.0) < 1e-3
    blk3, _ = b.roofline_block(512, per, 0, 1, False, "unit test", plans=same)
    assert blk3["traffic"] is None and "traffic_cross_check" not in blk3          # not measured in the invocation = null, never a committed file's figure
    assert abs(it["ms_iteration"] - (0.30 + 0.5 * (0.5 + 0.29))) < 1e-9
