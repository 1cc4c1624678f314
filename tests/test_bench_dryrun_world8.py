"""
The driver's own N = 8 command line, rehearsed on CPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
--master-addr 127.0.0.1 --master-port P <bench> --gpus 8 --steps K --warmup W` with bench.py's machine hooks replaced by
tests/bench_dryrun.py (gloo, emulation library, LOCAL_RANK -> device mapping recorded). Asserts the single JSON line of rank 0, the
contract's keys, one verified replica per rank with identical bits, 8 distinct device bindings -- for the replica mode (the BASELINE metric)
and for the sharded batch (`--workload config4`: 8 entries, one per rank). No multi-GPU curve is measured by this: the line says so.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(emu_library, tmp_path, world, extra):
    env = dict(os.environ, PHIHIP_EMU_LIB=emu_library.path, PHIHIP_DRYRUN_MAP_DIR=str(tmp_path), OMP_NUM_THREADS="1", PHIHIP_AUTOTUNE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "tests", "bench_dryrun.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly ONE line (rank 0's record), got {len(lines)}: {r.stdout[:500]}"
    rec = json.loads(lines[0])
    maps = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]
    assert sorted(m["would_bind"] for m in maps) == sorted(f"cuda:{k}" for k in range(world))          # one distinct device per rank
    assert all(m["world"] == world and m["local_rank"] == m["rank"] for m in maps)
    return rec


@pytest.mark.parametrize("world", [8])
def test_driver_command_line_replicas_world8(emu_library, tmp_path, world):
    rec = _run(emu_library, tmp_path, world, ["--size", "16", "--cg-iters", "10", "--config3-size", "0", "--profile-steps", "0", "--cpu-size", "0", "--pmc", "0"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in rec, key
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert rec["value"] > 0 and abs(rec["value"] - 16 ** 3 * world * 2 / (rec["ms_per_step"] * 2e-3)) <= 1e-6 * rec["value"]      # whole-job aggregate
    assert rec["iterations_verified"] == [[10]] * world
    rp = rec["replicas"]
    assert rp["all_bit_identical"] and rp["verified_ok"] == [True] * world and len(rp["bit_identical_to_rank0"]) == world
    assert set(rp["pinned_launch_plans"]) == {"family0", "family1", "family2", "family3"}
    assert "no" in rec["scaling_measured"] and "dry_run" in rec


def test_driver_command_line_sharded_batch_world8(emu_library, tmp_path):
    world = 8
    rec = _run(emu_library, tmp_path, world, ["--workload", "config4", "--size", "32", "--batch-total", "8", "--cg-iters", "8"])
    assert rec["n_gpus"] == world and rec["scaling"] == "strong"
    sh = rec["shards"]
    assert sh["entries_per_rank"] == [1] * world and sh["iterations_per_rank"] == [[8]] * world and sh["verified_ok"] == [True] * world
    assert sh["entry0_all_bit_identical"]
    assert len({tuple(c) for c in sh["owned_checksums"]}) == world          # per-entry inflow positions: every rank holds a different simulation


def test_phi_level_block_runs_on_the_emulation(emu_library, emu_ctx, monkeypatch):
    """ bench.py's `phi_level` block (the drop-in path timed next to the C-ABI loop) executes end to end -- tiny sizes, the kernel sources under the
    emulation, device synchronisation replaced: the bookkeeping of the block, not a measurement """
    import torch
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "device_sync", lambda device: None)
    rec = bench.phi_level_block(emu_ctx, emu_library, torch.device("cpu"), cg_iters=4, steps=2, sizes=(8,), plume_n=16, plume_steps=2)
    assert set(rec) == {"taylor_green_8", "smoke_plume_16x16", "note"}
    for k in ("taylor_green_8", "smoke_plume_16x16"):
        assert rec[k]["ms_c_abi"] > 0 and rec[k]["ms_phi_level"] > 0

