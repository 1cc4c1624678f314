"""
TEST INFRASTRUCTURE -- rehearsal of the driver's multi-GPU bench command line on a host WITHOUT GPUs:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P tests/bench_dryrun.py --gpus 8 ...

runs bench.main() unchanged except for the four machine hooks bench.py names (device_for_rank, DIST_BACKEND, load_library, device_sync): the
LOCAL_RANK -> device mapping is recorded instead of taken (every rank computes on the CPU), the process group is gloo, the library is the
kernel sources compiled against the fiber emulation (tests/hipemu). What is rehearsed is everything AROUND the kernels that a first 8-rank run
could die on: rendezvous from the environment, launch-plan broadcast and pinning, the per-step all-reduce, max-over-ranks timing, the
verification step on every rank, the all-gather of bit checksums, rank 0's single JSON line. The record is tagged "dry_run" and is not a
measurement. bench.py itself has no CPU path (it raises without a HIP device).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PHIHIP_AUTOTUNE"] = "0"

import torch  # noqa: E402

import bench  # noqa: E402
from phiflow_amd import _capi as C  # noqa: E402

EMU_LIB = os.environ.get("PHIHIP_EMU_LIB", os.path.join(ROOT, "tests", "hipemu", "libphihip_emu.so"))
MAPPING = os.environ.get("PHIHIP_DRYRUN_MAP_DIR")


def device_for_rank(local_rank):
    if MAPPING:      # what the real run would have bound: cuda:<LOCAL_RANK>, one distinct device per rank of the node
        with open(os.path.join(MAPPING, f"rank{os.environ.get('RANK', '0')}.json"), "w") as f:
            json.dump({"rank": int(os.environ.get("RANK", "0")), "local_rank": local_rank, "world": int(os.environ.get("WORLD_SIZE", "1")),
                       "would_bind": f"cuda:{local_rank}", "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}, f)
    return torch.device("cpu")


bench.device_for_rank = device_for_rank
bench.DIST_BACKEND = "gloo"
bench.load_library = lambda: C.Library(EMU_LIB)
bench.device_sync = lambda device: None
_emit = bench.emit_record
bench.emit_record = lambda rec: _emit({**rec, "dry_run": "CPU emulation + gloo: bookkeeping rehearsal, NOT a measurement", "data": "dry run"})

if __name__ == "__main__":
    bench.main()
