#!/bin/bash
# Round-3 GPU session O: the N > 1 bookkeeping of bench.py over RCCL on one GPU (PHIHIP_BENCH_FORCE_DIST), the driver's torchrun command line
# with one rank, configs 3 / 5 with the re-based byte model.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
O=gpurun_out/r3o; mkdir -p $O
export TMPDIR=/tmp
PHIHIP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 5 --warmup 2 --cpu-size 0 --config3-size 0 --pmc 0 --profile-steps 0 > $O/bench_forced_dist.json 2> $O/forced.err; echo "forced dist rc=$?"
python - <<PY
import json
d=json.load(open('$O/bench_forced_dist.json')); print(d['value'], d['ms_per_step'], json.dumps(d['replicas'])[:600])
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-size 0 --config3-size 0 --pmc 0 > $O/bench_torchrun1.json 2> $O/torchrun.err; echo "torchrun rc=$?"; head -c 300 $O/bench_torchrun1.json; echo
PHIHIP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --workload config4 --steps 10 --warmup 3 --cg-iters 50 > $O/bench_config4_forced.json 2>> $O/forced.err; echo "config4 forced rc=$?"; head -c 300 $O/bench_config4_forced.json; echo
PHIHIP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29566 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --workload slab --size 256 --steps 3 --warmup 1 > $O/bench_slab_forced.json 2>> $O/forced.err; echo "slab forced rc=$?"; head -c 300 $O/bench_slab_forced.json; echo
timeout 600 python tools/bench_configs.py 3 5 > $O/configs35.jsonl 2> $O/configs.err; cut -c1-330 $O/configs35.jsonl
tail -3 $O/forced.err
