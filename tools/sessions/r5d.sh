#!/bin/bash
# Round-5 GPU session D: the LDS-DMA fill of the tiled self-advection -- GPU parity (bit-identical to the register-staged kernel), A/B by the
# environment switch in the same library (two alternating rounds), SQ counters of the DMA kernel, the benchmark line
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5d}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "lds_dma or advection_matches" > $O/pytest_dma.log 2>&1; echo "dma rc=$?"; tail -5 $O/pytest_dma.log
for ROUND in 1 2; do for DMA in 0 1; do
  PHIHIP_ADVECT_DMA=$DMA timeout 300 python tools/time_frow.py --size 256 --dtype f32 --bc periodic --only advect_self,mac_cormack_self --reps 50 | sed "s/\"lib\": \"default\"/\"lib\": \"dma=$DMA\"/" >> $O/time_dma.jsonl 2>> $O/time_dma.err
  PHIHIP_ADVECT_DMA=$DMA timeout 300 python tools/time_frow.py --size 512 --dtype f32 --bc periodic --only advect_self --reps 10 | sed "s/\"lib\": \"default\"/\"lib\": \"dma=$DMA\"/" >> $O/time_dma.jsonl 2>> $O/time_dma.err
  PHIHIP_ADVECT_DMA=$DMA timeout 300 python tools/time_frow.py --size 256 --dtype f64 --bc periodic --only advect_self --reps 30 | sed "s/\"lib\": \"default\"/\"lib\": \"dma=$DMA\"/" >> $O/time_dma.jsonl 2>> $O/time_dma.err
done; done
python - <<PY
import json
for l in open('$O/time_dma.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
timeout 400 bash tools/prof_advect.sh "--size 256 --field tg" > $O/prof_advect.log 2>&1; cp gpurun_out/prof_adv_pmc.json $O/ 2>/dev/null; grep -E "dma_kernel|self_tile" $O/prof_advect.log | head -5
find gpurun_out/prof_adv_pmc gpurun_out/prof_adv_stats -name "*.csv" -size +1M -delete 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --pmc 0 --cpu-size 0 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; head -c 400 $O/bench_n1.json; echo
