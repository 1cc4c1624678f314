#!/bin/bash
# Round-5 GPU session K: the GEN instantiation of the LDS-DMA self-advection (closed / open boxes: constants table, patch elements) -- GPU parity
# (bit-identical to the register-staged kernel), a fuzz batch, A/B by the environment switch in the same library (two alternating rounds)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"
O=gpurun_out/${SESSION_TAG:-r5k}; mkdir -p $O; export TMPDIR=/tmp
python -c "from phiflow_amd import _capi as C; l=C.load_default_library(); print('build', l.build_id(), 'tree', l.built_from_tree())" > $O/build_id.txt 2>&1; cat $O/build_id.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "lds_dma or advection or advect or mac_cormack or tile_config" > $O/pytest_dma.log 2>&1; echo "dma rc=$?"; tail -5 $O/pytest_dma.log
timeout 600 python tests/fuzz_parity.py --first 0 --count 60 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; grep -E "ERROR|FAIL|fails|SUMMARY" $O/fuzz.log | tail -5
for ROUND in 1 2; do for DMA in 0 1; do
  for SPEC in "256 f32 closed 50" "256 f32 open 50" "384 f64 closed 15" "256 f64 closed 30" "512 f32 closed 10"; do
    set -- $SPEC
    PHIHIP_ADVECT_DMA=$DMA timeout 300 python tools/time_frow.py --size $1 --dtype $2 --bc $3 --only advect_self,mac_cormack_self --reps $4 | sed "s/\"lib\": \"default\"/\"lib\": \"dma=$DMA\"/" >> $O/time_dma.jsonl 2>> $O/time_dma.err
  done
done; done
python - <<PY
import json
for l in open('$O/time_dma.jsonl'):
    d=json.loads(l)
    print(d['lib'][:16].ljust(16), d['size'], d['dtype'], d['bc'], ' '.join(f"{k}={v.get('ms','ERR')}" for k,v in d['kernels'].items()), d.get('advect_fallback'))
PY
for ROUND in 1 2; do for DMA in 0 1; do
  PHIHIP_ADVECT_DMA=$DMA timeout 600 python bench.py --workload smoke256 --steps 20 --warmup 5 --pmc 0 --cpu-size 0 --phi-level 0 > $O/smoke256_dma${DMA}_$ROUND.json 2> $O/smoke256_dma${DMA}_$ROUND.err; echo "smoke256 dma=$DMA rc=$?"
  python -c "import json;d=json.loads(open('$O/smoke256_dma${DMA}_$ROUND.json').read().strip().splitlines()[-1]);print('dma=$DMA', d['ms_per_step'])"
done; done
echo finished
