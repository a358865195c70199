#!/bin/bash
# adaptive overlap: own-stream batches run one after the other while the latest batch total that has arrived says the input is dense (MI355_DENSE_PCT, default 40; 0 = always overlap)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bh
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "stream, adaptive|-||" "stream, always overlap|-|MI355_DENSE_PCT=0|" \
 "s0, adaptive|-||--regime s0 --batch 32" "s0, always overlap|-|MI355_DENSE_PCT=0|--regime s0 --batch 32" "s0, never overlap|-|MI355_PIPELINE=0|--regime s0 --batch 32" \
 "flip, adaptive|-||--regime flip --batch 32" "flip, always overlap|-|MI355_DENSE_PCT=0|--regime flip --batch 32" "flip, never overlap|-|MI355_PIPELINE=0|--regime flip --batch 32" \
 "apart, adaptive|-||--apart --batch 128" "apart, always overlap|-|MI355_DENSE_PCT=0|--apart --batch 128"
done
echo "== diff tests"; timeout -k 10 600 python -m pytest tests/test_diff_pack_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2
echo "== chain soak"; timeout -k 10 300 python tests/soak_chain.py 300 9 2>&1 | tail -1
} > gpurun_out/r04bh/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04bh/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-24s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
