#!/bin/bash
# pack grid of the pipelined batch (two launches share it) on the final library: workgroups in total
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bf
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "1024 (default)|-||" "768|-|MI355_K1_BLOCKS=768|" "896|-|MI355_K1_BLOCKS=896|" "1152|-|MI355_K1_BLOCKS=1152|" "1280|-|MI355_K1_BLOCKS=1280|" "1519 (a tile per wave)|-|MI355_K1_BLOCKS=1519|" \
 "split 55|-|MI355_SPLIT=55|" "split 60|-|MI355_SPLIT=60|" "split 45|-|MI355_SPLIT=45|"
done
} > gpurun_out/r04bf/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04bf/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-24s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
