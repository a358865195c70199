#!/bin/bash
# cache policy bits of the pack kernel's frame loads: a2 = nt (product), a3 = sc0 nt, a18 = sc1 nt, a19 = sc0 sc1 nt, a17 = sc0 sc1, a16 = sc1, a1 = sc0
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ax
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "a2|a2||" "a3|a3||" "a18|a18||" "a19|a19||" "a17|a17||" "a16|a16||" "a1|a1||" \
 "a2 seq|a2|MI355_PIPELINE=0|" "a3 seq|a3|MI355_PIPELINE=0|" "a18 seq|a18|MI355_PIPELINE=0|" "a19 seq|a19|MI355_PIPELINE=0|" "a17 seq|a17|MI355_PIPELINE=0|"
done
} > gpurun_out/r04ax/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04ax/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-10s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
