#!/bin/bash
# pack kernel log stores non-temporal (k1 codes, k2 records, k4 meta, k7 all) against the product (k0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ar
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "k0|k0||" "k1|k1||" "k2|k2||" "k4|k4||" "k7|k7||" \
 "k0 seq|k0|MI355_PIPELINE=0|" "k1 seq|k1|MI355_PIPELINE=0|" "k2 seq|k2|MI355_PIPELINE=0|" "k4 seq|k4|MI355_PIPELINE=0|" "k7 seq|k7|MI355_PIPELINE=0|"
done
} > gpurun_out/r04ar/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04ar/log.txt'):
    m=re.match(r'(.*?): digest \w+ (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(2)); print("%-10s %-6s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),d['mode'],d['ms_per_step'],d['frac'],d['kernels_us']))
PY
