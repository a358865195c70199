#!/bin/bash
# non-temporal output stores (xnt) against the product (base) at smaller batches: does the log stay in the Infinity Cache?
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ao
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "base 256|base||" "xnt 256|xnt||" \
 "base 128|base||--batch 128" "xnt 128|xnt||--batch 128" \
 "base 96|base||--batch 96" "xnt 96|xnt||--batch 96" \
 "base 64|base||--batch 64" "xnt 64|xnt||--batch 64" \
 "base 48|base||--batch 48" "xnt 48|xnt||--batch 48" \
 "base 32|base||--batch 32" "xnt 32|xnt||--batch 32" \
 "base seq 128|base|MI355_PIPELINE=0|--batch 128" "xnt seq 128|xnt|MI355_PIPELINE=0|--batch 128" \
 "base seq 64|base|MI355_PIPELINE=0|--batch 64" "xnt seq 64|xnt|MI355_PIPELINE=0|--batch 64"
done
} > gpurun_out/r04ao/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04ao/log.txt'):
    m=re.match(r'(.*?): digest \w+ (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(2)); print("%-16s batch %3d  %.4f ms/step  %.3f us/frame  frac %.4f  kernels %s"%(m.group(1),d['batch'],d['ms_per_step'],d['ms_per_step']*1e3/d['batch'],d['frac'],d['kernels_us']))
PY
