#!/bin/bash
# pair mode: non-temporal loads of cur (p1), prev (p2), both (p3); operands that do not overlap (two halves of a stream) and 4K pairs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04av
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "p0 pairs|p0||--pairs --batch 128" "p3 pairs|p3||--pairs --batch 128" "p0 apart|p0||--apart --batch 128" "p1 apart|p1||--apart --batch 128" "p2 apart|p2||--apart --batch 128" "p3 apart|p3||--apart --batch 128" \
 "p0 apart seq|p0|MI355_PIPELINE=0|--apart --batch 128" "p3 apart seq|p3|MI355_PIPELINE=0|--apart --batch 128" \
 "p0 4k apart|p0||--apart --width 3840 --height 2160 --batch 64" "p3 4k apart|p3||--apart --width 3840 --height 2160 --batch 64" \
 "p0 s0|p0||--regime s0 --batch 32" "p3 s0|p3||--regime s0 --batch 32" \
 "p0 flip|p0||--regime flip --batch 32" "p3 flip|p3||--regime flip --batch 32"
done
} > gpurun_out/r04av/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04av/log.txt'):
    m=re.match(r'(.*?): digest \w+ (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(2)); print("%-14s %-6s %.4f ms/step  frac %.4f  kernels %s  P/frame %d"%(m.group(1),d['mode'],d['ms_per_step'],d['frac'],d['kernels_us'],d['changed_bytes_per_frame']))
PY
