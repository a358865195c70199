#!/bin/bash
# the expander beside the pack kernel on fewer wave slots: dynamic LDS padding per workgroup (5 KB static: 32 per CU; +5 KB: 16; +15 KB: 8; +35 KB: 4)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bb
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "pad 0|-||" "pad 5K|-|MI355_XLDS_PAD=5120|" "pad 10K|-|MI355_XLDS_PAD=10240|" "pad 15K|-|MI355_XLDS_PAD=15360|" "pad 25K|-|MI355_XLDS_PAD=25600|" "pad 35K|-|MI355_XLDS_PAD=35840|" \
 "pad 15K, K1 1536|-|MI355_XLDS_PAD=15360 MI355_K1_BLOCKS=1536|" "pad 35K, K1 full|-|MI355_XLDS_PAD=35840 MI355_K1_BLOCKS=1519|"
done
} > gpurun_out/r04bb/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04bb/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-18s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
