#!/bin/bash
# dense waves of the expander: bytes placed by position in straight code (dw1) against the bit walk (dw0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bc
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "dw0 s0 seq|dw0|MI355_PIPELINE=0|--regime s0 --batch 32" "dw1 s0 seq|dw1|MI355_PIPELINE=0|--regime s0 --batch 32" \
 "dw0 s0|dw0||--regime s0 --batch 32" "dw1 s0|dw1||--regime s0 --batch 32" \
 "dw0 flip seq|dw0|MI355_PIPELINE=0|--regime flip --batch 32" "dw1 flip seq|dw1|MI355_PIPELINE=0|--regime flip --batch 32" \
 "dw0 stream|dw0||" "dw1 stream|dw1||" \
 "dw0 apart|dw0||--apart --batch 128" "dw1 apart|dw1||--apart --batch 128"
done
echo "== expander tests (in-tree = dw1)"; timeout -k 10 600 python -m pytest tests/test_diff_pack_gpu.py tests/test_fuzz_gpu.py tests/test_ref_f1f2_gpu.py tests/test_stream_ops_gpu.py -x -q 2>&1 | tail -3
} > gpurun_out/r04bc/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04bc/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-14s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
