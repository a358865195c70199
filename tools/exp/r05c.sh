#!/bin/bash
# r05c: conv with exact wait counts (3 rows in flight); rehearsal of the N > 1 job (process per rank); cvt RTZ probe; SQ counters of conv
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
{
echo "=== cvt_pk_u8_f32 under round-toward-zero"; timeout 60 tools/ubench/cvt_rtz
echo "=== conv parity (new library)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -5
for v in r04 new conv2 conv3 new conv2 conv3; do
  echo "--- filters $v"; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 200 tools/diffbench --filters --batch 192 --steps 5 2>&1 | grep -E "conv3x3|config 4" | cut -c1-200
done
echo "=== rehearsal tests"; timeout -k 10 900 python -m pytest tests/test_rehearsal_gpu.py -x -q 2>&1 | tail -30
echo "=== SQ counters, filters (conv)"; bash profiles/pmc_sq.sh r05c_filters --filters --batch 96 2>&1 | grep -A24 "conv3x3"
} > $O/log.txt 2>&1
tail -100 $O/log.txt
