#!/bin/bash
# r05e: conv with the (band, strip) pairs dealt to the lanes in one running number; exhaustive RTZ conversion check; rehearsal tests
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
{
echo "=== cvt_pk_u8_f32 under round-toward-zero, all floats"; timeout 120 tools/ubench/cvt_rtz | tail -3
echo "=== filters parity"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_ref_f1f2_gpu.py tests/test_diff_pack_gpu.py -x -q 2>&1 | tail -5
for v in quad lin quad lin; do
  echo "--- filters $v"; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 200 tools/diffbench --filters --batch 192 --steps 5 2>&1 | grep -E "conv3x3|config 4" | cut -c1-200
done
echo "=== rehearsal tests"; timeout -k 10 800 python -m pytest tests/test_rehearsal_gpu.py -x -v 2>&1 | grep -E "PASS|FAIL|ERROR|passed|failed|Error" 
} > $O/log.txt 2>&1
tail -40 $O/log.txt
