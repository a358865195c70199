#!/bin/bash
# r05b: conv rewrite (prefetch depth 3, DPP halos, one conversion per byte) at 4 and 2 waves per SIMD against round 4's
# library on the same box; round 4's library against the cleaned one on the headline and on two streams; cvt RTZ probe.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
{
echo "=== cvt_pk_u8_f32 under round-toward-zero"; timeout 60 tools/ubench/cvt_rtz
echo "=== conv parity (new library)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -k "conv or filter or fuzz or config" 2>&1 | tail -5
for v in r04 new conv2 r04 new conv2; do
  echo "--- filters $v"; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 200 tools/diffbench --filters --batch 192 --steps 5 2>&1 | grep -E "conv3x3|config 4|config 3" | cut -c1-260
done
for v in r04 new r04 new; do
  echo -n "--- stream $v: "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 120 tools/diffbench --steps 30 --digest 2>&1 | tr '\n' ' '; echo
  echo -n "--- 2 cores $v: "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 120 tools/diffbench --steps 30 --cores 2 2>&1 | tr '\n' ' '; echo
done
} > $O/log.txt 2>&1
tail -60 $O/log.txt
