#!/bin/bash
# (1) pack kernel split in 2 / 3 / 4 launches; (2) histogram LDS layout: copies of a bin spread over the banks
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ah
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=2 bash tools/exp/run_matrix.sh \
 "2 launches|-||" \
 "3 launches|-|MI355_PARTS=3|" \
 "4 launches|-|MI355_PARTS=4|" \
 "4 launches, K1 1536|-|MI355_PARTS=4 MI355_K1_BLOCKS=1536|" \
 "3 launches, K1 1536|-|MI355_PARTS=3 MI355_K1_BLOCKS=1536|"
done
for v in hold h8 h16 h32 h64 hold h32; do
  echo -n "hist $v: "; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py 2>/dev/null | grep "binarize\|config 3" | cut -c1-150 | tr '\n' ' '; echo
done
echo "== filter tests (in-tree = h32)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_server_hip_gpu.py -x -q 2>&1 | tail -3
} > gpurun_out/r04ah/log.txt 2>&1
cat gpurun_out/r04ah/log.txt
