#!/bin/bash
# r06b: what distinguishes output arrays on which the dense expansion is fast (202 us per 32 S0 pairs) from those on which it
# is slow (265): tools/diffbench --place with probes on every pair of arrays (streaming stores, the expansion's own store
# shape, random lines), the arrays re-used, mixed, freed and re-drawn, contiguous allocations, and ballast in front
# (DIFFBENCH_BURN_GB); the C++ drop-in's first-frame latency after MI355_PREPARE_EXEC.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_diff_pack_gpu.py tests/test_tools_gpu.py tests/test_server_hip_gpu.py tests/test_pipe_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -2 $O/pytest.log | tee -a $O/summary.txt
for i in 1 2 3; do timeout -k 10 60 tools/compat_pipe 1920 1080 12 2>&1 | tail -1 | tee -a $O/summary.txt; done
for i in 1 2; do
  timeout -k 10 200 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --place 24 > $O/place$i.log 2>&1 || echo "place $i failed" | tee -a $O/summary.txt
done
DIFFBENCH_BURN_GB=24 timeout -k 10 200 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --place 6 > $O/place_burn24.log 2>&1 || echo "place burn failed" | tee -a $O/summary.txt
grep -h "hipMalloc\|again\|mixed\|after_free\|contiguous\|burned" $O/place1.log | grep -v "^probe" | cut -c1-80 | tee -a $O/summary.txt
