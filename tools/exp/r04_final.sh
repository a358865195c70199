#!/bin/bash
# the round's last call: suites on the final library, then the evidence run
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r04final
{
echo "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== soak"; timeout -k 10 400 python tests/soak.py 3000 2>&1 | tail -2
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} > gpurun_out/r04final/log.txt 2>&1
cat gpurun_out/r04final/log.txt
bash tools/exp/r04_profiles.sh > /dev/null 2>&1
tail -3 gpurun_out/r04prof/log.txt | cut -c1-200
