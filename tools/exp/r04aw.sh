#!/bin/bash
# pair mode picks non-temporal loads when its operands share no frame: detection check (forced off / automatic) and the suite
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04aw
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "consecutive pairs, automatic|-||--pairs --batch 128" "consecutive pairs, forced once|-|MI355_PAIR_ONCE=1|--pairs --batch 128" \
 "disjoint pairs, automatic|-||--apart --batch 128" "disjoint pairs, forced plain|-|MI355_PAIR_ONCE=0|--apart --batch 128" \
 "4K disjoint pairs, automatic|-||--apart --width 3840 --height 2160 --batch 64" "4K disjoint pairs, forced plain|-|MI355_PAIR_ONCE=0|--apart --width 3840 --height 2160 --batch 64"
echo "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== soak"; timeout -k 10 400 python tests/soak.py 3000 2>&1 | tail -2
} > gpurun_out/r04aw/log.txt 2>&1
cut -c1-260 gpurun_out/r04aw/log.txt
