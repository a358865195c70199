#!/bin/bash
# log layout: frame t's units in chunk t / kChunkFrames of the tile (no room checks in the pack kernel) against round 3's densely packed KiB chunks (old)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04am
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=2 bash tools/exp/run_matrix.sh \
 "old pipelined|old||" \
 "cf16 pipelined|cf16||" \
 "cf8 pipelined|cf8||" \
 "cf32 pipelined|cf32||" \
 "cf4 pipelined|cf4||" \
 "old sequential|old|MI355_PIPELINE=0|" \
 "cf16 sequential|cf16|MI355_PIPELINE=0|" \
 "cf8 sequential|cf8|MI355_PIPELINE=0|" \
 "cf32 sequential|cf32|MI355_PIPELINE=0|"
done
REPS=1 bash tools/exp/run_matrix.sh \
 "old pairs|old||--pairs --batch 128" "cf16 pairs|cf16||--pairs --batch 128" \
 "old s0|old||--regime s0 --batch 32" "cf16 s0|cf16||--regime s0 --batch 32" \
 "old flip|old||--regime flip --batch 32" "cf16 flip|cf16||--regime flip --batch 32" \
 "old 4k|old||--width 3840 --height 2160 --batch 64" "cf16 4k|cf16||--width 3840 --height 2160 --batch 64"
echo "== gpu suite (in-tree = cf16)"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== soak"; timeout -k 10 400 python tests/soak.py 3000 2>&1 | tail -2
} > gpurun_out/r04am/log.txt 2>&1
cat gpurun_out/r04am/log.txt
