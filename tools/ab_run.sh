#!/bin/bash
# Times the variants tools/ab_build.sh made (build/ab/<name>/libmi355diff.so) with tools/diffbench on the GPU box:
#   bash tools/ab_run.sh base k1v1 nostore            (DIFFBENCH_ARGS="--batch 64", REPS=2, PAIRS=1 optional)
# Prints per variant the diffbench line(s) and the digest of the outputs (variants that are meant to be
# bit-exact must print the same digest; MI355_ABLATE builds are wrong by design).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for name in "$@"; do
  d=build/ab/$name
  [ -f $d/libmi355diff.so ] || { echo "$name: not built"; continue; }
  for rep in $(seq 1 ${REPS:-2}); do
    echo -n "$name stream: "; LD_LIBRARY_PATH=$d timeout -k 5 120 tools/diffbench --steps 20 --digest ${DIFFBENCH_ARGS:-} 2>&1 | tr '\n' ' '; echo
  done
  [ -z "${PAIRS:-}" ] || { echo -n "$name pairs:  "; LD_LIBRARY_PATH=$d timeout -k 5 120 tools/diffbench --steps 20 --pairs --digest ${DIFFBENCH_ARGS:-} 2>&1 | tr '\n' ' '; echo; }
done
