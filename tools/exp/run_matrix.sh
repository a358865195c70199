#!/bin/bash
# Runs a matrix of diffbench configurations on the GPU box: every argument is "label|variant|ENV=.. ENV=..|diffbench args"
# (variant = a build/ab/<variant> library made by tools/ab_build.sh, or "-" for the in-tree one).  REPS (default 2).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for spec in "$@"; do
  IFS='|' read -r label variant envs args <<< "$spec"
  libdir=cudavideostream_amd; [ "$variant" != "-" ] && libdir=build/ab/$variant
  [ -f $libdir/libmi355diff.so ] || { echo "$label: $libdir not built"; continue; }
  for rep in $(seq 1 ${REPS:-2}); do
    echo -n "$label: "
    env LD_LIBRARY_PATH=$libdir $envs timeout -k 5 120 tools/diffbench --steps 30 --digest $args 2>&1 | tr '\n' ' ' | sed -e 's/"harness": "diffbench", //' -e 's/"width": [0-9]*, "height": [0-9]*, //' -e 's/, "workspace_bytes": [0-9]*//'
    echo
  done
done
