#!/bin/bash
# A/B builds of libmi355diff on the GPU box (never shipped): every argument is "name:compiler flags";
# each variant is compiled into gpurun_out/ablate/<name>/ and timed with tools/diffbench.
#   bash tools/ablate.sh "base:" "nostore:-DMI355_ABLATE=1" "pad8:-DMI355_PAD=8"   (-DMI355_LAB=1 is added: csrc/lab.h)
# DIFFBENCH_ARGS adds harness arguments (e.g. "--batch 64").  Outputs of MI355_ABLATE>0 builds are
# wrong by design; only the kernel time matters.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  d=gpurun_out/ablate/$name; mkdir -p $d
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -DMI355_LAB=1 $flags -shared \
      -o $d/libmi355diff.so cudavideostream_amd/csrc/core.hip cudavideostream_amd/csrc/diff_pack.hip cudavideostream_amd/csrc/filters.hip cudavideostream_amd/csrc/stream_ops.hip cudavideostream_amd/csrc/group.hip cudavideostream_amd/csrc/diag.hip -ldl
  [ -n "${NO_STREAM:-}" ] || for rep in 1 2; do
    echo -n "$name stream: "; LD_LIBRARY_PATH=$d timeout -k 5 120 tools/diffbench --steps 20 ${DIFFBENCH_ARGS:-}
  done
  [ -z "${FILTERS:-}" ] || { echo "$name filters:"; LD_LIBRARY_PATH=$d timeout -k 5 200 tools/diffbench --filters --batch 96 --steps 5 | grep "${FILTERS}" | cut -c1-170; }
  [ -n "${NO_PAIRS:-}" ] || { echo -n "$name pairs:  "; LD_LIBRARY_PATH=$d timeout -k 5 120 tools/diffbench --steps 20 --pairs ${DIFFBENCH_ARGS:-}; }
done
