#!/bin/bash
# Ablation of the pack kernel on the GPU box: builds libmi355diff variants with -DMI355_ABLATE=k into
# gpurun_out/ablate/k/ and times each with tools/diffbench (outputs of k>0 are wrong by design; only the
# kernel time matters).  Usage: bash tools/ablate.sh "0 1 2 3" [extra diffbench args]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for k in $1; do
  d=gpurun_out/ablate/$k; mkdir -p $d
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -DMI355_ABLATE=$k -shared \
      -o $d/libmi355diff.so cudavideostream_amd/csrc/core.hip cudavideostream_amd/csrc/diff_pack.hip cudavideostream_amd/csrc/filters.hip
  for rep in 1 2; do
    echo -n "ablate=$k stream: "; LD_LIBRARY_PATH=$d timeout -k 5 120 tools/diffbench --steps 20 ${@:2}
  done
  echo -n "ablate=$k pairs:  "; LD_LIBRARY_PATH=$d timeout -k 5 120 tools/diffbench --steps 20 --pairs ${@:2}
done
