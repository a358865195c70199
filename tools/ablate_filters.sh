#!/bin/bash
# A/B builds of libmi355diff for the filter kernels: every argument is "name:compiler flags"; each variant
# is built into gpurun_out/ablate/<name>/ and timed with tools/bench_filters.py (the variant is copied over the in-tree library of the GPU box's scratch
# copy of the repository; the last line rebuilds the product).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  d=gpurun_out/ablate/$name; mkdir -p $d
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off $flags -shared \
      -o $d/libmi355diff.so cudavideostream_amd/csrc/core.hip cudavideostream_amd/csrc/diff_pack.hip cudavideostream_amd/csrc/filters.hip cudavideostream_amd/csrc/stream_ops.hip
  cp $d/libmi355diff.so cudavideostream_amd/libmi355diff.so
  echo -n "$name: "; timeout -k 5 200 python3 tools/bench_filters.py ${FILTER_ARGS:-} 2>/dev/null | grep "${FILTER_GREP:-conv}" | cut -c1-120
done
touch cudavideostream_amd/csrc/core.hip; make -C cudavideostream_amd/csrc -s
