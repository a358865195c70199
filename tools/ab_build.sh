#!/bin/bash
# A/B builds of libmi355diff made in the BUILD container (hipcc cross-compiles gfx950 without a GPU), so that the
# GPU box only has to run them: every argument is "name:compiler flags"; variant <name> ends up in
# build/ab/<name>/libmi355diff.so (build/ is git-ignored but travels with the gpurun snapshot).
#   bash tools/ab_build.sh "base:" "nostore:-DMI355_ABLATE=1" "pad8:-DMI355_PAD=8" "xrounds:-DMI355_XABLATE=3"
# Every variant is compiled with -DMI355_LAB=1 (csrc/lab.h: the only switches the kernels have); source-level
# experiments are variants of the SOURCE (AB_SRC=dir: take the .hip files from there), not of preprocessor flags.
# Only the files named in AB_FILES (default: diff_pack.hip) are recompiled per variant; the other objects are
# built once.  tools/ab_run.sh times the variants with tools/diffbench on the GPU box.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/cudavideostream_amd/csrc
OUT=$ROOT/build/ab
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I$SRC"
ALL="core diff_pack filters stream_ops group diag"
VAR_FILES=${AB_FILES:-diff_pack}
mkdir -p $OUT/common
pids=()
for f in $ALL; do
  case " $VAR_FILES " in *" $f "*) continue;; esac
  stale=0   # (every header counts: filters.hip includes the GENERATED median_net.h, everything includes lab.h and the C-ABI)
  for h in $SRC/*.h $ROOT/include/mi355diff.h $SRC/$f.hip; do [ $h -nt $OUT/common/$f.o ] && stale=1; done
  if [ ! -f $OUT/common/$f.o ] || [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc $FLAGS -c -o $OUT/common/$f.o $SRC/$f.hip & pids+=($!)
  fi
done
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p $OUT/$name
  for f in $VAR_FILES; do
    src=$SRC/$f.hip
    case "$flags" in *AB_SRC=*) d=${flags##*AB_SRC=}; d=${d%% *}; [ -f $d/$f.hip ] && src=$d/$f.hip; flags=${flags/AB_SRC=$d/};; esac
    /opt/rocm/bin/hipcc $FLAGS -DMI355_LAB=1 $flags -c -o $OUT/$name/$f.o $src & pids+=($!)
    # keep at most 8 compilers running
    while [ $(jobs -rp | wc -l) -ge 8 ]; do wait -n; done
  done
done
for p in "${pids[@]}"; do wait $p; done
for v in "$@"; do
  name=${v%%:*}
  objs=""
  for f in $ALL; do
    case " $VAR_FILES " in *" $f "*) objs="$objs $OUT/$name/$f.o";; *) objs="$objs $OUT/common/$f.o";; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$name/libmi355diff.so $objs -ldl
  echo "built $name"
done
