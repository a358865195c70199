#!/bin/bash
# r05d: the expander's quad path (four tiles per round where they fit) against the pair-only expander and round 4's library
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
{
echo "=== rehearsal, direct (2 ranks on the one GPU)"
timeout -k 10 240 python bench.py --gpus 2 --rehearse-on-one-gpu --width 640 --height 360 --batch 8 --steps 3 --warmup 2 --gather-every-steps 2 --no-cpu --no-pair --no-filters --no-host-path --no-config5 > $O/rehearse.json 2> $O/rehearse.err; echo "rc=$?"; tail -c 1500 $O/rehearse.err; head -c 600 $O/rehearse.json; echo
echo "=== filters parity (conv: store fix, RTZ conversion)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -5
for v in new quad new quad; do
  echo "--- filters $v"; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 200 tools/diffbench --filters --batch 192 --steps 5 2>&1 | grep -E "conv3x3|config 4" | cut -c1-200
done
echo "=== parity of the diff path (quad library = the product)"; timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_fuzz_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py -x -q 2>&1 | tail -5
for v in r04 new quad r04 new quad; do
  echo -n "--- stream $v: "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 120 tools/diffbench --steps 30 --digest 2>&1 | tr '\n' ' '; echo
done
for v in new quad; do
  echo -n "--- sequential $v: "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 120 tools/diffbench --steps 30 --opt 1=0 2>&1 | tr '\n' ' '; echo
  echo -n "--- pairs $v: "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 120 tools/diffbench --steps 20 --apart --batch 128 --digest 2>&1 | tr '\n' ' '; echo
  echo -n "--- 4K $v: "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 120 tools/diffbench --steps 20 --width 3840 --height 2160 --batch 64 --digest 2>&1 | tr '\n' ' '; echo
  echo -n "--- s0 $v: "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 120 tools/diffbench --steps 20 --regime s0 --batch 32 --digest 2>&1 | tr '\n' ' '; echo
done
} > $O/log.txt 2>&1
tail -40 $O/log.txt
