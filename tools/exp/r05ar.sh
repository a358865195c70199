#!/bin/bash
# r05ar: is the slow start of a run a matter of TIME under load (the chip's clocks) or of the NUMBER of batches (the library's
# pipeline)?  The same 10-ms windows with 256-frame and with 64-frame batches after warm-ups of equal time / equal count.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ar; mkdir -p $O; : > $O/log.txt
run() { echo "$* : $(timeout -k 10 120 tools/diffbench "$@" 2>&1 | tr '\n' ' ' | grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ')" >> $O/log.txt; }
for rep in 1 2; do
for w in 2 5 10 25 100 400; do run --batch 256 --steps 20 --warmup $w; done
for w in 5 20 40 100 400 1600; do run --batch 64 --steps 80 --warmup $w; done
done
cat $O/log.txt
