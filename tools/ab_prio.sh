#!/bin/bash
# pipelined batches: priority of the expansion / index waves (build/ab/<variant>) x workgroups of the pack kernel
for v in "$@"; do for b in 0 768; do for rep in 1 2; do echo -n "$v blocks=$b: "; LD_LIBRARY_PATH=build/ab/$v MI355_K1_BLOCKS=$b tools/diffbench --steps 30 2>&1 | grep -o '"frames_per_s": [0-9.]*, "kernel_ms": [0-9.]*, "all_kernels_ms": [0-9.]*'; done; done; done
echo -n "sequential: "; LD_LIBRARY_PATH=build/ab/$1 MI355_PIPELINE=0 tools/diffbench --steps 30 2>&1 | grep -o '"frames_per_s": [0-9.]*, "kernel_ms": [0-9.]*, "all_kernels_ms": [0-9.]*'
