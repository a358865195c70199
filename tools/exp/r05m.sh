#!/bin/bash
# r05m: the batch total for the adaptive overlap stored by the index kernel (no copy, no event behind every batch) against the committed library
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05m; mkdir -p $O
{
echo "=== parity (product = note)"; timeout -k 10 600 python -m pytest tests/test_diff_pack_gpu.py tests/test_fuzz_gpu.py tests/test_pipe_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in fin note; do
  echo -n "[$v pipelined] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 30 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
done
done
for rep in 1 2; do
for v in fin note; do
  echo -n "[$v bench] "; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-host-path --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['regimes']
print(d['ms_per_step'], 'c3', d['config3']['us_per_frame'], d['config3']['sequential_us_per_frame'], 'c4', d['config4']['us_per_frame'], d['config4']['sequential_us_per_frame'], 'pair', d['pair_mode']['frac'], d['pair_mode']['frac_sequential'], 'S0', r['S0_refrand_pairs']['frac'], r['S0_refrand_pairs']['frac_sequential'], 'PN', r['P_eq_N_pairs']['frac'], r['P_eq_N_pairs']['frac_sequential'], '2s', d['two_streams_one_gpu']['frac'])"
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
