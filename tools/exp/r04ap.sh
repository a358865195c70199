#!/bin/bash
# non-temporal output stores as the product: the whole bench line (secondary lines included) against the plain-store build, then the suites
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ap
export TMPDIR=/tmp
{
echo "== bench.py, plain stores (build/ab/base)"; MI355DIFF_LIB=$PWD/build/ab/base/libmi355diff.so python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r04ap/bench_base.json 2>gpurun_out/r04ap/bench_base.err; echo rc=$?
echo "== bench.py, non-temporal stores (in-tree)"; python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r04ap/bench_xnt.json 2>gpurun_out/r04ap/bench_xnt.err; echo rc=$?
echo "== bench.py, plain stores again"; MI355DIFF_LIB=$PWD/build/ab/base/libmi355diff.so python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r04ap/bench_base2.json 2>/dev/null; echo rc=$?
echo "== bench.py, non-temporal stores again"; python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r04ap/bench_xnt2.json 2>/dev/null; echo rc=$?
python3 - <<'PY'
import json
for n in ('base','xnt','base2','xnt2'):
    try: d=json.loads(open('gpurun_out/r04ap/bench_%s.json'%n).read().strip().splitlines()[-1])
    except Exception as e: print(n, 'unreadable', e); continue
    print("%-6s headline %.4f ms frac %.4f | two_streams %s | pair %.4f | S0 %.4f PN %.4f P0 %.4f | c3 %.3f (seq %.3f) c4 %.3f (seq %.3f) | parity %s" % (n, d['ms_per_step'], d['roofline']['frac'], d.get('two_streams_one_gpu',{}).get('frac'), d['pair_mode']['frac'], d['regimes']['S0_refrand_pairs']['frac'], d['regimes']['P_eq_N_pairs']['frac'], d['regimes']['P_eq_0_pairs']['frac'], d['config3']['us_per_frame'], d['config3']['sequential_us_per_frame'], d['config4']['us_per_frame'], d['config4']['sequential_us_per_frame'], d.get('parity')))
PY
echo "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== soak"; timeout -k 10 400 python tests/soak.py 3000 2>&1 | tail -2
} > gpurun_out/r04ap/log.txt 2>&1
cat gpurun_out/r04ap/log.txt
