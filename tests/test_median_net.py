"""csrc/median_net.h -- the selection program of k_median5x5_strip -- is generated (tools/median_net); this checks the
header that is compiled, on the CPU: every zero-one input with sorted columns (which proves it for all inputs with
sorted columns, see tools/median_net/search.py), random byte inputs against numpy's sort, and that the header is
what the committed program file generates."""
import importlib.util
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "cudavideostream_amd", "csrc", "median_net.h")
TOOL = os.path.join(ROOT, "tools", "median_net", "search.py")
PROGRAM = os.path.join(ROOT, "tools", "median_net", "median_net.txt")


def _tool():
    spec = importlib.util.spec_from_file_location("median_net_search", TOOL)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _parse_header():
    """-> (ops, out): ops[i] = (kind, a, b) with wires 0..24 = 5 * column + rank and 25 + i = the result of ops[i]"""
    text = open(HEADER).read()
    wire = lambda s: (5 * int(s[0]) + int(s[1])) if isinstance(s, tuple) else 25 + int(s)
    tok = r"(?:MEDNET_IN\((\d), (\d)\)|t(\d+))"
    ops = []
    for m in re.finditer(r"MEDNET_(MIN|MAX)\(t(\d+), " + tok + ", " + tok + r"\)", text):
        assert int(m.group(2)) == len(ops)
        a = wire((m.group(3), m.group(4))) if m.group(3) is not None else wire(m.group(5))
        b = wire((m.group(6), m.group(7))) if m.group(6) is not None else wire(m.group(8))
        ops.append((1 if m.group(1) == "MAX" else 0, a, b))
    m = re.search(r"MEDNET_OUT\(" + tok + r"\)", text)
    out = wire((m.group(1), m.group(2))) if m.group(1) is not None else wire(m.group(3))
    assert len(ops) == int(re.search(r"#define MEDNET_OPS (\d+)", text).group(1))
    return ops, out


def test_header_is_right_on_every_zero_one_input_with_sorted_columns():
    S = _tool()
    ops, out = _parse_header()
    assert all(a < 25 + i and b < 25 + i for i, (k, a, b) in enumerate(ops))
    assert S.Prog(ops, out).ok()
    assert len(S.CASES) == 7776


def test_header_against_numpy_on_bytes():
    ops, out = _parse_header()
    rng = np.random.default_rng(7)
    n = 20000
    cols = np.sort(rng.integers(0, 256, (n, 5, 5), dtype=np.uint8), axis=2)       # [case, column, rank] ascending
    cols[: n // 4] = np.sort(rng.integers(0, 3, (n // 4, 5, 5), dtype=np.uint8) * 127, axis=2)   # many ties
    v = [cols[:, i // 5, i % 5] for i in range(25)]
    for k, a, b in ops:
        v.append(np.maximum(v[a], v[b]) if k else np.minimum(v[a], v[b]))
    want = np.sort(cols.reshape(n, 25), axis=1)[:, 12]
    assert np.array_equal(v[out], want)


def test_header_is_generated_from_the_committed_program(tmp_path):
    out = tmp_path / "median_net.h"
    r = subprocess.run([sys.executable, TOOL, "--load", PROGRAM, "--emit", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == open(HEADER).read()
