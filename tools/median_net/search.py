#!/usr/bin/env python3
"""Selection networks for the 5x5 median whose five columns arrive SORTED (k_median5x5_strip, csrc/filters.hip).

The kernel sorts every column of five rows once and shares it between the five outputs whose windows contain it
(tests/noise_filter_benchmark/v3.cu:79-88 sorts all 25 values per output).  What is left per output is a straight-line
program of min / max operations on 25 values of which the five groups of five are known to be ascending.  This script
derives such programs by pruning known-good networks and checks them exhaustively:

  * zero-one principle with a precondition: a min/max program returns the median for every input whose columns are
    sorted iff it does so for every 0/1 input whose columns are sorted (thresholding commutes with min, max and the
    median, and keeps columns sorted) -- 6^5 = 7776 cases, each wire is one 7776-bit integer (min = AND, max = OR);
  * pruning: an operation is replaced by one of its operands wherever the result stays right on all cases (this removes
    exchanges that never swap, halves of exchanges nobody reads, and whole sub-networks the sortedness makes
    redundant), in random orders, from several starting networks and wire assignments; the shortest survivor is
    written out as C macros (csrc/median_net.h) together with the operation count.

Usage: search.py [--seconds S] [--seed N] [--emit header] [--dump program.txt]      prune, write the best
       anneal SEED SECONDS < program.txt > better.txt                                (anneal.cpp: local search, fused cost)
       search.py --load better.txt --emit ../../cudavideostream_amd/csrc/median_net.h   re-check in Python, write the header
A program file holds one operation per line, "kind a b" (kind 0 = min, 1 = max; wires 0..24 = 5 * column + rank, 25 + i =
the result of line i), and a last line "out wire".
"""
import argparse, itertools, random, sys, time

NCOL, NROW = 5, 5
CASES = list(itertools.product(range(NROW + 1), repeat=NCOL))   # ones per column (ascending: the ones are at the top ranks)
NCASE = len(CASES)
FULL = (1 << NCASE) - 1


def input_bits():
    """wire (col, rank) -> bitset over the cases: the value is 1 iff rank >= 5 - ones(col)"""
    w = {}
    for c in range(NCOL):
        for r in range(NROW):
            b = 0
            for i, ones in enumerate(CASES):
                if r >= NROW - ones[c]:
                    b |= 1 << i
            w[(c, r)] = b
    return w


def expected_bits():
    b = 0
    for i, ones in enumerate(CASES):
        if sum(ones) >= 13:          # the 13th smallest of 25 is 1 iff at least 13 ones
            b |= 1 << i
    return b


INP = input_bits()
WANT = expected_bits()


class Prog:
    """ops[i] = (kind, a, b): kind 0 = min, 1 = max; operands are wire numbers: 0..24 inputs (5 * col + rank), 25 + i = ops[i]"""

    def __init__(self, ops, out):
        self.ops, self.out = list(ops), out

    def eval_bits(self):
        v = [INP[(i // 5, i % 5)] for i in range(25)]
        for k, a, b in self.ops:
            v.append(v[a] | v[b] if k else v[a] & v[b])
        return v

    def ok(self):
        return self.eval_bits()[self.out] == WANT

    def dce(self):
        live = set([self.out])
        for i in range(len(self.ops) - 1, -1, -1):
            if 25 + i in live:
                live.add(self.ops[i][1]); live.add(self.ops[i][2])
        remap, ops = {i: i for i in range(25)}, []
        for i, (k, a, b) in enumerate(self.ops):
            if 25 + i in live:
                remap[25 + i] = 25 + len(ops)
                ops.append((k, remap[a], remap[b]))
        return Prog(ops, remap[self.out])

    def simplify(self):
        """constant facts on the case set: an op whose result equals an operand (or an earlier wire) on all cases is that wire"""
        v = [INP[(i // 5, i % 5)] for i in range(25)]
        seen = {}
        for i, b in enumerate(v):
            seen.setdefault(b, i)
        alias = list(range(25))
        ops = []
        for k, a, b in self.ops:
            a, b = alias[a], alias[b]
            r = v[a] | v[b] if k else v[a] & v[b]
            if r in seen:
                alias.append(seen[r]); v.append(r); ops.append((k, a, b))   # kept as dead code, removed by dce
            else:
                seen[r] = 25 + len(ops)
                alias.append(25 + len(ops)); v.append(r); ops.append((k, a, b))
        p = Prog(ops, alias[self.out])
        return p.dce()


def from_exchanges(ces, assign, out_pos):
    """compare-exchange list on positions -> SSA min/max program; assign[pos] = input wire"""
    cur = list(assign)
    ops = []
    for a, b in ces:
        lo = 25 + len(ops); ops.append((0, cur[a], cur[b]))
        hi = 25 + len(ops); ops.append((1, cur[a], cur[b]))
        cur[a], cur[b] = lo, hi
    return Prog(ops, cur[out_pos])


DEVILLARD25 = [(0, 1), (3, 4), (2, 4), (2, 3), (6, 7), (5, 7), (5, 6), (9, 10), (8, 10), (8, 9), (12, 13), (11, 13), (11, 12), (15, 16),
               (14, 16), (14, 15), (18, 19), (17, 19), (17, 18), (21, 22), (20, 22), (20, 21), (23, 24), (2, 5), (3, 6), (0, 6), (0, 3),
               (4, 7), (1, 7), (1, 4), (11, 14), (8, 14), (8, 11), (12, 15), (9, 15), (9, 12), (13, 16), (10, 16), (10, 13), (20, 23),
               (17, 23), (17, 20), (21, 24), (18, 24), (18, 21), (19, 22), (8, 17), (9, 18), (0, 18), (0, 9), (10, 19), (1, 19), (1, 10),
               (11, 20), (2, 20), (2, 11), (12, 21), (3, 21), (3, 12), (13, 22), (4, 22), (4, 13), (14, 23), (5, 23), (5, 14), (15, 24),
               (6, 24), (6, 15), (7, 16), (7, 19), (13, 21), (15, 23), (7, 13), (7, 15), (1, 9), (3, 11), (5, 17), (11, 17), (9, 17),
               (4, 10), (6, 12), (7, 14), (4, 6), (4, 7), (12, 14), (10, 14), (6, 7), (10, 12), (6, 10), (6, 17), (12, 17), (7, 17),
               (7, 10), (12, 18), (7, 12), (10, 18), (12, 20), (10, 20), (10, 12)]


def batcher(n):
    """odd-even merge sort on n wires (n padded to a power of two, exchanges touching the padding dropped)"""
    p2 = 1
    while p2 < n:
        p2 *= 2
    ces = []

    def merge(lo, m, r):
        step = r * 2
        if step < m:
            merge(lo, m, step); merge(lo + r, m, step)
            for i in range(lo + r, lo + m - r, step):
                ces.append((i, i + r))
        else:
            ces.append((lo, lo + r))

    def sort(lo, m):
        if m > 1:
            h = m // 2
            sort(lo, h); sort(lo + h, h); merge(lo, m, 1)

    sort(0, p2)
    # padding wires hold +infinity: they sit at the top positions; an exchange with one of them never moves anything
    return [(a, b) for a, b in ces if a < n and b < n]


SORT5 = [(0, 1), (3, 4), (2, 4), (2, 3), (0, 3), (0, 2), (1, 4), (1, 3), (1, 2)]


def rows_then_sort():
    """sort the five rows (rank r of every column), then a full sorter on all 25 (position 5 * r + c)"""
    ces = []
    for r in range(5):
        ces += [(5 * r + a, 5 * r + b) for a, b in SORT5]
    return ces + batcher(25)


def prune(p, rng, rounds=3):
    p = p.simplify()
    for _ in range(rounds):
        changed = False
        order = list(range(len(p.ops)))
        rng.shuffle(order)
        for i in order:
            if i >= len(p.ops):
                continue
            k, a, b = p.ops[i]
            for rep in (a, b) if rng.random() < 0.5 else (b, a):
                ops = [(kk, rep if aa == 25 + i else aa, rep if bb == 25 + i else bb) for kk, aa, bb in p.ops]
                out = rep if p.out == 25 + i else p.out
                q = Prog(ops, out)
                if q.ok():
                    p = q
                    changed = True
                    break
        p = p.dce().simplify()
        if not changed:
            break
    return p


def depth(p):
    d = [0] * 25
    for k, a, b in p.ops:
        d.append(1 + max(d[a], d[b]))
    return d[p.out]


def emit(p, path):
    lines = ["// GENERATED by tools/median_net/search.py -- do not edit.  The 13th smallest of 25 values whose five columns",
             "// c[col][rank] are ascending, as %d min / max operations (depth %d; %d instructions where an operation whose only" % (len(p.ops), depth(p), fused_cost(p)),
             "// reader has the same kind folds into a three-input one); exhaustively checked on the 7776 zero-one",
             "// inputs with sorted columns (which proves it for every input with sorted columns: see the script).",
             "// MEDNET_IN(col, rank) names an input, MEDNET_MIN / MEDNET_MAX(dst, a, b) define temporaries t<n>.",
             "#define MEDNET_OPS %d" % len(p.ops),
             "#define MEDNET_BODY \\"]

    def name(w):
        return "MEDNET_IN(%d, %d)" % (w // 5, w % 5) if w < 25 else "t%d" % (w - 25)

    for i, (k, a, b) in enumerate(p.ops):
        lines.append("    %s(t%d, %s, %s) \\" % ("MEDNET_MAX" if k else "MEDNET_MIN", i, name(a), name(b)))
    lines.append("    MEDNET_OUT(%s)" % name(p.out))
    open(path, "w").write("\n".join(lines) + "\n")


def fused_cost(p):
    """instructions when an operation whose only reader has the same kind is folded into it (three-input minimum / maximum)"""
    uses = [0] * (25 + len(p.ops))
    for k, a, b in p.ops:
        uses[a] += 1; uses[b] += 1
    uses[p.out] += 2
    absorbed, c = set(), 0
    for i in range(len(p.ops) - 1, -1, -1):
        if i in absorbed:
            continue
        c += 1
        k, a, b = p.ops[i]
        for w in (a, b):
            if w >= 25 and uses[w] == 1 and p.ops[w - 25][0] == k and a != b:
                absorbed.add(w - 25)
                break
    return c


def load(path):
    ops, out = [], None
    for line in open(path):
        f = line.split()
        if not f:
            continue
        if f[0] == "out":
            out = int(f[1])
        else:
            ops.append((int(f[0]), int(f[1]), int(f[2])))
    return Prog(ops, out)


def dump(p, path):
    with open(path, "w") as f:
        for k, a, b in p.ops:
            f.write("%d %d %d\n" % (k, a, b))
        f.write("out %d\n" % p.out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--emit")
    ap.add_argument("--dump")
    ap.add_argument("--load")
    a = ap.parse_args()
    if a.load:
        p = load(a.load)
        assert all(x < 25 + i and y < 25 + i for i, (k, x, y) in enumerate(p.ops)), "operands must come first"
        assert p.ok(), "the program is wrong on some zero-one input with sorted columns"
        p = p.dce()
        print("%d operations, %d instructions with three-input minimum / maximum, depth %d: correct on all %d cases"
              % (len(p.ops), fused_cost(p), depth(p), NCASE))
        if a.emit:
            emit(p, a.emit)
        if a.dump:
            dump(p, a.dump)
        return
    rng = random.Random(a.seed)
    best = None
    t0 = time.time()
    n = 0
    while time.time() - t0 < a.seconds:
        kind = n % 3
        n += 1
        if kind == 0:        # Devillard's network, columns assigned to its positions in a random way
            cols = list(range(5)); rng.shuffle(cols)
            pos = list(range(25))
            if n > 3:
                rng.shuffle(pos)
            assign = [0] * 25
            for j, ps in enumerate(pos):
                assign[ps] = 5 * cols[j // 5] + j % 5
            p = from_exchanges(DEVILLARD25, assign, 12)
        elif kind == 1:      # Batcher's merge sort on column-major positions (its first phases sort the columns: they vanish)
            cols = list(range(5)); rng.shuffle(cols)
            assign = [5 * cols[j // 5] + j % 5 for j in range(25)]
            p = from_exchanges(batcher(25), assign, 12)
        else:                # sorted rows, then a full sorter
            cols = list(range(5)); rng.shuffle(cols)
            assign = [5 * cols[j % 5] + j // 5 for j in range(25)]      # position 5 r + c = column c, rank r
            p = from_exchanges(rows_then_sort(), assign, 12)
        assert p.ok(), kind
        q = prune(p, rng)
        assert q.ok()
        if best is None or len(q.ops) < len(best.ops):
            best = q
            print("start %d (%d ops) -> %d ops, depth %d  [%.0f s]" % (kind, len(p.ops), len(q.ops), depth(q), time.time() - t0), flush=True)
            if a.emit:
                emit(best, a.emit)
            if a.dump:
                dump(best, a.dump)
    print("best: %d ops" % len(best.ops))


if __name__ == "__main__":
    main()
