// Local search over min / max programs for "the 13th smallest of 25 values in five ascending columns" (see search.py for
// the proof obligation: all 7776 zero-one inputs with sorted columns; a wire is a 7776-bit set, min = AND, max = OR).
// Starts from the program search.py wrote (stdin: lines "k a b", last line "out w") and minimises the number of
// instructions AFTER fusing: gfx950 has three-input packed minimum / maximum (v_pk_minimum3_f16 / v_pk_maximum3_f16), so an
// operation whose only reader is an operation of the same kind costs nothing extra as long as that reader has not
// absorbed another one.  Moves: redirect the readers of an operation to an earlier wire; give an operation other
// operands; insert a fresh operation and let a later one read it.  Accepts equal cost always, worse cost with a small
// probability.  Prints the best program in the same format.
//
// SHARE = K > 1: the kernel computes K neighbouring windows of one colour channel in a lane; they have the columns 0, 1, 2
// of the program in common and differ in the columns 3, 4 (for the windows P0..P4, P1..P5, P2..P6 of seven columns:
// common P2 P3 P4; own P0 P1 / P1 P5 / P5 P6 -- the median does not care which column is which).  An operation that reads
// only common columns is computed once for the K windows (the compiler merges the identical expressions), so it costs
// 1 / K; the cost minimised is K x (instructions that depend on column 3 or 4) + (instructions that do not), and a shared
// operation never folds into a private reader (it has K of them).
// (SHARE = 2, NPRIV = 1: two windows P0..P4, P1..P5 of six columns with four columns in common.)
//   g++ -O2 -std=c++17 -o anneal anneal.cpp && ./anneal SEED SECONDS [TEMP [SHARE [NPRIV]]] < start.txt > best.txt
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static const int NC = 7776, NW = (NC + 63) / 64;
struct Bits { uint64_t w[NW]; };
static Bits INP[25], WANT;
static bool eq(const Bits &a, const Bits &b) { return !memcmp(a.w, b.w, sizeof a.w); }

struct Op { int k, a, b; };
struct Prog { std::vector<Op> ops; int out; };

static void init_cases() {
    memset(INP, 0, sizeof INP); memset(&WANT, 0, sizeof WANT);
    int idx = 0;
    for (int c0 = 0; c0 <= 5; c0++) for (int c1 = 0; c1 <= 5; c1++) for (int c2 = 0; c2 <= 5; c2++)
    for (int c3 = 0; c3 <= 5; c3++) for (int c4 = 0; c4 <= 5; c4++, idx++) {
        const int ones[5] = {c0, c1, c2, c3, c4};   // same order as itertools.product
        for (int c = 0; c < 5; c++) for (int r = 0; r < 5; r++)
            if (r >= 5 - ones[c]) INP[5 * c + r].w[idx >> 6] |= 1ull << (idx & 63);
        if (c0 + c1 + c2 + c3 + c4 >= 13) WANT.w[idx >> 6] |= 1ull << (idx & 63);
    }
}

static void eval(const Prog &p, std::vector<Bits> &v) {
    v.resize(25 + p.ops.size());
    for (int i = 0; i < 25; i++) v[i] = INP[i];
    for (size_t i = 0; i < p.ops.size(); i++) {
        const Op &o = p.ops[i];
        Bits &d = v[25 + i];
        const Bits &x = v[o.a], &y = v[o.b];
        if (o.k) for (int j = 0; j < NW; j++) d.w[j] = x.w[j] | y.w[j];
        else for (int j = 0; j < NW; j++) d.w[j] = x.w[j] & y.w[j];
    }
}

// alias wires with equal bits to the earliest one, drop dead code
static Prog clean(const Prog &p, std::vector<Bits> &v) {
    eval(p, v);
    const int n = (int)p.ops.size();
    std::vector<int> alias(25 + n);
    std::vector<uint64_t> hash(25 + n);
    for (int i = 0; i < 25 + n; i++) {
        uint64_t h = 1469598103934665603ull;
        for (int j = 0; j < NW; j++) h = (h ^ v[i].w[j]) * 1099511628211ull;
        hash[i] = h;
        alias[i] = i;
        for (int e = 0; e < i; e++)
            if (hash[e] == h && alias[e] == e && eq(v[e], v[i])) { alias[i] = e; break; }
    }
    std::vector<char> live(25 + n, 0);
    const int out = alias[p.out];
    live[out] = 1;
    for (int i = n - 1; i >= 0; i--)
        if (live[25 + i]) { live[alias[p.ops[i].a]] = 1; live[alias[p.ops[i].b]] = 1; }
    Prog q;
    std::vector<int> remap(25 + n, -1);
    for (int i = 0; i < 25; i++) remap[i] = i;
    for (int i = 0; i < n; i++)
        if (live[25 + i] && alias[25 + i] == 25 + i) {
            remap[25 + i] = 25 + (int)q.ops.size();
            q.ops.push_back({p.ops[i].k, remap[alias[p.ops[i].a]], remap[alias[p.ops[i].b]]});
        }
    q.out = remap[out];
    return q;
}

static int SHARE = 1, NPRIV = 2;
// instructions after fusing: an op with exactly one reader of the same kind is absorbed by it if that reader is free
static int cost(const Prog &p) {
    const int n = (int)p.ops.size();
    std::vector<char> priv(25 + n, 0);
    for (int i = 5 * (5 - NPRIV); i < 25; i++) priv[i] = 1;
    for (int i = 0; i < n; i++) priv[25 + i] = priv[p.ops[i].a] | priv[p.ops[i].b];
    std::vector<int> uses(25 + n, 0);
    for (const Op &o : p.ops) { uses[o.a]++; uses[o.b]++; }
    uses[p.out] += 2;
    std::vector<char> absorbed(n, 0);
    int c = 0;
    for (int i = n - 1; i >= 0; i--) {   // readers first
        if (absorbed[i]) continue;       // it is part of its reader
        c += (SHARE > 1 && priv[25 + i]) ? SHARE : 1;
        const Op &o = p.ops[i];
        // absorb one operand op of the same kind with a single use (which may itself not absorb: it has become part of i,
        // whose three inputs are then full)
        for (int w : {o.a, o.b})
            if (w >= 25 && uses[w] == 1 && p.ops[w - 25].k == o.k && o.a != o.b && (SHARE == 1 || priv[w] == priv[25 + i])) {
                absorbed[w - 25] = 1;
                break;
            }
    }
    return c;
}

static bool ok(const Prog &p, std::vector<Bits> &v) { eval(p, v); return eq(v[p.out], WANT); }

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 60;
    const double temp = argc > 3 ? atof(argv[3]) : 0.02;   // probability of accepting a slightly worse program
    SHARE = argc > 4 ? atoi(argv[4]) : 1;
    NPRIV = argc > 5 ? atoi(argv[5]) : 2;    // the last NPRIV columns are a window's own (2 with K = 3 windows, 1 with K = 2)
    if (NPRIV < 1 || NPRIV > 4 || SHARE < 1) return 2;
    init_cases();
    Prog p;
    char word[16]; int a, b, c;
    while (scanf("%15s", word) == 1) {
        if (!strcmp(word, "out")) { if (scanf("%d", &a) != 1) return 2; p.out = a; break; }
        c = atoi(word);
        if (scanf("%d %d", &a, &b) != 2) return 2;
        p.ops.push_back({c, a, b});
    }
    std::vector<Bits> v;
    if (p.ops.empty()) { fprintf(stderr, "no program on stdin\n"); return 2; }
    if (!ok(p, v)) { fprintf(stderr, "start program is wrong\n"); return 1; }
    p = clean(p, v);
    std::mt19937_64 rng(seed);
    auto rnd = [&](int n) { return (int)(rng() % (uint64_t)n); };
    Prog best = p;
    int cbest = cost(p), ccur = cbest;
    fprintf(stderr, "start: %zu ops, cost %d\n", p.ops.size(), cbest);
    const auto t0 = std::chrono::steady_clock::now();
    long it = 0;
    while (true) {
        if ((++it & 255) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) break;
        const int n = (int)p.ops.size();
        Prog q = p;
        const int mv = rnd(100);
        const int i = rnd(n);
        if (mv < 35) {                 // readers of op i read an earlier wire instead
            const int x = rnd(25 + i);
            for (Op &o : q.ops) { if (o.a == 25 + i) o.a = x; if (o.b == 25 + i) o.b = x; }
            if (q.out == 25 + i) q.out = x;
        } else if (mv < 70) {          // other operands / kind for op i
            q.ops[i].k = rnd(4) ? q.ops[i].k : rnd(2);
            if (rnd(2)) q.ops[i].a = rnd(25 + i); else q.ops[i].b = rnd(25 + i);
        } else if (mv < 85) {          // one operand of op i becomes an earlier wire
            if (rnd(2)) q.ops[i].a = rnd(25 + i); else q.ops[i].b = rnd(25 + i);
        } else {                        // insert a fresh op before op i and let op i (or a later one) read it
            Op f{rnd(2), rnd(25 + i), rnd(25 + i)};
            q.ops.insert(q.ops.begin() + i, f);
            for (size_t j = i + 1; j < q.ops.size(); j++) {
                if (q.ops[j].a >= 25 + i) q.ops[j].a++;
                if (q.ops[j].b >= 25 + i) q.ops[j].b++;
            }
            if (q.out >= 25 + i) q.out++;
            const int j = i + 1 + rnd((int)q.ops.size() - i - 1);
            if (rnd(2)) q.ops[j].a = 25 + i; else q.ops[j].b = 25 + i;
        }
        if (!ok(q, v)) continue;
        q = clean(q, v);
        const int cq = cost(q);
        if (cq <= ccur || (cq <= ccur + SHARE && (rng() % 10000) < temp * 10000)) {
            p = q; ccur = cq;
            if (cq < cbest) {
                cbest = cq; best = q;
                fprintf(stderr, "cost %d (%zu ops) after %ld moves, %.0f s\n", cq, q.ops.size(), it,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
        }
    }
    for (const Op &o : best.ops) printf("%d %d %d\n", o.k, o.a, o.b);
    printf("out %d\n", best.out);
    fprintf(stderr, "best cost %d, %zu ops, %ld moves\n", cbest, best.ops.size(), it);
    return 0;
}
