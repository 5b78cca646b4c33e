"""Generates cupyimg_amd/csrc/median27_net.hpp: the last stage of the 3 x 3 x 3 median kernel (csrc/median3d.hip).

The 27 samples of a window, sorted along z, then x, then y, form a cube q[i][j][k] that is monotone in every index.  Position
(i, j, k) then has (i+1)(j+1)(k+1) - 1 samples known to be <= it and (3-i)(3-j)(3-k) - 1 known to be >= it; the median (13 on
either side) can only sit where both counts are <= 13: 19 positions, 4 of the other 8 are below and 4 above it, so the median of
the window is the MEDIAN OF THOSE 19.  The network that takes it is found here: Batcher's odd-even merge sort on 32 wires (the
19 candidates, 6 wires at -inf, 7 at +inf, output wire 15), with
  * every comparator dropped that never exchanges on any input the partial order allows -- by the 0/1 principle (thresholding
    commutes with min / max and keeps the order) these are the 980 monotone 0/1 labelings of the 3 x 3 x 3 poset;
  * comparators with a padding wire turned into renames;
  * every min / max dropped that cannot reach the output;
and the placement of the candidates on the wires chosen by annealing on the instruction count (rank27_wires.json; the median: 62
min / max).
The emitted code is checked against every labeling and against random windows before it is written.

The same argument gives every other rank r of the window (percentile_filter / rank_filter with the full 3 x 3 x 3 footprint):
the candidates are the positions with at most r samples known below and at most 26 - r known above, 3 ... 19 of them.

usage: python scripts/gen_median27_network.py            (writes the header from scripts/rank27_wires.json)
       python scripts/gen_median27_network.py --search SECONDS [WORKERS]   (anneals every rank, keeps the better placement)"""
import itertools
import math
import os
import random
import sys
import time

CELLS = [(i, j, k) for i in range(3) for j in range(3) for k in range(3)]
N = 32
OUT_WIRE = 15            # r = 13: 6 wires at -inf below the 19 candidates, their median (rank 9) is position 15
RANK = 13                # the rank the module-level CAND / tests() / verify() refer to (set_rank())


def candidates(r):
    """positions of the sorted cube that can hold the sample of rank r (0 = smallest), and how many positions are known below it"""
    cand = [c for c in CELLS if (c[0] + 1) * (c[1] + 1) * (c[2] + 1) - 1 <= r and (3 - c[0]) * (3 - c[1]) * (3 - c[2]) - 1 <= 26 - r]
    below = [c for c in CELLS if (3 - c[0]) * (3 - c[1]) * (3 - c[2]) - 1 > 26 - r]
    return cand, len(below)


CAND, NBELOW = candidates(RANK)


def set_rank(r):
    global RANK, CAND, NBELOW
    RANK = r
    CAND, NBELOW = candidates(r)


def start_wires(r):
    """candidates, then pads: the wanted sample (rank r - NBELOW among the candidates) comes out on OUT_WIRE"""
    cand, nb = candidates(r)
    lo = OUT_WIRE - (r - nb)
    hi = N - len(cand) - lo
    assert lo >= 0 and hi >= 0
    return list(range(len(cand))) + [-2] * lo + [-1] * hi


def labelings():
    """monotone 0/1 labelings of the cube: f(i, j, k) = 1 iff k >= h[i][j], h non-increasing in i and j"""
    out = []
    for hs in itertools.product(range(4), repeat=9):
        h = [hs[0:3], hs[3:6], hs[6:9]]
        if all(h[i][j] >= h[i + 1][j] for i in range(2) for j in range(3)) and all(h[i][j] >= h[i][j + 1] for i in range(3) for j in range(2)):
            out.append(tuple(1 if k >= h[i][j] else 0 for (i, j, k) in CELLS))
    return out


def tests():
    idx = {c: n for n, c in enumerate(CELLS)}
    t = set()
    for f in labelings():
        t.add((tuple(f[idx[c]] for c in CAND), 1 if sum(f) >= 27 - RANK else 0))
    return sorted(t)


def batcher(n):
    net = []

    def merge(lo, hi, r):
        step = r * 2
        if step < hi - lo:
            merge(lo, hi, step)
            merge(lo + r, hi, step)
            for i in range(lo + r, hi - r, step):
                net.append((i, i + r))
        else:
            net.append((lo, lo + r))

    def sort(lo, hi):
        if hi - lo >= 1:
            mid = lo + (hi - lo) // 2
            sort(lo, mid)
            sort(mid + 1, hi)
            merge(lo, hi, 1)

    sort(0, n - 1)
    return net


def reduce_network(net, wires, tvecs):
    """-> (instruction count, [(a, b, 'ce' | 'swap', live_a, live_b)]) or (None, None) when the output is wrong.
    wires[w]: candidate index on wire w, -1 = +inf, -2 = -inf"""
    vals = [[(vec[p] if p >= 0 else (1 if p == -1 else 0)) for p in wires] for vec, _ in tvecs]
    pad = [0 if p >= 0 else (1 if p == -1 else -1) for p in wires]
    keep = []
    for (a, b) in net:
        if pad[a] or pad[b]:
            if (pad[a] == 1 and pad[b] != 1) or (pad[b] == -1 and pad[a] != -1):          # the values change places: a rename
                pad[a], pad[b] = pad[b], pad[a]
                for v in vals:
                    v[a], v[b] = v[b], v[a]
                keep.append((a, b, "swap"))
            continue
        if any(v[a] > v[b] for v in vals):
            keep.append((a, b, "ce"))
            for v in vals:
                if v[a] > v[b]:
                    v[a], v[b] = v[b], v[a]
    if any(v[OUT_WIRE] != want for v, (_, want) in zip(vals, tvecs)):
        return None, None
    live, ops, kept = {OUT_WIRE}, 0, []
    for (a, b, kind) in reversed(keep):
        la, lb = a in live, b in live
        if kind == "swap":
            live.discard(a)
            live.discard(b)
            if la:
                live.add(b)
            if lb:
                live.add(a)
            kept.append((a, b, kind, la, lb))
        elif la or lb:
            ops += int(la) + int(lb)
            kept.append((a, b, kind, la, lb))
            live.add(a)
            live.add(b)
    kept.reverse()
    return ops, kept


def anneal(seconds, seed, rank=13):
    set_rank(rank)
    random.seed(seed)
    net, tv = batcher(N), tests()
    wires = start_wires(rank)
    cur = None
    while cur is None:
        random.shuffle(wires)
        cur, _ = reduce_network(net, wires, tv)
    best, T, t0 = (cur, list(wires)), 3.0, time.time()
    while time.time() - t0 < seconds:
        a, b = random.sample(range(N), 2)
        if wires[a] == wires[b]:
            continue
        wires[a], wires[b] = wires[b], wires[a]
        ops, _ = reduce_network(net, wires, tv)
        if ops is not None and (ops <= cur or random.random() < math.exp((cur - ops) / T)):
            cur = ops
            if ops < best[0]:
                best = (ops, list(wires))
        else:
            wires[a], wires[b] = wires[b], wires[a]
        T = max(0.3, T * 0.999)
    return best


def straight_line(kept, wires):
    """-> [(dst, 'min' | 'max', src_a, src_b)], name of the result; names: c<n> = candidate n, t<n> = temporaries"""
    name = {w: ("c%d" % p if p >= 0 else None) for w, p in enumerate(wires)}
    code, nt = [], 0
    for (a, b, kind, la, lb) in kept:
        if kind == "swap":
            name[a], name[b] = name[b], name[a]
            continue
        xa, xb = name[a], name[b]
        assert xa is not None and xb is not None
        if la:
            code.append(("t%d" % nt, "min", xa, xb))
            name[a] = "t%d" % nt
            nt += 1
        else:
            name[a] = None
        if lb:
            code.append(("t%d" % nt, "max", xa, xb))
            name[b] = "t%d" % nt
            nt += 1
        else:
            name[b] = None
    return code, name[OUT_WIRE]


def run_code(code, result, c):
    env = {"c%d" % n: v for n, v in enumerate(c)}
    for dst, op, a, b in code:
        env[dst] = min(env[a], env[b]) if op == "min" else max(env[a], env[b])
    return env[result]


def verify(code, result):
    idx = {c: n for n, c in enumerate(CELLS)}
    for f in labelings():
        assert run_code(code, result, [f[idx[c]] for c in CAND]) == (1 if sum(f) >= 27 - RANK else 0)
    rng = random.Random(5)
    for trial in range(4000):
        w = [rng.randint(0, 9) if trial % 2 else rng.random() for _ in range(27)]
        cube = [[[w[(i * 3 + j) * 3 + k] for k in range(3)] for j in range(3)] for i in range(3)]
        for axis in range(3):                      # sort along z, x, y in turn (any order of the axes gives a monotone cube)
            for u in range(3):
                for v in range(3):
                    if axis == 0:
                        col = sorted(cube[t][u][v] for t in range(3))
                        for t in range(3):
                            cube[t][u][v] = col[t]
                    elif axis == 1:
                        col = sorted(cube[u][t][v] for t in range(3))
                        for t in range(3):
                            cube[u][t][v] = col[t]
                    else:
                        col = sorted(cube[u][v][t] for t in range(3))
                        for t in range(3):
                            cube[u][v][t] = col[t]
        assert run_code(code, result, [cube[i][j][k] for (i, j, k) in CAND]) == sorted(w)[RANK], trial


WIRES_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank27_wires.json")


def load_wires():
    import json
    d = json.load(open(WIRES_FILE))
    return {int(k): v for k, v in d.items()}


def network_for(r, wires):
    set_rank(r)
    ops, kept = reduce_network(batcher(N), wires, tests())
    assert ops is not None, r
    code, result = straight_line(kept, wires)
    assert len(code) == ops
    return code, result


def emit(path, check=True):
    placements = load_wires()
    L = []
    L.append("// median27_net.hpp -- GENERATED by scripts/gen_median27_network.py from scripts/rank27_wires.json; do not edit.")
    L.append("// Rank27Net<KO, R>: the positions (i, j, k) of a 3 x 3 x 3 window sorted along its three axes that can hold the sample of")
    L.append("// rank R (0 = smallest; 13 = the median), and the min / max network that takes it from those (see the generator).")
    L.append("// KO: the key operations (median3d_impl.hpp): KO::K, KO::min3 / med3 / max3 (a, b, c), KO::mn / mx (a, b).")
    L.append("#pragma once")
    L.append("namespace mi {")
    L.append("")
    L.append("template <class KO, int R> struct Rank27Net;")
    summary = []
    for r in range(1, 26):
        code, result = network_for(r, placements[r])
        if check:
            verify(code, result)
        nc = len(CAND)
        summary.append((r, nc, len(code)))
        L.append("")
        L.append("// rank %d: %d candidates (%d positions known below), %d min / max instructions" % (r, nc, NBELOW, len(code)))
        L.append("template <class KO> struct Rank27Net<KO, %d> {" % r)
        L.append("    using K = typename KO::K;")
        L.append("    static constexpr int NC = %d;" % nc)
        L.append("    // up / dn: the nine x-sorted values [i * 3 + j] of the rows above and below in LDS (64 keys apart), q: this row's, in registers")
        L.append("    static __device__ __forceinline__ void candidates(const K *up, const K (&q)[9], const K *dn, K (&c)[NC])")
        L.append("    {")
        for comp in range(9):
            if not any(i * 3 + j == comp for (i, j, k) in CAND):
                continue
            L.append("        {")
            L.append("            const K a = up[%d * 64], b = q[%d], d = dn[%d * 64];" % (comp, comp, comp))
            for n, (i, j, k) in enumerate(CAND):
                if i * 3 + j == comp:
                    L.append("            c[%d] = KO::%s(a, b, d);      // (%d, %d, %d)" % (n, ("min3", "med3", "max3")[k], i, j, k))
            L.append("        }")
        L.append("    }")
        L.append("    static __device__ __forceinline__ K select(const K (&c)[NC])")
        L.append("    {")
        L.append("        const K " + ", ".join("c%d = c[%d]" % (n, n) for n in range(nc)) + ";")
        for dst, op, a, b in code:
            L.append("        const K %s = KO::%s(%s, %s);" % (dst, "mn" if op == "min" else "mx", a, b))
        L.append("        return %s;" % result)
        L.append("    }")
        L.append("};")
    L.append("")
    L.append("}  // namespace mi")
    open(path, "w").write("\n".join(L) + "\n")
    print("wrote", path)
    for r, nc, ops in summary:
        print("  rank %2d: %2d candidates, %3d instructions" % (r, nc, ops))


def search_all(seconds, workers):
    import json
    from concurrent.futures import ProcessPoolExecutor
    placements = load_wires() if os.path.exists(WIRES_FILE) else {}
    jobs = [(seconds, 100 + r, r) for r in range(1, 26)]
    with ProcessPoolExecutor(workers) as ex:
        for (secs, seed, r), best in zip(jobs, ex.map(_anneal_job, jobs)):
            old = None
            if r in placements:
                set_rank(r)
                old, _ = reduce_network(batcher(N), placements[r], tests())
            if old is None or best[0] < old:
                placements[r] = best[1]
            print("rank", r, "instructions", best[0], "(kept %s)" % old if old is not None and old <= best[0] else "", flush=True)
    json.dump({str(k): v for k, v in sorted(placements.items())}, open(WIRES_FILE, "w"))


def _anneal_job(job):
    return anneal(*job)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--search":
        search_all(float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    else:
        emit(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cupyimg_amd", "csrc", "median27_net.hpp"))
