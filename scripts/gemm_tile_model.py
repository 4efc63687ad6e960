"""(round 3) CPU model of the tile cutting / LPT of the grouped GEMM for the two theta lists: useful / executed flops and the makespan balance, with and
without a 96-wide remainder class and with split-K of the heaviest tiles."""
import sys, heapq
import numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from cyten_amd import workloads as wl
from cyten_amd import abelian as ab
from numpy_backend import NumpyGroupedBackend
bbn = NumpyGroupedBackend()

def problems(A, B):
    a, b = ab.AbelianTensor.from_spec(bbn, A), ab.AbelianTensor.from_spec(bbn, B)
    plan = ab.compose_plan(a, b, 1)
    out = []
    for g, shp in zip(plan.pairs, plan.res_shapes):
        M = int(np.prod(shp[:len(shp)//2])) if False else None
    a2, b2 = ab._compose_operands(bbn, a, b, 1, plan)
    for g in plan.pairs:
        Ks = [a2[i].shape[1] for i, j in g]
        M, N = a2[g[0][0]].shape[0], b2[g[0][1]].shape[1]
        out.append((M, N, Ks))
    return out

def cut(ext, base, with96):
    out = []; off = 0
    while off + base <= ext:
        out.append(base); off += base
    r = ext - off
    if r == 0: return out
    w = base
    if off > 0 or base == 64:
        if r <= 16: w = 16
        elif r <= 32: w = 32
        elif r <= 64: w = 64
        elif with96 and r <= 96: w = 96
    out.append(min(w, base))
    return out

def pick_class(M, N):
    s, l = min(M, N), max(M, N)
    if s >= 96 and l >= 128: return 128
    if s >= 40: return 64
    return 32

def tiles(probs, with96, ncu=256):
    n128 = sum(-(-M//128) * -(-N//128) for M, N, Ks in probs if pick_class(M, N) == 128)
    demote = 0 < n128 < ncu
    ts = []
    useful = 0
    for M, N, Ks in probs:
        base = pick_class(M, N)
        if base == 128 and demote: base = 64
        if base == 32:
            for tm in range(-(-M//32)):
                for tn in range(-(-N//32)): ts.append((32, 32, Ks))
        else:
            for bm in cut(M, base, with96):
                for bn in cut(N, base, with96): ts.append((bm, bn, Ks))
        useful += 2 * M * N * sum(Ks)
    return ts, useful

def cost(bm, bn, Ks, ovh):
    kt = sum(-(-k // 16) for k in Ks)            # k-tiles (each segment padded to 16)
    return (kt + ovh) * (bm * bn) / (128 * 128) if False else kt * (bm * bn) / (128.0 * 128.0) * eff(bm, bn) + ovh

def eff(bm, bn):
    # relative per-flop cost of a class against 128x128 (narrow tiles use the MFMA pipe less well)
    a = bm * bn
    if a >= 128 * 96: return 1.0
    if a >= 96 * 96: return 1.08
    if a >= 64 * 64: return 1.25
    return 1.6

def lpt(costs, slots=512):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in sorted(costs, reverse=True):
        t = heapq.heappop(h); heapq.heappush(h, t + c)
    return max(h)

for name, (A, B) in (('u1', wl.config_u1_mps(4096)), ('u1u1', wl.config_u1u1_mps(4096))):
    probs = problems(A, B)
    for with96 in (False, True):
        ts, useful = tiles(probs, with96)
        executed = sum(2 * bm * bn * sum(-(-k // 16) * 16 for k in Ks) for bm, bn, Ks in ts)
        for ovh in (2.0, 4.0):
            costs = [cost(bm, bn, Ks, ovh) for bm, bn, Ks in ts]
            mk = lpt(costs); ideal = sum(costs) / 512
            # split-K of every tile above the ideal load into two halves (+1 k-tile of fix-up each)
            sp = []
            for c in costs:
                if c > 0.6 * ideal: sp += [c / 2 + 1.0, c / 2 + 1.0]
                else: sp.append(c)
            mk2 = lpt(sp)
            print(f'{name} 96={with96} ovh={ovh}: {len(ts)} tiles, useful/executed {useful / executed:.3f}, makespan {mk:.1f} vs ideal {ideal:.1f} (balance {ideal / mk:.3f}); split-K: {len(sp)} units makespan {mk2:.1f} (balance vs unsplit-ideal {ideal / mk2:.3f})')

print('--- U(1) cost distribution')
probs = problems(*wl.config_u1_mps(4096))
ts, useful = tiles(probs, False)
costs = sorted([cost(bm, bn, Ks, 3.0) for bm, bn, Ks in ts], reverse=True)
import collections
print('top', [round(c,1) for c in costs[:12]], 'n', len(costs), 'sum/512', sum(costs)/512)
hist = collections.Counter(round(c) for c in costs); print(sorted(hist.items(), reverse=True)[:25])
ideal = sum(costs)/512
for thr in (1.0, 0.95, 0.9, 0.8, 0.7):
    for parts in (2, 3):
        sp = []
        for c in costs:
            if c > thr * ideal: sp += [c / parts + 1.5] * parts
            else: sp.append(c)
        print(f'split tiles > {thr} ideal into {parts}: units {len(sp)} makespan {lpt(sp):.1f} (unsplit {lpt(costs):.1f}, ideal {ideal:.1f})')
