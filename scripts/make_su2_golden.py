"""Generate tests/golden/su2_chi512.npz: the block structure of an SU(2)-symmetric two-site MPS update at chi = 512
(BASELINE cfg4) from first principles -- fusion rules for the coupled-sector matrices of FusionTreeBackend
(/root/reference/src/backends/fusion_tree_backend.cpp:669-698: ONE matrix_dot per coupled sector) and Wigner 6j symbols
(sympy.physics.wigner) for the recoupling table of a tree move (TreePairMapping::transform_tensor,
src/backends/fusion_tree_mapping.cpp:391-513: every new tree block is a linear combination of old tree blocks).

Physics of the fixture: spin-1/2 chain, bond legs carry SU(2) sectors j with multiplicities m_j; a bond to the left of an
even site has integer spins, the next one half-integer spins.  All spins are stored doubled (2j) as integers.
  * compose  theta = A . B,  A: [vL (x) p1 -> mid],  B: [mid -> p2 (x) vR]: per coupled sector J (= mid spin)
        rows(J) = m_L(J - 1/2) + m_L(J + 1/2),  K(J) = m_mid(J),  cols(J) = m_R(J - 1/2) + m_R(J + 1/2)
  * tree move on T: [(vL (x) p1) (x) p2 -> vR]  ->  [vL (x) (p1 (x) p2) -> vR]  (one F-move): for every left spin a, coupled
    spin J and new intermediate f in {0, 1}
        new[(a, f), J] = sum_e F^{a 1/2 1/2}_{J; e f} old[(a, e), J],   e in {a - 1/2, a + 1/2},
        F^{abc}_{J;ef} = (-1)^{a+b+c+J} sqrt((2e+1)(2f+1)) {a b e; c J f}      (Racah recoupling coefficient)
    The script checks F against its definition by Clebsch-Gordan sums for the small spins and its orthogonality for all.

Run here (sympy is in the image): python scripts/make_su2_golden.py
"""
import os
import sys

import numpy as np
from sympy import Rational
from sympy.physics.wigner import clebsch_gordan, wigner_6j

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHI = 512


def leg(twoj_list, chi, sigma=1.5):
    """multiplicities m_j ~ exp(-j^2 / 2 sigma^2) with sum m_j (2j+1) as close to chi as the rounding allows"""
    tj = np.array(twoj_list)
    w = np.exp(-(tj / 2.0) ** 2 / (2 * sigma ** 2))
    m = np.floor(chi * w / np.sum(w * (tj + 1))).astype(int)
    keep = m > 0
    return tj[keep], m[keep]


def racah_F(a2, b2, c2, J2, e2, f2):
    a, b, c, J, e, f = (Rational(x, 2) for x in (a2, b2, c2, J2, e2, f2))
    sj = wigner_6j(a, b, e, c, J, f)
    val = (-1) ** int(a + b + c + J) * ((2 * e + 1) * (2 * f + 1)) ** Rational(1, 2) * sj
    return float(val)


def F_by_clebsch_gordan(a2, b2, c2, J2, e2, f2):
    """<(ab)e, c; J M | a, (bc)f; J M> as a sum of four Clebsch-Gordan coefficients (M = J)"""
    a, b, c, J, e, f = (Rational(x, 2) for x in (a2, b2, c2, J2, e2, f2))
    M = J
    tot = 0
    rng_ = lambda j: [-j + k for k in range(int(2 * j) + 1)]
    for ma in rng_(a):
        for mb in rng_(b):
            mc = M - ma - mb
            if abs(mc) > c:
                continue
            me, mf = ma + mb, mb + mc
            if abs(me) > e or abs(mf) > f:
                continue
            tot += (clebsch_gordan(a, b, e, ma, mb, me) * clebsch_gordan(e, c, J, me, mc, M)
                    * clebsch_gordan(b, c, f, mb, mc, mf) * clebsch_gordan(a, f, J, ma, mf, M))
    return float(tot)


def spin_range(x2, y2):
    return range(abs(x2 - y2), x2 + y2 + 1, 2)


def charged_leg(chi, sigma_n=2.0, sigma_j=1.5, nmax=8, jmax2=8):
    """SU(2) x U(1) leg (spinful fermions): sectors (n, 2j) with n + 2j even, Gaussian weights, sum m (2j+1) ~ chi"""
    secs, w = [], []
    for n in range(-nmax, nmax + 1):
        for j2 in range(0, jmax2 + 1):
            if (n + j2) % 2 == 0:
                secs.append((n, j2))
                w.append(np.exp(-n * n / (2 * sigma_n ** 2) - (j2 / 2.0) ** 2 / (2 * sigma_j ** 2)))
    secs, w = np.array(secs), np.array(w)
    m = np.floor(chi * w / np.sum(w * (secs[:, 1] + 1))).astype(int)
    keep = m > 0
    return secs[keep], m[keep]


SITE = [(0, 0), (2, 0), (1, 1)]        # Hubbard site: empty, doubly occupied (spin 0), singly occupied (spin 1/2)


def hubbard_structure(chi):
    """compose list and F-move table of a two-site update with SU(2) x U(1) symmetry (many small coupled sectors)"""
    secs, m = charged_leg(chi)
    mult = {tuple(s): int(x) for s, x in zip(secs.tolist(), m.tolist())}
    comp = []
    for (N, J2), K in mult.items():                       # coupled sector of the middle bond
        rows = sum(mult.get((N - n_p, a2), 0) for n_p, p2 in SITE for a2 in spin_range(J2, p2))
        if rows and K:
            comp.append((N, J2, rows, K, rows))           # (left and right bond carry the same leg)
    rows_old, rows_new, terms = [], [], []
    f_cache = {}
    for (N, J2) in mult:                                  # coupled sector = right bond sector
        off_old = off_new = 0
        for (na, a2), ma in mult.items():
            for nb, b2 in SITE:
                for nc, c2 in SITE:
                    if na + nb + nc != N:
                        continue
                    es = [e2 for e2 in spin_range(a2, b2) if J2 in spin_range(e2, c2)]
                    fs = [f2 for f2 in spin_range(b2, c2) if J2 in spin_range(a2, f2)]
                    if not es:
                        assert not fs
                        continue
                    assert len(es) == len(fs), (a2, b2, c2, J2)
                    key = (a2, b2, c2, J2)
                    if key not in f_cache:
                        Fm = np.array([[racah_F(a2, b2, c2, J2, e2, f2) for f2 in fs] for e2 in es])
                        assert np.abs(Fm @ Fm.T - np.eye(len(es))).max() < 1e-12, key
                        f_cache[key] = Fm
                    Fm = f_cache[key]
                    for e2 in es:
                        rows_old.append((N, J2, na, a2, nb, b2, nc, c2, e2, off_old, ma))
                        off_old += ma
                    for k, f2 in enumerate(fs):
                        rows_new.append((N, J2, na, a2, nb, b2, nc, c2, f2, off_new, ma))
                        off_new += ma
                        for i, e2 in enumerate(es):
                            if abs(Fm[i, k]) > 0:
                                terms.append(((N, J2, na, a2, nb, b2, nc, c2, e2, f2), Fm[i, k]))
        assert off_old == off_new
    return secs, m, comp, rows_old, rows_new, terms


def main():
    L2, mL = leg(range(0, 14, 2), CHI)         # integer spins on the outer bonds
    M2, mM = leg(range(1, 14, 2), CHI)         # half-integer spins on the middle bond
    R2, mR = L2.copy(), mL.copy()
    mult_L = dict(zip(L2.tolist(), mL.tolist()))
    mult_M = dict(zip(M2.tolist(), mM.tolist()))
    mult_R = dict(zip(R2.tolist(), mR.tolist()))
    # ---- compose list: one GEMM per coupled sector J of the middle bond
    comp = []
    for J2, K in zip(M2.tolist(), mM.tolist()):
        rows = mult_L.get(J2 - 1, 0) + mult_L.get(J2 + 1, 0)
        cols = mult_R.get(J2 - 1, 0) + mult_R.get(J2 + 1, 0)
        if rows and cols and K:
            comp.append((J2, rows, K, cols))
    # ---- F-move table of T: [(a 1/2) 1/2 -> J] -> [a (1/2 1/2) -> J], coupled sector J = right bond spin (integer)
    rows_old, rows_new, terms = [], [], []      # rows_*: (J2, a2, intermediate2, offset, size) of every tree block
    for J2 in R2.tolist():
        off_old = off_new = 0
        for a2 in L2.tolist():
            es = [e2 for e2 in (a2 - 1, a2 + 1) if e2 >= 0 and abs(e2 - 1) <= J2 <= e2 + 1]
            fs = [f2 for f2 in (0, 2) if abs(a2 - f2) <= J2 <= a2 + f2]
            if not es:
                assert not fs
                continue
            assert len(es) == len(fs), (a2, J2, es, fs)        # both bases span the same multiplicity space
            Fm = np.array([[racah_F(a2, 1, 1, J2, e2, f2) for f2 in fs] for e2 in es])
            assert np.abs(Fm @ Fm.T - np.eye(len(es))).max() < 1e-12, (a2, J2)
            if a2 <= 4:
                Fc = np.array([[F_by_clebsch_gordan(a2, 1, 1, J2, e2, f2) for f2 in fs] for e2 in es])
                assert np.abs(Fm - Fc).max() < 1e-12, (a2, J2, Fm, Fc)
            for e2 in es:
                rows_old.append((J2, a2, e2, off_old, mult_L[a2]))
                off_old += mult_L[a2]
            for k, f2 in enumerate(fs):
                rows_new.append((J2, a2, f2, off_new, mult_L[a2]))
                off_new += mult_L[a2]
                for i, e2 in enumerate(es):
                    if abs(Fm[i, k]) > 0:
                        terms.append((J2, a2, e2, f2, Fm[i, k]))
        assert off_old == off_new
    hsecs, hm, hcomp, hold, hnew, hterms = hubbard_structure(CHI)
    out = os.path.join(ROOT, 'tests', 'golden', 'su2_chi512.npz')
    np.savez_compressed(out, chi=CHI, L2=L2, mL=mL, M2=M2, mM=mM, R2=R2, mR=mR,
                        compose=np.array(comp, dtype=np.int64),
                        rows_old=np.array(rows_old, dtype=np.int64), rows_new=np.array(rows_new, dtype=np.int64),
                        terms_idx=np.array([t[:4] for t in terms], dtype=np.int64), terms_coeff=np.array([t[4] for t in terms]),
                        h_sectors=hsecs, h_mults=hm, h_compose=np.array(hcomp, dtype=np.int64),
                        h_rows_old=np.array(hold, dtype=np.int64), h_rows_new=np.array(hnew, dtype=np.int64),
                        h_terms_idx=np.array([t[0] for t in hterms], dtype=np.int64), h_terms_coeff=np.array([t[1] for t in hterms]))
    dimL = int(np.sum(mL * (L2 + 1)))
    print(f'wrote {out}')
    print(f'  spin-1/2 chain, SU(2): outer bond {len(L2)} sectors, sum m(2j+1) = {dimL}; middle bond {len(M2)} sectors; '
          f'{len(comp)} coupled-sector GEMMs {comp}; tree move: {len(rows_new)} new tree blocks from {len(terms)} terms')
    print(f'  Hubbard chain, SU(2) x U(1): bond {len(hsecs)} sectors, sum m(2j+1) = {int(np.sum(hm * (hsecs[:, 1] + 1)))}; '
          f'{len(hcomp)} coupled-sector GEMMs, rows {min(c[2] for c in hcomp)}..{max(c[2] for c in hcomp)}, '
          f'K {min(c[3] for c in hcomp)}..{max(c[3] for c in hcomp)}; tree move: {len(hnew)} new tree blocks from {len(hterms)} terms')


if __name__ == '__main__':
    sys.exit(main())
