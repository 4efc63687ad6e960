"""Seeded inputs shared by the CPU and GPU tests of the remaining AbelianBackend callers (partial_compose, _mask_contract,
qr / lq of two-leg tensors): plain-data tensors (cyten_amd.workloads.TensorSpec) and the dense contractions they must equal."""
import numpy as np

from cyten_amd import workloads as wl

SEED_SHIFT = 0          # scripts/callers_fuzz.py moves it: the same generators, other charges / multiplicities / fill patterns


def legs_u1(rng, n, sign):
    qs = np.sort(rng.choice(np.arange(-1, 2), size=n, replace=False))      # (few charges: the fused charges of different legs overlap)
    return wl.make_leg((0,), qs[:, None], rng.integers(1, 5, n), sign)


def partial_compose_cases(seed=7):
    """[(a, b, a_first_leg, dense_result)]: b sits on consecutive codomain legs of a (its domain is contracted) or on
    consecutive domain legs (its codomain is contracted).  Symmetries: U(1), Z3, U(1) x Z2; missing blocks in both."""
    rng = np.random.default_rng(seed + SEED_SHIFT)
    out = []
    for moduli in [(0,), (3,), (0, 2)]:
        def leg(n, sign):
            if moduli == (0,):
                return legs_u1(rng, n, sign)
            if moduli == (3,):
                return wl.make_leg(moduli, np.arange(3)[:n, None], rng.integers(1, 4, n), sign)
            secs = [(q, z) for q in (-1, 0, 1) for z in (0, 1)]
            pick = rng.choice(len(secs), size=n, replace=False)
            return wl.make_leg(moduli, np.array(secs)[pick], rng.integers(1, 4, n), sign)
        x, y, z, w, v, u = leg(3, +1), leg(3, +1), leg(2, +1), leg(3, -1), leg(2, -1), leg(3, +1)
        # a: codomain [x, y, z], domain flat [w, v]   (flat legs = codomain + reversed domain)
        a = wl.random_tensor(moduli, [x, y, z, w, v], rng, num_codomain=3, fill=0.8)
        # case 1: b on a's codomain legs (y, z): b.domain = (y, z) -> flat reversed [z(-), y(-)], b.codomain = [u]
        b1 = wl.random_tensor(moduli, [u, wl.flip(z), wl.flip(y)], rng, num_codomain=1, fill=0.8)
        out.append((a, b1, 1))
        # case 2: b on a's domain legs: a's flat domain legs [w(-), v(-)]; b.codomain must carry the duals, flat order [v(+), w(+)]
        # (compose pairs a_perm.legs[-1-i] with b.legs[i]); b.domain = one new leg
        b2 = wl.random_tensor(moduli, [wl.flip(v), wl.flip(w), wl.flip(u)], rng, num_codomain=2, fill=0.8)
        out.append((a, b2, 3))
        # case 3: a single contracted leg in the middle of the codomain, two added legs
        b3 = wl.random_tensor(moduli, [u, wl.flip(u), wl.flip(y)], rng, num_codomain=2, fill=0.9)
        out.append((a, b3, 1))
    return out


def dense_partial_compose(a_dense, b_dense, a, b, a_first_leg):
    """the contraction partial_compose stands for, on dense arrays over the flat legs"""
    a_n_cod, b_n_cod, b_n = a.num_codomain, b.num_codomain, len(b.legs)
    if a_first_leg < a_n_cod:
        nc, add = b_n - b_n_cod, list(range(b_n_cod))
        b_contr = [b_n - 1 - j for j in range(nc)]             # a flat[first + j] <-> b flat[b_n - 1 - j]
    else:
        nc, add = b_n_cod, list(range(b_n_cod, b_n))
        b_contr = [nc - 1 - j for j in range(nc)]              # a flat[first + j] <-> b flat[nc - 1 - j]
    a_contr = [a_first_leg + j for j in range(nc)]
    res = np.tensordot(a_dense, b_dense, (a_contr, b_contr))   # a's kept legs, then b's kept legs (= `add`, ascending)
    n_keep = a_dense.ndim - nc
    perm = list(range(a_first_leg)) + list(range(n_keep, n_keep + len(add))) + list(range(a_first_leg, n_keep))
    return np.transpose(res, perm)


def two_leg_cases(seed=11):
    """two-leg tensors [cod(+), dom(-)] with missing blocks and with sectors only one leg holds"""
    rng = np.random.default_rng(seed + SEED_SHIFT)
    out = []
    for n_c, n_d in [(4, 4), (5, 3), (3, 5)]:
        qc = np.sort(rng.choice(np.arange(-3, 4), size=n_c, replace=False))
        qd = np.sort(rng.choice(np.arange(-3, 4), size=n_d, replace=False))
        cod = wl.make_leg((0,), qc[:, None], rng.integers(1, 7, n_c), +1)
        dom = wl.make_leg((0,), qd[:, None], rng.integers(1, 7, n_d), -1)
        out.append(wl.random_tensor((0,), [cod, dom], rng, num_codomain=1, fill=0.7))
    return out
