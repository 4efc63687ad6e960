"""The Lanczos matvec group of the DMRG inner loop on the grouped device path.

* :class:`HEffective` -- the two-site effective Hamiltonian of the reference's DMRG toycode
  (/root/reference/toycodes/tenpy_toycodes/d_dmrg.py:55-86): ``matvec`` = 4 ``compose`` + 4
  ``permute_legs``.  Every compose is ONE grouped-GEMM launch (plus at most one batched copy that makes
  permuted operands contiguous); the host-side sector matching of a compose is computed once per
  operand structure and reused by all later matvecs (the block tables do not change between Lanczos
  iterations).
* :class:`LanczosGroundState` / :func:`lanczos` -- the reference's Lanczos iteration
  (/root/reference/src/tensors/krylov_based.cpp:803-946, options :276-288): same recurrences,
  same convergence test, same result assembly; ``inner`` / ``norm`` / ``axpy`` run as one launch
  each over the whole block list, the (k+1) x (k+1) tridiagonal problem stays on the host as in the
  reference (numpy.linalg.eigh, :922-946).

Leg orders used here (signs: + ket-like, - dual):
    theta [vL, p0, p1, vR]
    LP    [vL', wL, vL*]          W1 [p0', wC, p0*, wL*]
    W2    [p1', wR, p1*, wC*]     RP [wR*, vR*, vR']
``compose(a, b, k)`` contracts the last k legs of a with the first k legs of b, a's in reversed order.
"""
from __future__ import annotations

import numpy as np

from . import abelian as ab


class HEffective:
    """theta' = H_eff theta for a two-site DMRG update (d_dmrg.py:74-86)."""

    def __init__(self, bb, LP, W1, W2, RP, replay: bool = True, cache: dict | None = None):
        """`cache`: a dict the caller keeps across operators (a DMRG run keeps one for all bonds and sweeps).  The
        recorded launch sequences are relocatable in ALL five operands (environments, MPO tensors and the vector), so a
        bond whose block layouts have been seen before -- every bond from the second sweep on, once the sector
        structure has settled -- is served by replays only."""
        self.bb, self.LP, self.W1, self.W2, self.RP = bb, LP, W1, W2, RP
        self._plans = {}
        self.flops_per_matvec = None
        # recorded launch sequences, one per block layout of the operands (cyten_amd/replay.py); only a backend that
        # issues C-ABI launches can be recorded
        self._recordings = (cache if cache is not None else {}) if (replay and hasattr(bb, 'ctx')) else None
        self._op_layout = None
        self.n_replayed = 0
        self.n_recorded = 0

    def _compose(self, tag, a, b, k):
        key = (tag, a.block_inds.tobytes(), b.block_inds.tobytes())
        plan = self._plans.get(key)
        if plan is None:
            plan = ab.compose_plan(a, b, k)
            self._plans[key] = plan
        na_keep = a.nlegs - k
        if not plan.pairs:
            return ab.AbelianTensor(a.symmetry, plan.legs, [], plan.res_block_inds, na_keep), 0.0
        a2, b2 = ab._compose_operands(self.bb, a, b, k, plan)
        outs = self.bb.matrix_dot_grouped([[(a2[i], b2[j]) for i, j in g] for g in plan.pairs])
        blocks = [self.bb.reshape(o, shp) for o, shp in zip(outs, plan.res_shapes)]
        return ab.AbelianTensor(a.symmetry, plan.legs, blocks, plan.res_block_inds, na_keep), plan.flops

    def matvec(self, theta):
        """H_eff theta.  The first application to an input of a given block layout runs the ordinary path while its
        allocations and launches are recorded; later applications replay them with the pointers rewritten."""
        if self._recordings is None:
            return self._matvec(theta)
        from .replay import apply_recorded, tensor_layout
        if self._op_layout is None:
            bufs, sizes = [], {}
            sig = tensor_layout([self.LP, self.W1, self.W2, self.RP], bufs, sizes)
            self._op_layout = (sig, bufs, sizes)
        out, how, rec = apply_recorded(self.bb, self._recordings, 'heff', lambda: self._matvec(theta), [theta],
                                       fixed=self._op_layout)
        if how == 'recorded':
            self.n_recorded += 1
            rec.flops = self.flops_per_matvec
        elif how == 'replayed':
            self.n_replayed += 1
            self.flops_per_matvec = rec.flops
        return out

    def _matvec(self, theta):
        bb = self.bb
        flops = 0.0
        x, f = self._compose('LP', self.LP, theta, 1)                 # [vL', wL, p0, p1, vR]
        flops += f
        x = ab.permute_legs(bb, x, [1, 2, 3, 4, 0])                   # [wL, p0, p1, vR, vL']
        x, f = self._compose('W1', self.W1, x, 2)                     # [p0', wC, p1, vR, vL']
        flops += f
        x = ab.permute_legs(bb, x, [1, 2, 3, 4, 0])                   # [wC, p1, vR, vL', p0']
        x, f = self._compose('W2', self.W2, x, 2)                     # [p1', wR, vR, vL', p0']
        flops += f
        x = ab.permute_legs(bb, x, [3, 4, 0, 2, 1])                   # [vL', p0', p1', vR, wR]
        x, f = self._compose('RP', x, self.RP, 2)                     # [vL', p0', p1', vR']
        flops += f
        self.flops_per_matvec = flops
        x.num_codomain = theta.num_codomain
        return x


class LanczosGroundState:
    """Lanczos for the lowest eigenvector of a Hermitian ``H`` (krylov_based.cpp:803-946).

    Options (defaults of krylov_based.cpp:276-288, 808-809): N_min=2, N_max=20, P_tol=1e-14,
    min_gap=1e-12, reortho=False, cutoff=100*eps, E_tol=inf, E_shift=None, N_cache=N_max."""

    def __init__(self, bb, H, psi0, options=None):
        o = dict(options or {})
        self.bb, self.H, self.psi0 = bb, H, psi0
        self.N_min = int(o.get('N_min', 2))
        self.N_max = int(o.get('N_max', 20))
        self.P_tol = float(o.get('P_tol', 1e-14))
        self.min_gap = float(o.get('min_gap', 1e-12))
        self.reortho = bool(o.get('reortho', False))
        self.cutoff = float(o.get('cutoff', np.finfo(np.float64).eps * 100))
        self.E_tol = float(o.get('E_tol', np.inf))
        self.E_shift = o.get('E_shift', None)
        self.N_cache = int(o.get('N_cache', self.N_max))
        if self.N_min < 2:
            raise ValueError('Should perform at least 2 steps.')
        if self.N_cache < 2:
            raise ValueError('Need to cache at least two vectors.')
        self._h = np.zeros((self.N_max + 1, self.N_max + 1))
        self.Es = np.zeros((self.N_max, self.N_max))
        self._cache = []
        self._result_krylov = np.ones(1)

    # -- small helpers -------------------------------------------------------------------------
    def _to_cache(self, w):
        self._cache.append(w)
        if len(self._cache) > self.N_cache:
            self._cache.pop(0)

    def _matvec(self, w):
        w = self.H.matvec(w)
        if self.E_shift is not None:
            w = ab.linear_combination(self.bb, 1.0, w, float(self.E_shift), self._cache[-1])
        return w

    def run(self):
        N = self._build_krylov()
        E0 = float(self.Es[N - 1, 0])
        if self.E_shift is not None:
            E0 -= float(self.E_shift)
        if N == 1:
            return E0, self.psi0, N
        return E0, self._calc_result_full(N), N

    def _build_krylov(self):
        bb = self.bb
        w = self.psi0
        beta = ab.norm(bb, w)
        if beta < self.cutoff:
            raise ValueError(f'Norm of self.psi0 too small: {beta}')
        self.psi0 = ab.scale(bb, 1.0 / beta, w)
        performed = 0
        for k in range(self.N_max):
            w = ab.scale(bb, 1.0 / beta, w)
            self._to_cache(w)
            w = self._matvec(w)
            alpha = float(np.real(ab.inner(bb, w, self._cache[-1])))   # krylov_based.cpp:861: inner(...).real()
            self._h[k, k] = alpha
            self._calc_result_krylov(k)
            w = ab.linear_combination(bb, 1.0, w, -alpha, self._cache[-1])
            if self.reortho:
                for v in self._cache[:-1]:
                    ov = ab.inner(bb, v, w)
                    w = ab.linear_combination(bb, 1.0, w, -ov, v)
            elif k > 0:
                w = ab.linear_combination(bb, 1.0, w, -beta, self._cache[-2])
            beta = ab.norm(bb, w)
            self._h[k, k + 1] = self._h[k + 1, k] = beta
            performed = k + 1
            if abs(beta) < self.cutoff or (k + 1 >= self.N_min and self._converged(k)):
                break
        return performed

    def _converged(self, k):
        v0k = self._result_krylov[k]
        ritz_res = abs(v0k) * abs(self._h[k, k + 1])
        gap = max(self.Es[k, 1] - self.Es[k, 0], self.min_gap)
        P_err = (ritz_res / gap) ** 2
        Delta_E0 = self.Es[k - 1, 0] - self.Es[k, 0]
        return P_err < self.P_tol and Delta_E0 < self.E_tol

    def _calc_result_krylov(self, k):
        if k == 0:
            self.Es[0, 0] = self._h[0, 0]
            self._result_krylov = np.ones(1)
            return
        n = k + 1
        E_kr, v_kr = np.linalg.eigh(self._h[:n, :n])
        self.Es[k, :n] = E_kr
        self._result_krylov = v_kr[:, 0].copy()

    def _calc_result_full(self, N):
        bb = self.bb
        vf = self._result_krylov
        if not (N == len(vf) and len(vf) > 1):
            raise RuntimeError('KrylovBased._calc_result_full: expected N == len(vf) > 1')
        psif = ab.scale(bb, float(vf[0]), self.psi0)
        len_cache = len(self._cache)
        for k in range(1, min(len_cache + 1, N)):
            psif = ab.linear_combination(bb, 1.0, psif, float(vf[N - k]), self._cache[len_cache - k])
        self._cache = []
        psif = self._rebuild_krylov_for_result_full(psif, N - len_cache - 1)
        nrm = ab.norm(bb, psif)
        return ab.scale(bb, 1.0 / nrm, psif)

    def _rebuild_krylov_for_result_full(self, psif, n_rebuild):
        """Vectors that fell out of the cache are regenerated from psi0 (krylov_based.cpp:896-920)."""
        bb = self.bb
        vf = self._result_krylov
        w = self.psi0
        beta = 0.0
        for k in range(max(n_rebuild, 0)):
            self._to_cache(w)
            w = self._matvec(w)
            alpha = self._h[k, k]
            w = ab.linear_combination(bb, 1.0, w, -alpha, self._cache[-1])
            if self.reortho:
                for v in self._cache[:-1]:
                    ov = ab.inner(bb, v, w)
                    w = ab.linear_combination(bb, 1.0, w, -ov, v)
            elif k > 0:
                w = ab.linear_combination(bb, 1.0, w, -beta, self._cache[-2])
            beta = self._h[k, k + 1]
            w = ab.scale(bb, 1.0 / beta, w)
            psif = ab.linear_combination(bb, 1.0, psif, float(vf[k + 1]), w)
        return psif


def lanczos(bb, H, psi, options=None):
    """(E0, psi0, N) -- krylov_based.cpp:1022-1025."""
    return LanczosGroundState(bb, H, psi, options).run()
