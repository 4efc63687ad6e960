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


class _NotFlat(Exception):
    """The vector left the block structure the flat representation was built for (caught by LanczosGroundState.run)."""


class _TensorOps:
    """Vector operations of the recurrences on block-sparse tensors (one launch per block LIST, per-block host work)."""

    def __init__(self, bb, H):
        self.bb, self.H = bb, H

    def enter(self, t):
        return t

    def leave(self, w):
        return w

    def matvec(self, w):
        return self.H.matvec(w)

    def norm(self, w):
        return ab.norm(self.bb, w)

    def inner(self, v, w):
        return ab.inner(self.bb, v, w)

    def scale(self, a, w):
        return ab.scale(self.bb, a, w)

    def lincomb(self, a, w, b, v):
        return ab.linear_combination(self.bb, a, w, b, v)


class _FlatOps:
    """The same operations on Krylov vectors kept as ONE contiguous float64 pool each: the block
    offsets of the structure are computed once, every ``scale / axpy / inner / norm`` of the recurrences is a
    single-descriptor launch over the whole pool -- no per-block host work, one descriptor copy.  The pools are zero
    between blocks (32-element alignment gaps), so reductions over the flat range are exact.  Only the operator
    application sees tensors: its input is a list of views into the pool, its output is copied back in one batched
    launch.  Real float64 tensors on the HIP backend only; anything else uses :class:`_TensorOps`."""

    def __init__(self, bb, H, template, block_inds=None):
        """`template`: a tensor on the legs of the vectors; `block_inds` (default: the template's): the block table of the
        pool -- a superset of the template's when the operator creates blocks the start vector does not have."""
        from .block_backend import HipBlock, _c_strides
        from . import _lib
        import ctypes
        self.bb, self.H, self.t = bb, H, template
        self._HipBlock, self._lib, self._C = HipBlock, _lib, ctypes
        self.block_inds = template.block_inds if block_inds is None else block_inds
        self.shapes = [template.block_shape(r) for r in self.block_inds]
        self.strides = [_c_strides(sh) for sh in self.shapes]
        offs, tot = [], 0
        for sh in self.shapes:
            n = 1
            for x in sh:
                n *= x
            offs.append(tot)
            tot += (n + 31) // 32 * 32
        self.offs, self.total = offs, tot
        self.key = self.block_inds.tobytes()
        self.index = None     # block row -> position, built only if a tensor has fewer blocks than the pool
        # positions of the pool's blocks that any vector so far holds (the start vector's, plus what the operator produced):
        # the pool is laid out over every charge-allowed block, but the operator only ever sees -- and the result only carries --
        # this support, as the reference's tensors do (krylov_based.cpp works on tensors whose block tables grow the same way)
        self.support = set()

    @staticmethod
    def usable(bb, t) -> bool:
        return (hasattr(bb, 'ctx') and len(t.blocks) > 0 and not any(b.is_complex or b.is_bool for b in t.blocks)
                and type(bb).__name__ != 'DeferredBlockBackend')

    # -- pools
    def _alloc(self, zero):
        bb = self.bb
        buf = bb.ctx.empty(self.total)
        if zero:
            bb.ctx.sync_stream()
            self._lib.check(bb.lib.cyb_memset(bb.ctx.handle, self._C.c_void_p(buf.data_ptr()), 0, 8 * self.total))
        return buf

    def _views(self, buf, which=None):
        bb, mk = self.bb, self._HipBlock._trusted
        idx = range(len(self.offs)) if which is None else which
        return [mk(bb, buf, self.offs[i], self.shapes[i], self.strides[i], True) for i in idx]

    def enter(self, t):
        """tensor -> pool (one memset + one batched copy)."""
        buf = self._alloc(True)
        if t.block_inds.tobytes() == self.key:
            which = None
        else:   # fewer blocks than the template (missing blocks are zero); a block outside the template ends flat mode
            if self.index is None:
                self.index = {tuple(r): i for i, r in enumerate(self.block_inds.tolist())}
            try:
                which = [self.index[tuple(r)] for r in t.block_inds.tolist()]
            except KeyError:
                raise _NotFlat()
        views = self._views(buf, which)
        if any(v.shape != tuple(b.shape) or b.is_complex for v, b in zip(views, t.blocks)):
            raise _NotFlat()
        self.support.update(range(len(self.offs)) if which is None else which)
        self.bb.copy_many(list(zip(views, t.blocks)))
        return buf

    def leave(self, buf):
        t = self.t
        if len(self.support) == len(self.offs):
            return ab.AbelianTensor(t.symmetry, t.legs, self._views(buf), self.block_inds, t.num_codomain, t.labels)
        idx = sorted(self.support)     # (positions ascend with the lexsorted block table)
        return ab.AbelianTensor(t.symmetry, t.legs, self._views(buf, idx), self.block_inds[idx], t.num_codomain, t.labels)

    def matvec(self, buf):
        return self.enter(self.H.matvec(self.leave(buf)))

    # -- BLAS-1 over the flat range
    def _desc(self, x, y, out):
        arr = np.zeros(1, dtype=self._lib.VEC_DTYPE)
        arr['x'][0] = x.data_ptr()
        arr['y'][0] = y.data_ptr() if y is not None else 0
        arr['out'][0] = out.data_ptr() if out is not None else 0
        arr['n'][0] = self.total
        return arr

    def lincomb(self, a, w, b, v):
        bb = self.bb
        out = self._alloc(False)
        arr = self._desc(w, v, out)
        bb.ctx.sync_stream()
        self._lib.check(bb.lib.cyb_axpby_batched_f64(bb.ctx.handle, arr.ctypes.data_as(self._C.POINTER(self._lib.VecDesc)), 1,
                                                     float(a), float(b)))
        return out

    def scale(self, a, w):
        bb = self.bb
        out = self._alloc(False)
        arr = self._desc(w, None, out)
        bb.ctx.sync_stream()
        self._lib.check(bb.lib.cyb_axpby_batched_f64(bb.ctx.handle, arr.ctypes.data_as(self._C.POINTER(self._lib.VecDesc)), 1,
                                                     float(a), 0.0))
        return out

    def inner(self, v, w):
        bb = self.bb
        res = bb.ctx.empty(1)
        arr = self._desc(v, w, None)
        bb.ctx.sync_stream()
        self._lib.check(bb.lib.cyb_dot_batched_f64(bb.ctx.handle, arr.ctypes.data_as(self._C.POINTER(self._lib.VecDesc)), 1,
                                                   self._C.c_void_p(res.data_ptr())))
        return float(bb.ctx.d2h(res, 1, np.float64)[0])

    def norm(self, w):
        return float(np.sqrt(self.inner(w, w)))


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
        self.flat = o.get('flat', True)   # (not a reference option: Krylov vectors as flat pools where possible, _FlatOps)
        self.V = None
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
        V = self.V
        out = V.matvec(w)
        if self.E_shift is not None:
            out = V.lincomb(1.0, out, float(self.E_shift), self._cache[-1])
        return out

    def run(self):
        """The recurrences run on flat pools where the backend allows it (:class:`_FlatOps`); a vector that leaves the
        block structure of psi0 (an operator that creates blocks psi0 does not have) restarts them on tensors."""
        psi_in = self.psi0
        flat = bool(self.flat) and _FlatOps.usable(self.bb, psi_in)
        # pool structure: every block the charge rule allows on these legs (what an operator can create at most)
        allowed = ab.AbelianTensor.allowed_block_inds(psi_in.symmetry, psi_in.legs) if flat else None
        for use_flat in ([True, False] if flat else [False]):
            self.V = _FlatOps(self.bb, self.H, psi_in, allowed) if use_flat else _TensorOps(self.bb, self.H)
            self._h[:] = 0.0
            self.Es[:] = 0.0
            self._cache = []
            self._result_krylov = np.ones(1)
            try:
                self.psi0 = self.V.enter(psi_in)
                N = self._build_krylov()
                E0 = float(self.Es[N - 1, 0])
                if self.E_shift is not None:
                    E0 -= float(self.E_shift)
                if N == 1:
                    return E0, self.V.leave(self.psi0), N
                return E0, self.V.leave(self._calc_result_full(N)), N
            except _NotFlat:
                continue
            finally:
                self.psi0 = psi_in     # (the working copy is a pool buffer: a second run() starts from the caller's tensor again)
        raise RuntimeError('unreachable')

    def _build_krylov(self):
        V = self.V
        w = self.psi0
        beta = V.norm(w)
        if beta < self.cutoff:
            raise ValueError(f'Norm of self.psi0 too small: {beta}')
        self.psi0 = V.scale(1.0 / beta, w)
        performed = 0
        for k in range(self.N_max):
            w = V.scale(1.0 / beta, w)
            self._to_cache(w)
            w = self._matvec(w)
            alpha = float(np.real(V.inner(w, self._cache[-1])))   # krylov_based.cpp:861: inner(...).real()
            self._h[k, k] = alpha
            self._calc_result_krylov(k)
            w = V.lincomb(1.0, w, -alpha, self._cache[-1])
            if self.reortho:
                for v in self._cache[:-1]:
                    ov = V.inner(v, w)
                    w = V.lincomb(1.0, w, -ov, v)
            elif k > 0:
                w = V.lincomb(1.0, w, -beta, self._cache[-2])
            beta = V.norm(w)
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
        V = self.V
        vf = self._result_krylov
        if not (N == len(vf) and len(vf) > 1):
            raise RuntimeError('KrylovBased._calc_result_full: expected N == len(vf) > 1')
        psif = V.scale(float(vf[0]), self.psi0)
        len_cache = len(self._cache)
        for k in range(1, min(len_cache + 1, N)):
            psif = V.lincomb(1.0, psif, float(vf[N - k]), self._cache[len_cache - k])
        self._cache = []
        psif = self._rebuild_krylov_for_result_full(psif, N - len_cache - 1)
        nrm = V.norm(psif)
        return V.scale(1.0 / nrm, psif)

    def _rebuild_krylov_for_result_full(self, psif, n_rebuild):
        """Vectors that fell out of the cache are regenerated from psi0 (krylov_based.cpp:896-920)."""
        V = self.V
        vf = self._result_krylov
        w = self.psi0
        beta = 0.0
        for k in range(max(n_rebuild, 0)):
            self._to_cache(w)
            w = self._matvec(w)
            alpha = self._h[k, k]
            w = V.lincomb(1.0, w, -alpha, self._cache[-1])
            if self.reortho:
                for v in self._cache[:-1]:
                    ov = V.inner(v, w)
                    w = V.lincomb(1.0, w, -ov, v)
            elif k > 0:
                w = V.lincomb(1.0, w, -beta, self._cache[-2])
            beta = self._h[k, k + 1]
            w = V.scale(1.0 / beta, w)
            psif = V.lincomb(1.0, psif, float(vf[k + 1]), w)
        return psif


def lanczos(bb, H, psi, options=None):
    """(E0, psi0, N) -- krylov_based.cpp:1022-1025."""
    return LanczosGroundState(bb, H, psi, options).run()
