"""Record / replay of the device work of a block-structured operator application.

The Lanczos matvec of DMRG applies the SAME sequence of grouped launches (compose GEMMs, leg-rotation copies) to
vectors of the SAME block structure; only the buffers differ from one application to the next.  The host work of an
application -- hundreds of block views, descriptor marshalling -- is therefore recorded ONCE: every device allocation
and every C-ABI launch with its descriptor arrays, each pointer resolved to (buffer, byte offset).  A replay performs
the same allocations, rewrites the pointer columns with a handful of vectorised numpy operations and issues the same
launches: ~0.3 ms of host time instead of ~4.4 ms per H_eff matvec at chi=4096, which was the bottleneck once the
kernels of a matvec had come down to 3.5 ms.  (The role a HIP graph plays for a fixed launch sequence, but with
relocatable buffers: a captured graph would pin the addresses of the Krylov vectors.)

Only launches whose descriptors are fully understood are replayed (strided copies, grouped GEMM enqueues, the operand
expansion of complex products, memsets);
a recording that meets anything else, or an input whose block layout differs from the recorded one, falls back to the
ordinary path.  Nothing here computes on block data.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_PTR_FIELDS = {
    'cyb_copy_strided_batched': (('dst', 'src'),),
    'cyb_gemm_grouped_enqueue_f64': (('C',), ('A', 'B')),
    'cyb_complex_expand_batched_f64': (('src', 'dst'),),
}
_STRUCTS = {   # ctypes element types of the descriptor arrays, in argument order
    'cyb_copy_strided_batched': ('CopyDesc',),
    'cyb_gemm_grouped_enqueue_f64': ('GemmProb', 'GemmSeg'),
    'cyb_complex_expand_batched_f64': ('CExpandDesc',),
}
_PASS_THROUGH = {'cyb_ctx_set_stream', 'cyb_last_error'}


class _RecordingLib:
    """Proxy of the loaded library: forwards every call and reports it to the recorder."""

    def __init__(self, real, rec):
        self._real, self._rec = real, rec

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in _PASS_THROUGH:
            return fn

        def wrapper(*args):
            self._rec._on_call(name, args)
            return fn(*args)
        return wrapper


class Recording:
    """The device work of one operator application (see module docstring).

    `external`: (data_ptr, nbytes) of the input buffers that change between applications, in a canonical order.
    Every pointer of a recorded launch is resolved AT THE TIME OF THE CALL to (source, byte offset): sources
    0..len(external)-1 are the inputs, then the allocations made so far, most recent first (the caching allocator
    hands the address range of a freed temporary out again, and the live owner of an address is always the latest
    allocation that contains it); anything else -- the fixed operands of the operator -- stays an absolute address."""

    def __init__(self, bb, external):
        self.bb = bb
        self.n_external = len(external)
        self._ranges = [(int(p), int(p) + int(b)) for p, b in external]   # sources, in order of creation
        self.plan = []            # ('alloc', n, dtype) | ('call', name, arrays, scalars, fields) | ('memset', ...)
        self.valid = True
        self.reason = ''

    # ---- recording
    def _resolve(self, ptrs):
        ptrs = np.asarray(ptrs, dtype=np.uint64)
        src = np.full(ptrs.shape, -1, dtype=np.int64)
        off = ptrs.copy()
        todo = np.ones(ptrs.shape, dtype=bool)
        for i in range(len(self._ranges) - 1, -1, -1):
            lo, hi = self._ranges[i]
            hit = todo & (ptrs >= lo) & (ptrs < hi)
            if hit.any():
                src[hit] = i
                off[hit] = ptrs[hit] - np.uint64(lo)
                todo &= ~hit
        return src, off

    def _on_alloc(self, n, dtype, tensor):
        self.plan.append(('alloc', int(n), dtype))
        self._ranges.append((tensor.data_ptr(), tensor.data_ptr() + tensor.numel() * tensor.element_size()))

    def _on_call(self, name, args):
        if name in _PTR_FIELDS:
            arrays, scalars = [], []
            for a in args[1:]:  # args[0] is the context handle
                arr = getattr(a, '_arr', None)
                if arr is not None:
                    arrays.append(arr.copy())
                else:
                    scalars.append(a)
            fields = [[(nm,) + self._resolve(arr[nm]) for nm in names] for arr, names in zip(arrays, _PTR_FIELDS[name])]
            self.plan.append(('call', name, arrays, scalars, fields))
        elif name == 'cyb_memset':
            src, off = self._resolve([int(args[1].value)])
            self.plan.append(('memset', int(src[0]), int(off[0]), int(args[1].value), int(args[2]), int(args[3])))
        else:
            self.valid = False
            self.reason = f'{name} is not replayable'

    def record(self, fn):
        """Run ``fn()`` while recording the allocations and launches it makes through this backend."""
        bb, ctx = self.bb, self.bb.ctx
        real_lib, real_ctx_lib, real_empty = bb.lib, ctx.lib, ctx.empty

        def empty(n, dtype='float64'):
            t = real_empty(n, dtype)
            self._on_alloc(n, dtype, t)
            return t
        bb.lib = _RecordingLib(real_lib, self)
        ctx.lib = _RecordingLib(real_ctx_lib, self)
        ctx.empty = empty
        try:
            return fn()
        finally:
            bb.lib, ctx.lib = real_lib, real_ctx_lib
            del ctx.empty

    def locate(self, ptr):
        """(index of the recorded allocation, byte offset) of an address that is live at the end of the recording."""
        for i in range(len(self._ranges) - 1, self.n_external - 1, -1):
            lo, hi = self._ranges[i]
            if lo <= ptr < hi:
                return i - self.n_external, ptr - lo
        return None

    @property
    def n_allocs(self):
        return len(self._ranges) - self.n_external

    # ---- replay
    def replay(self, external_ptrs):
        """Re-issue the recorded work with the inputs at `external_ptrs`; returns the tensors of the allocations."""
        bb, ctx = self.bb, self.bb.ctx
        lib, handle = bb.lib, ctx.handle
        ctx.sync_stream()
        bases = np.zeros(self.n_external + self.n_allocs, dtype=np.uint64)
        bases[:self.n_external] = external_ptrs
        tensors = []
        k = self.n_external
        for ev in self.plan:
            if ev[0] == 'alloc':
                t = ctx.empty(ev[1], ev[2])
                tensors.append(t)
                bases[k] = t.data_ptr()
                k += 1
            elif ev[0] == 'memset':
                _, src, off, absolute, value, nbytes = ev
                ptr = int(bases[src]) + off if src >= 0 else absolute
                _lib.check(lib.cyb_memset(handle, C.c_void_p(ptr), value, nbytes))
            else:
                _, name, arrays, scalars, fields = ev
                for arr, flds in zip(arrays, fields):
                    for nm, src, off in flds:
                        moved = src >= 0
                        if moved.any():
                            col = arr[nm]
                            col[moved] = bases[src[moved]] + off[moved]
                if name == 'cyb_gemm_grouped_enqueue_f64':   # (probs, n_probs, segs, n_segs)
                    probs = arrays[0].ctypes.data_as(C.POINTER(_lib.GemmProb))
                    segs = arrays[1].ctypes.data_as(C.POINTER(_lib.GemmSeg))
                    _lib.check(lib.cyb_gemm_grouped_enqueue_f64(handle, probs, scalars[0], segs, scalars[1]))
                else:                                         # (descs, scalars...)
                    ptr = arrays[0].ctypes.data_as(C.POINTER(getattr(_lib, _STRUCTS[name][0])))
                    _lib.check(getattr(lib, name)(handle, ptr, *scalars))
        return tensors


# ---------------------------------------------------------------------------------------------------------------------
# recorded application of a function of block-sparse tensors
# ---------------------------------------------------------------------------------------------------------------------

MAX_CACHED = 512   # recordings kept per cache dict (oldest dropped first)


def tensor_layout(tensors, bufs, sizes):
    """Hashable description of where the blocks of `tensors` sit: per tensor its symmetry, legs, block table and per
    block (buffer number, offset, shape, strides).  `bufs` (base addresses in order of first appearance) and `sizes`
    (address -> bytes) are extended in place, so layouts of several tensor groups can share one numbering."""
    index = {p: i for i, p in enumerate(bufs)}
    sig = []
    for t in tensors:
        blocks = []
        for blk in t.blocks:
            p = blk.buf.data_ptr()
            i = index.get(p)
            if i is None:
                i = index[p] = len(bufs)
                bufs.append(p)
                sizes[p] = blk.buf.numel() * blk.buf.element_size()
            blocks.append((i, blk.offset, blk.shape, blk.strides, blk.is_complex))
        legs = tuple((l.sectors.tobytes(), l.mults.tobytes(), l.sign) for l in t.legs)
        sig.append((t.symmetry.moduli, legs, t.block_inds.tobytes(), t.num_codomain, tuple(blocks)))
    return tuple(sig)


def apply_recorded(bb, cache, tag, fn, tensors, fixed=None):
    """``fn()`` -- a function of the block-sparse `tensors` whose device work consists of replayable launches and whose
    result is one tensor -- through the recording cache `cache` (a dict the caller owns).

    The first call with a given (tag, layouts of all tensors) runs ``fn()`` while recording; later calls with the same
    layouts, whatever the buffers and values, replay the launches with every tensor's buffers as relocatable sources and
    never call ``fn``.  `fixed` = (sig, bufs, sizes) of leading tensors whose layout the caller has computed before
    (`tensor_layout`).  ``fn`` must be determined by the STRUCTURE of its inputs (no branch on block values, no
    device-to-host read): compose / permute_legs chains are, decompositions with truncation are not.
    Returns (result, how, recording) with how in 'recorded', 'replayed', 'plain' (recording not replayable)."""
    from .abelian import AbelianTensor
    from .block_backend import HipBlock
    if fixed is None:
        pre_sig, bufs, sizes = (), [], {}
    else:
        pre_sig, bufs, sizes = fixed[0], list(fixed[1]), dict(fixed[2])
    key = (tag, pre_sig, tensor_layout(tensors, bufs, sizes))
    rec = cache.get(key)
    if rec is None:
        rec = Recording(bb, [(p, sizes[p]) for p in bufs])
        out = rec.record(fn)
        if rec.valid:
            template = []
            for blk in out.blocks:
                loc = rec.locate(blk.buf.data_ptr())
                if loc is None or loc[1] != 0:
                    rec.valid = False
                    rec.reason = 'result block outside the recorded allocations'
                    break
                template.append((loc[0], blk.offset, blk.shape, blk.strides))
            rec.result = (out.symmetry, out.legs, template, out.block_inds, out.num_codomain)
        while len(cache) >= MAX_CACHED:
            cache.pop(next(iter(cache)))
        cache[key] = rec
        return out, 'recorded', rec
    if not rec.valid:
        return fn(), 'plain', rec
    tensors_out = rec.replay(bufs)
    sym, legs, template, block_inds, ncod = rec.result
    blocks = [HipBlock(bb, tensors_out[a], off, shp, st) for a, off, shp, st in template]
    return AbelianTensor(sym, legs, blocks, block_inds, ncod), 'replayed', rec
