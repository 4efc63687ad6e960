"""``HipBlockBackend``: host-side mirror of cyten's ``BlockBackend`` operator API for MI355X.

Same method names, argument meaning and error behaviour as the reference interface
(``/root/reference/include/cyten/block_backend/block_backend.h:18-500``; numpy implementation
``src/block_backend/numpy.cpp``), so a caller written against ``bb.matrix_dot / bb.matrix_svd /
bb.permute_axes / ...`` runs unchanged.  Everything that touches block *data* goes through the
C-ABI of ``include/cyten_amd.h`` (hand-written HIP for gfx950); there is no CPU fallback.

On top of the one-block-at-a-time reference API the backend offers the *grouped* entry points the
hardware wants (``matrix_dot_grouped``, ``matrix_svd_batched``, ``matrix_qr_batched``,
``eigh_batched``, ``copy_many`` ...): the tensor backends in :mod:`cyten_amd.abelian` use those so
that one tensor operation is one (or a few) kernel launches; the single-block methods are their
n=1 special case.

Blocks are fp64 on the device.  Like numpy, ``permute_axes``/``reshape``/basic slicing return
*views* (metadata only) whenever the strides allow it.
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import Sequence

import numpy as np

from . import _lib
from .runtime import Context, get_context

__all__ = ['HipBlock', 'Scalar', 'HipBlockBackend', 'GemmPlan', 'DeviceIndex']

# ---- dtype surface (dtypes.h:12-21: bool, int64, float32, complex64, float64, complex128) --------------------------------
# The device holds float64, complex128 and bool storage.  float32 / complex64 / int64 blocks are held in double words and
# carry the nominal type as their dtype: arithmetic runs in double precision and the result is rounded to the nominal type
# when it is stored (compute-in-f64, cast-on-store), so values, promotion rules and `to_numpy()` dtypes are numpy's.
_NOMINAL = (np.dtype('float32'), np.dtype('complex64'), np.dtype('int64'))
_STORAGE_OF = {np.dtype('float32'): np.dtype('float64'), np.dtype('complex64'): np.dtype('complex128'), np.dtype('int64'): np.dtype('float64')}


def _norm_dtype(dtype) -> np.dtype:
    """numpy dtype of a dtype-like (numpy dtype / type / string, or an enum member with a lower-case name like the
    reference's ``Dtype.float32``)"""
    name = getattr(dtype, 'name', None)
    if isinstance(name, str) and not isinstance(dtype, (np.dtype, type)):
        dtype = name.lower()
    d = np.dtype(dtype)
    if d not in _NOMINAL and d not in (np.dtype('float64'), np.dtype('complex128'), np.dtype('bool')):
        raise NotImplementedError(f'HipBlockBackend blocks are bool, int64, float32, complex64, float64 or complex128, not {d}')
    return d


_ZERO_PAD = [(0,) * (_lib.CYB_MAX_NDIM - k) for k in range(_lib.CYB_MAX_NDIM + 1)]


@functools.lru_cache(maxsize=16384)
def _c_strides_cached(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= max(int(s), 1)
    return tuple(reversed(st))


def _c_strides(shape):
    """C-order element strides of `shape` (block shapes repeat: memoised)."""
    return _c_strides_cached(shape if type(shape) is tuple else tuple(int(x) for x in shape))


def _nocopy_reshape_strides(shape, strides, new_shape):
    """Strides of a reshaped *view*, or None if a copy is needed (numpy's no-copy reshape rule)."""
    old = [(d, s) for d, s in zip(shape, strides) if d != 1]
    new_strides = [0] * len(new_shape)
    oi, ni = 0, 0
    on, nn = len(old), len(new_shape)
    while oi < on and ni < nn:
        if new_shape[ni] == 1:
            new_strides[ni] = 0
            ni += 1
            continue
        np_, op = new_shape[ni], old[oi][0]
        oj, nj = oi + 1, ni + 1
        while np_ != op:
            if np_ < op:
                np_ *= new_shape[nj]
                nj += 1
            else:
                op *= old[oj][0]
                oj += 1
        for k in range(oi, oj - 1):  # merged old axes must be mutually contiguous
            if old[k][1] != old[k + 1][0] * old[k + 1][1]:
                return None
        st = old[oj - 1][1]
        for k in range(nj - 1, ni - 1, -1):
            new_strides[k] = st
            st *= new_shape[k]
        oi, ni = oj, nj
    for k in range(ni, nn):
        new_strides[k] = 0 if new_shape[k] == 1 else 1
    return tuple(new_strides)


class HipBlock:
    """A dense float64 or complex128 block in HBM: a strided view (offset, shape, strides in elements)
    of a device buffer.  Counterpart of ``BlockBackend::Block`` (block_backend.h:60-166).  The dtype is
    that of the buffer (a torch float64 or complex128 tensor), so every view inherits it."""

    __slots__ = ('buf', 'offset', 'shape', 'strides', 'backend', '_contig', '_ptr', '_nom')

    def __init__(self, backend, buf, offset, shape, strides):
        self.backend = backend
        self.buf = buf
        self.offset = int(offset)
        self.shape = tuple(map(int, shape))
        self.strides = tuple(map(int, strides))
        self._contig = None   # a view never changes: contiguity and address are computed once, on first use
        self._ptr = None
        self._nom = None      # nominal dtype (float32 / complex64 / int64) of a block held in double words, see `_DtypePolicy`

    @classmethod
    def _trusted(cls, backend, buf, offset, shape, strides, contig=None):
        """Constructor for internal hot paths whose offset / shape / strides already are Python ints in tuples
        (`contig`: the caller knows the answer of ``is_contiguous`` -- a block carved out of a pool is)."""
        self = object.__new__(cls)
        self.backend, self.buf, self.offset, self.shape, self.strides = backend, buf, offset, shape, strides
        self._contig = contig
        self._ptr = None
        self._nom = None
        return self

    # -- metadata (answerable without touching the device)
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def is_complex(self) -> bool:
        return self.buf.is_complex()

    @property
    def is_bool(self) -> bool:
        return self.buf.element_size() == 1

    @property
    def dtype(self):
        nom = getattr(self, '_nom', None)
        if nom is not None:
            return nom
        if self.buf.is_complex():
            return np.dtype('complex128')
        return np.dtype('bool') if self.is_bool else np.dtype('float64')

    @property
    def device(self):
        return self.backend.default_device

    @property
    def ptr(self) -> int:
        p = self._ptr
        if p is None:
            p = self._ptr = self.buf.data_ptr() + self.buf.element_size() * self.offset
        return p

    def is_contiguous(self) -> bool:
        c = self._contig
        if c is None:
            c = True
            if self.size > 1:
                expect = 1
                for d, s in zip(reversed(self.shape), reversed(self.strides)):
                    if d == 1:
                        continue
                    if s != expect:
                        c = False
                        break
                    expect *= d
            self._contig = c
        return c

    def get_backend(self):
        return self.backend

    def to_numpy(self) -> np.ndarray:
        return self.backend.to_numpy(self)

    # -- arithmetic of the Block interface (block_backend.h:91-118)
    def __add__(self, other):
        return self.backend._binary(self, other, 0)

    def __sub__(self, other):
        return self.backend._binary(self, other, 1)

    def __mul__(self, other):
        if isinstance(other, HipBlock):
            return self.backend._binary(self, other, 2)
        return self.backend.mul(other, self)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, HipBlock):
            return self.backend._binary(self, other, 3)
        return self.backend.mul(1.0 / other, self)

    def __abs__(self):
        return self.backend.abs(self)

    # comparisons give boolean blocks (block_backend.h:97-108; numpy.cpp:229-263)
    def __lt__(self, other):
        return self.backend._compare(self, other, 0)

    def __le__(self, other):
        return self.backend._compare(self, other, 1)

    def __gt__(self, other):
        return self.backend._compare(self, other, 2)

    def __ge__(self, other):
        return self.backend._compare(self, other, 3)

    def __eq__(self, other):
        return self.backend._compare(self, other, 4)

    def __ne__(self, other):
        return self.backend._compare(self, other, 5)

    __hash__ = object.__hash__

    def pow(self, exponent):
        """Block::pow (block_backend.h:111-112; numpy.cpp:265-276)."""
        return self.backend._pow(self, exponent)

    __pow__ = pow

    def __getitem__(self, key):
        """Block::get_item (block_backend.h:119-141): an integer multi-index (a tuple or a list of ``ndim`` ints) gives a
        0-d device :class:`Scalar`; slices / index arrays / boolean masks give blocks."""
        if isinstance(key, list) and len(key) == self.ndim and all(isinstance(k, (int, np.integer)) for k in key):
            key = tuple(key)   # the reference reads a list of ndim ints as ONE multi-index (test_block_backend_cpp.py:35-37)
        out = self.backend.get_item(self, key)
        return Scalar(out) if out.ndim == 0 else out

    def __setitem__(self, key, value):
        """Block::set_item (block_backend.h:143-155) with a Block or a Scalar value."""
        if isinstance(key, list) and len(key) == self.ndim and all(isinstance(k, (int, np.integer)) for k in key):
            key = tuple(key)
        self.backend.set_item(self, key, value)

    def save_hdf5(self, hdf5_saver, h5gr, subpath):
        """``Block::save_hdf5`` (block_backend.h:158-161; numpy.cpp:278-283): the payload is the block's array under
        ``subpath + 'arr'``, written through the caller's saver object -- one download, no HDF5 code here."""
        hdf5_saver.save(self.backend.to_numpy(self), subpath + 'arr')

    def __repr__(self):
        return f'HipBlock(shape={self.shape}, strides={self.strides}, device={self.device!r})'


class Scalar:
    """One value with a dtype, held as a 0-d DEVICE block (``BlockBackend::Scalar``, block_backend.h:170-240, implementation
    block_backend.cpp:276-625).  Arithmetic, comparisons and the elementwise functions run on the device through the
    backend exactly as the reference composes them (``operator+`` = ``linear_combination``, ``operator*`` = ``mul``,
    ``sqrt`` = ``backend.sqrt`` ...); the ``as_*`` accessors are the only device-to-host reads.  ``float(s)`` /
    ``complex(s)`` / ``bool(s)`` are host conveniences on top of them."""

    __slots__ = ('_blk',)

    def __init__(self, block):
        if not isinstance(block, HipBlock) or block.ndim != 0:
            raise ValueError('Scalar: block must have ndim() == 0 (trivial empty shape)')   # block_backend.cpp:276-282
        self._blk = block

    # -- metadata / host accessors (block_backend.cpp:284-355)
    @property
    def dtype(self):
        return self._blk.dtype

    @property
    def _block(self):
        return self._blk

    def _item(self):
        b, bb = self._blk, self._blk.backend
        if b.is_bool:
            return bool(bb.ctx.d2h(b.buf, 1, np.uint8, b.offset)[0])
        if b.is_complex:
            return complex(bb.ctx.d2h(b.buf, 1, np.complex128, b.offset)[0])
        return float(bb.ctx.d2h(b.buf, 1, np.float64, b.offset)[0])

    def as_float64(self) -> float:
        if self._blk.is_bool:
            raise RuntimeError('Scalar::as_float64: dtype is Bool')
        if self._blk.is_complex:
            raise RuntimeError('Scalar::as_float64: dtype is complex')
        return self._item()

    def as_complex128(self) -> complex:
        return complex(self._item())

    def as_bool(self) -> bool:
        if not self._blk.is_bool:
            raise RuntimeError('Scalar::as_bool: dtype is not Bool')
        return self._item()

    def as_int64(self) -> int:
        raise RuntimeError('Scalar::as_int64: dtype is not Int64')     # (no int64 blocks on the device)

    def as_float32(self):
        raise RuntimeError('Scalar::as_float32: dtype is not Float32')

    def as_complex64(self):
        raise RuntimeError('Scalar::as_complex64: dtype is not Complex64')

    def to_numpy(self):
        return self.dtype.type(self._item())

    def __float__(self):
        return self.as_float64()

    def __complex__(self):
        return self.as_complex128()

    def __bool__(self):
        v = self._item()
        return bool(v)

    def __repr__(self):
        return f'Scalar({self._item()!r}, dtype={self.dtype})'

    # -- arithmetic (block_backend.cpp:357-408): on the device, through the backend
    def _coerce(self, other):
        if isinstance(other, Scalar):
            return other
        return self._blk.backend.as_scalar(other)

    def __add__(self, other):
        return Scalar(self._blk.backend.linear_combination(1.0, self._blk, 1.0, self._coerce(other)._blk))

    __radd__ = __add__

    def __neg__(self):
        return Scalar(self._blk.backend.mul(-1.0, self._blk))

    def __sub__(self, other):
        return Scalar(self._blk.backend.linear_combination(1.0, self._blk, -1.0, self._coerce(other)._blk))

    def __rsub__(self, other):
        return self._coerce(other) - self

    def __mul__(self, other):
        return Scalar(self._blk.backend.multiply_blocks(self._blk, self._coerce(other)._blk))

    __rmul__ = __mul__

    def inverse(self):
        z = self.as_complex128()
        if z == 0:
            raise RuntimeError('Division by zero')                       # block_backend.cpp:394-404
        bb = self._blk.backend
        return bb.as_scalar(1.0 / z if self._blk.is_complex else 1.0 / z.real)

    def __truediv__(self, other):
        return self * self._coerce(other).inverse()

    def __rtruediv__(self, other):
        return self._coerce(other) * self.inverse()

    def _cmp(self, other, op):
        o = self._coerce(other)
        if self._blk.is_complex or o._blk.is_complex or self._blk.is_bool or o._blk.is_bool:
            if op not in (4, 5):
                raise TypeError('ordering comparisons are defined for real scalars only')
            same = self._item() == o._item()          # complex / bool (in)equality: two host reads, one bool scalar back
            return self._blk.backend.as_scalar(same if op == 4 else not same)
        return Scalar(self._blk.backend._compare(self._blk, o._blk, op))

    def __lt__(self, other):
        return self._cmp(other, 0)

    def __le__(self, other):
        return self._cmp(other, 1)

    def __gt__(self, other):
        return self._cmp(other, 2)

    def __ge__(self, other):
        return self._cmp(other, 3)

    def __eq__(self, other):
        return self._cmp(other, 4)

    def __ne__(self, other):
        return self._cmp(other, 5)

    __hash__ = object.__hash__

    # -- convenience access, delegating to the block backend (block_backend.cpp:585-625)
    def real(self):
        return Scalar(self._blk.backend.real(self._blk))

    def imag(self):
        return Scalar(self._blk.backend.imag(self._blk))

    def abs(self):
        return Scalar(self._blk.backend.abs(self._blk))

    __abs__ = abs

    def sqrt(self):
        return Scalar(self._blk.backend.sqrt(self._blk))

    def exp(self):
        return Scalar(self._blk.backend.exp(self._blk))

    def log(self):
        return Scalar(self._blk.backend.log(self._blk))

    def pow(self, exponent):
        e = exponent._blk if isinstance(exponent, Scalar) else exponent
        return Scalar(self._blk.backend._pow(self._blk, e))

    __pow__ = pow


class DeviceIndex:
    """`n` ascending int64 positions in device memory at `ptr` (one sector's kept singular values as written by
    ``truncate_select``); accepted by ``mask_gather_many`` in place of a host mask.  `owner` keeps the table alive."""

    __slots__ = ('ptr', 'n', 'owner')

    def __init__(self, ptr, n, owner):
        self.ptr, self.n, self.owner = int(ptr), int(n), owner


class GemmPlan:
    """Device-resident launch plan of one grouped block GEMM (tile queue + descriptors)."""

    def __init__(self, backend, handle, outs, keepalive):
        self.backend = backend
        self.handle = handle
        self.outs = outs
        self._keepalive = keepalive
        flops, nbytes = C.c_double(), C.c_double()
        ntiles, nlaunch = C.c_int64(), C.c_int32()
        _lib.check(backend.lib.cyb_gemm_plan_info(handle, C.byref(flops), C.byref(nbytes), C.byref(ntiles), C.byref(nlaunch)))
        self.flops, self.bytes, self.n_tiles, self.n_launches = flops.value, nbytes.value, ntiles.value, nlaunch.value

    def run(self):
        self.backend.ctx.sync_stream()
        _lib.check(self.backend.lib.cyb_gemm_plan_run(self.backend.ctx.handle, self.handle))
        return self.outs

    def destroy(self):
        if self.handle is not None:
            _lib.check(self.backend.lib.cyb_gemm_plan_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class HipBlockBackend:
    """MI355X block backend (see module docstring)."""

    svd_algorithms = ['jacobi', 'gesdd', 'gesvd', 'robust', 'robust_silent']

    def __init__(self, default_device: str = 'cuda:0'):
        self.default_device = self.as_device(default_device)
        self.ctx: Context = get_context(int(self.default_device.split(':')[1]))
        self.lib = self.ctx.lib

    # ------------------------------------------------------------------ identity / devices
    def get_backend_name(self) -> str:
        return 'HipBlockBackend'

    def __repr__(self):
        return f'HipBlockBackend({self.default_device!r})'

    def __eq__(self, other):
        return isinstance(other, HipBlockBackend) and other.default_device == self.default_device

    def __hash__(self):
        return hash(('HipBlockBackend', self.default_device))

    def as_device(self, device) -> str:
        """Canonical device string (torch.cpp:85-116 normalises to ``type:index``)."""
        if device is None:
            return getattr(self, 'default_device', 'cuda:0')
        device = str(device)
        if device in ('cuda', 'gpu', 'hip'):
            return 'cuda:0'
        if device.startswith(('cuda:', 'hip:', 'gpu:')):
            return 'cuda:' + str(int(device.split(':')[1]))
        raise ValueError(f'HipBlockBackend: unsupported device {device!r}')

    def possible_svd_algorithms(self):
        return list(self.svd_algorithms)

    def synchronize(self):
        self.ctx.synchronize()

    def is_correct_block_type(self, block) -> bool:
        return isinstance(block, HipBlock)

    def test_block_sanity(self, block, expect_shape=None, expect_dtype=None, expect_device=None):
        if not isinstance(block, HipBlock):
            raise RuntimeError('wrong block type')
        if expect_shape is not None and tuple(expect_shape) != block.shape:
            raise RuntimeError(f'wrong block shape {block.shape} != {tuple(expect_shape)}')
        if expect_dtype is not None and np.dtype(expect_dtype) != block.dtype:
            raise RuntimeError('wrong block dtype')
        if expect_device is not None and self.as_device(expect_device) != block.device:
            raise RuntimeError('wrong block device')

    # ------------------------------------------------------------------ nominal dtypes (float32 / complex64 / int64)
    _n_tagged = 0        # blocks that ever got a nominal dtype on this backend: while 0 the dtype policy costs one attribute read
    _policy_depth = 0

    def _retag(self, blk: HipBlock, nominal, round_values: bool) -> HipBlock:
        """The block `blk` (same buffer) with the nominal dtype `nominal` (None: its storage dtype); `round_values`: the
        values are first rounded to the nominal type IN PLACE (callers pass freshly computed results only)."""
        if nominal is not None and round_values and blk.size:
            target = blk if blk.is_contiguous() else self.contiguous(blk)
            flat = self._fview(target) if target.is_complex else target
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_unary_batched_f64(self.ctx.handle, self._vec_descs([flat], None, [flat]), 1,
                                                      8 if nominal == np.dtype('int64') else 7))
            if target is not blk:
                self.copy_many([(blk, target)])
        out = HipBlock._trusted(self, blk.buf, blk.offset, blk.shape, blk.strides, blk._contig)
        out._nom = nominal
        if nominal is not None:
            self._n_tagged += 1
        return out

    # ------------------------------------------------------------------ creation / transfer
    def _new(self, shape, cplx: bool = False) -> HipBlock:
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        return HipBlock(self, self.ctx.empty(n, 'complex128' if cplx else 'float64'), 0, shape, _c_strides(shape))

    def _new_bool(self, shape) -> HipBlock:
        shape = tuple(int(s) for s in shape)
        return HipBlock(self, self.ctx.empty(math.prod(shape), 'bool'), 0, shape, _c_strides(shape))

    def _new_like(self, a: HipBlock, shape=None) -> HipBlock:
        shape = a.shape if shape is None else shape
        return self._new_bool(shape) if a.is_bool else self._new(shape, a.is_complex)

    def _new_many(self, shapes, cplx: bool = False, zero: bool = False):
        """Blocks of the given shapes carved out of ONE device buffer (256-byte aligned offsets): one allocation and,
        with `zero`, one memset for the block list of a tensor operation instead of one per block."""
        shapes = [tuple(int(x) for x in sh) for sh in shapes]
        offs, tot = [], 0
        for sh in shapes:
            n = 1
            for x in sh:
                n *= x
            offs.append(tot)
            tot += (n + 31) // 32 * 32
        buf = self.ctx.empty(tot, 'complex128' if cplx else 'float64')
        if zero and tot:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_memset(self.ctx.handle, C.c_void_p(buf.data_ptr()), 0, buf.element_size() * tot))
        return [HipBlock._trusted(self, buf, o, sh, _c_strides(sh), True) for o, sh in zip(offs, shapes)]

    def zeros_many(self, shapes, dtype=None, device=None):
        """``zeros`` for a list of shapes (the result blocks of ``AbelianBackend::combine_legs``,
        abelian.cpp:1200-1214): one buffer, one memset."""
        return self._new_many(shapes, dtype is not None and np.dtype(dtype).kind == 'c', zero=True)

    # -- complex128 helpers (interleaved storage: a complex block IS a float64 block with a trailing axis of 2)
    def _fview(self, a: HipBlock) -> HipBlock:
        """float64 alias of a complex view: shape + (2,), metadata only."""
        fbuf = self.ctx.torch.view_as_real(a.buf).reshape(-1)
        return HipBlock(self, fbuf, 2 * a.offset, a.shape + (2,), tuple(2 * s for s in a.strides) + (1,))

    def _plane(self, a: HipBlock, which: int) -> HipBlock:
        """real (0) or imaginary (1) part of a complex view as a strided float64 view (no copy)."""
        fbuf = self.ctx.torch.view_as_real(a.buf).reshape(-1)
        return HipBlock(self, fbuf, 2 * a.offset + which, a.shape, tuple(2 * s for s in a.strides))

    def as_complex(self, a: HipBlock) -> HipBlock:
        """complex128 copy of a float64 block (imaginary part zero)."""
        if a.is_complex:
            return a
        out = self.zeros(a.shape, dtype='complex128')
        self.copy_many([(self._plane(out, 0), a)])
        return out

    def empty_block(self, shape) -> HipBlock:
        return self._new(shape)

    def _check_device(self, device):
        """a backend instance serves ONE device (torch.cpp:669-695 keeps one singleton per device the same way)"""
        if device is not None and self.as_device(device) != self.default_device:
            raise ValueError(f'{self!r} holds its blocks on {self.default_device}, not on {self.as_device(device)}: use the backend of that device')

    def as_block(self, a, dtype=None, device=None) -> HipBlock:
        self._check_device(device)
        if isinstance(a, HipBlock):
            return a if dtype is None or np.dtype(dtype) == a.dtype else self.to_dtype(a, dtype)
        return self.block_from_numpy(np.asarray(a), dtype, device)

    def block_from_numpy(self, a: np.ndarray, dtype=None, device=None) -> HipBlock:
        self._check_device(device)
        a = np.asarray(a)
        shape = a.shape   # (np.ascontiguousarray promotes 0-d to 1-d: a Scalar's block keeps its empty shape)
        if (a.dtype == np.bool_ and dtype is None) or (dtype is not None and np.dtype(dtype).kind == 'b'):
            a = np.ascontiguousarray(a, dtype=np.bool_)
            blk = self._new_bool(shape)
            self.ctx.h2d(blk.buf, a.view(np.uint8))
            return blk
        want = _norm_dtype(dtype) if dtype is not None else (a.dtype if a.dtype in _NOMINAL else None)
        if want is not None and want in _NOMINAL:      # held in double words, values already representable in the nominal type
            a = a.astype(want)
            blk = self.block_from_numpy(a.astype(_STORAGE_OF[want]))
            return self._retag(blk, want, False)
        cplx = np.iscomplexobj(a) or (dtype is not None and np.dtype(dtype).kind == 'c')
        a = np.ascontiguousarray(a, dtype=np.complex128 if cplx else np.float64)
        blk = self._new(shape, cplx)
        self.ctx.h2d(blk.buf, a)
        return blk

    def to_numpy(self, a: HipBlock, numpy_dtype=None) -> np.ndarray:
        c = self.contiguous(a)
        if c.is_bool:
            out = self.ctx.d2h(c.buf, c.size, np.uint8, c.offset).reshape(c.shape).astype(np.bool_)
        else:
            out = self.ctx.d2h(c.buf, c.size, np.complex128 if c.is_complex else np.float64, c.offset).reshape(c.shape)
        nom = getattr(a, '_nom', None)
        if nom is not None and numpy_dtype is None:
            return out.astype(nom)
        return out if numpy_dtype is None else out.astype(numpy_dtype)

    def concatenate_to_numpy(self, blocks) -> np.ndarray:
        """The flattened blocks of a list as ONE host array: one batched gather into a staging buffer and one download
        (the singular values of all sectors for the host-side truncation, abelian.cpp:3631)."""
        blocks = list(blocks)
        if not blocks:
            return np.zeros(0)
        if any(b.is_complex != blocks[0].is_complex or b.is_bool for b in blocks):
            return np.concatenate([self.to_numpy(b).reshape(-1) for b in blocks])
        n = sum(b.size for b in blocks)
        stage = self._new((n,), blocks[0].is_complex)
        pairs, off = [], 0
        for b in blocks:
            pairs.append((HipBlock(self, stage.buf, stage.offset + off, b.shape, _c_strides(b.shape)), b))
            off += b.size
        self.copy_many(pairs)
        return self.to_numpy(stage)

    def zeros(self, shape, dtype=None, device=None) -> HipBlock:
        kind = 'f' if dtype is None else _norm_dtype(dtype).kind
        blk = self._new_bool(shape) if kind == 'b' else self._new(shape, kind == 'c')
        if blk.size:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_memset(self.ctx.handle, C.c_void_p(blk.ptr), 0, blk.buf.element_size() * blk.size))
        return blk

    def ones_block(self, shape, dtype=None, device=None) -> HipBlock:
        """numpy.cpp:853-866 (np.ones(shape, dtype))"""
        kind = 'f' if dtype is None else _norm_dtype(dtype).kind
        blk = self._new(shape)
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_fill_f64(self.ctx.handle, C.c_void_p(blk.ptr), blk.size, 1.0))
        return blk if kind == 'f' else self.to_dtype(blk, 'complex128' if kind == 'c' else 'bool')

    def eye_matrix(self, dim, dtype=None, device=None) -> HipBlock:
        """numpy.cpp:1197-1207 (np.eye(dim, dtype=dtype))"""
        kind = 'f' if dtype is None else _norm_dtype(dtype).kind
        blk = self._new((dim, dim))
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_eye_f64(self.ctx.handle, C.c_void_p(blk.ptr), int(dim)))
        return blk if kind == 'f' else self.to_dtype(blk, 'complex128' if kind == 'c' else 'bool')

    def eye_block(self, legs, dtype=None, device=None) -> HipBlock:
        """block_backend.cpp:1013-1031: identity on prod(legs), reshaped to legs + legs."""
        legs = [int(d) for d in legs]
        n = int(np.prod(legs)) if legs else 1
        return self.reshape(self.eye_matrix(n, dtype), legs + legs)

    def random_normal(self, dims, dtype=None, sigma=1.0, device=None, seed=None) -> HipBlock:
        """numpy.cpp:934-963: standard deviation `sigma`; a complex dtype splits it over real and imaginary part
        (sigma / sqrt(2) each)."""
        cplx = dtype is not None and _norm_dtype(dtype).kind == 'c'
        blk = self._new(dims, cplx)
        if seed is None:
            seed = int(np.random.default_rng().integers(0, 2 ** 63 - 1))
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_random_normal_f64(self.ctx.handle, C.c_void_p(blk.ptr), blk.size * (2 if cplx else 1), int(seed),
                                                  float(sigma) / (math.sqrt(2.0) if cplx else 1.0)))
        return blk

    def copy_block(self, a: HipBlock, device=None) -> HipBlock:
        out = self._new_like(a)
        self.copy_many([(out, a)])
        return out

    # ------------------------------------------------------------------ metadata-only ops
    def get_shape(self, a):
        return list(a.shape)

    def get_dtype(self, a):
        return a.dtype

    def get_device(self, a):
        return a.device

    def is_real(self, a):
        return not a.is_complex

    def permute_axes(self, a: HipBlock, permutation: Sequence[int]) -> HipBlock:
        """A view, like numpy's transpose (numpy.cpp:924-931)."""
        permutation = [int(p) for p in permutation]
        if sorted(permutation) != list(range(a.ndim)):
            raise ValueError(f'invalid permutation {permutation} for {a.ndim} axes')
        return HipBlock(self, a.buf, a.offset, [a.shape[p] for p in permutation], [a.strides[p] for p in permutation])

    def reshape(self, a: HipBlock, shape: Sequence[int]) -> HipBlock:
        """numpy reshape semantics incl. one ``-1`` (numpy.cpp:1057-1064): a view if the strides
        allow it, else a contiguous copy."""
        shape = [int(s) for s in shape]
        if shape.count(-1) > 1:
            raise ValueError('can only specify one unknown dimension')
        if -1 in shape:
            known = 1
            for s in shape:
                if s != -1:
                    known *= s
            if known == 0 or a.size % known:
                raise ValueError(f'cannot reshape block of size {a.size} into shape {tuple(shape)}')
            shape[shape.index(-1)] = a.size // known
        n = 1
        for s in shape:
            n *= s
        if n != a.size:
            raise ValueError(f'cannot reshape block of size {a.size} into shape {tuple(shape)}')
        if a.size == 0:
            return HipBlock(self, a.buf, a.offset, shape, _c_strides(shape))
        st = _c_strides(shape) if a.is_contiguous() else _nocopy_reshape_strides(a.shape, a.strides, shape)
        if st is None:
            c = self.contiguous(a)
            return HipBlock(self, c.buf, c.offset, shape, _c_strides(shape))
        return HipBlock(self, a.buf, a.offset, shape, st)

    def add_axis(self, a: HipBlock, pos: int) -> HipBlock:
        shape, strides = list(a.shape), list(a.strides)
        shape.insert(pos, 1)
        strides.insert(pos, 0)
        return HipBlock(self, a.buf, a.offset, shape, strides)

    def squeeze_axes(self, a: HipBlock, idcs) -> HipBlock:
        idcs = [i % a.ndim for i in idcs]
        for i in idcs:
            if a.shape[i] != 1:
                raise ValueError('cannot squeeze an axis of extent != 1')
        keep = [k for k in range(a.ndim) if k not in idcs]
        return HipBlock(self, a.buf, a.offset, [a.shape[k] for k in keep], [a.strides[k] for k in keep])

    def get_item(self, a: HipBlock, key) -> HipBlock:
        """Basic indexing (ints and slices) as a view; index arrays gather.  A slice with a negative step is served as a
        gather along its axis (a copy where numpy gives a view: blocks are values to the tensor backends, which never write
        through an alias of an input)."""
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) > a.ndim:
            raise IndexError('too many indices for block')
        key = key + (slice(None),) * (a.ndim - len(key))
        offset, shape, strides = a.offset, [], []
        gather = None
        reversed_axes = []
        for ax, k in enumerate(key):
            d, s = a.shape[ax], a.strides[ax]
            if isinstance(k, (int, np.integer)):
                k = int(k)
                if k < 0:
                    k += d
                if not 0 <= k < d:
                    raise IndexError(f'index {k} out of bounds for axis {ax} with size {d}')
                offset += k * s
            elif isinstance(k, slice):
                start, stop, step = k.indices(d)
                if step < 0:
                    reversed_axes.append((len(shape), np.arange(start, stop, step, dtype=np.int64)))
                    shape.append(d)
                    strides.append(s)
                    continue
                n = max(0, (stop - start + step - 1) // step)
                offset += start * s
                shape.append(n)
                strides.append(s * step)
            else:
                if gather is not None:
                    raise NotImplementedError('more than one index array')
                gather = (len(shape), np.asarray(k))
                shape.append(d)
                strides.append(s)
        view = HipBlock(self, a.buf, offset, shape, strides)
        if gather is not None:
            ax, idx = gather
            if idx.dtype == bool:
                view = self.apply_mask(view, idx, ax)
            else:
                view = self._gather_axis(view, np.asarray(idx, dtype=np.int64), ax)
        for ax, idx in reversed_axes:
            view = self._gather_axis(view, idx, ax)
        return view

    def subblock(self, a: HipBlock, r0: int, r1: int, c0: int, c1: int) -> HipBlock:
        """``a[r0:r1, c0:c1]`` of a 2-D block as a view, without the generality (and the cost) of ``get_item``: the
        placement of hundreds of sector sub-blocks per combine / split."""
        s0, s1 = a.strides
        return HipBlock._trusted(self, a.buf, a.offset + int(r0) * s0 + int(c0) * s1, (int(r1 - r0), int(c1 - c0)), (s0, s1))

    def set_item(self, a: HipBlock, key, value):
        """``a[key] = value`` (numpy.cpp:145-190; abelian.cpp:1212-1214): basic keys (ints, forward slices) are ONE strided
        copy into the view; keys that ``get_item`` serves as gathered copies -- an index array, a boolean mask, a slice with
        a negative step -- are written through as one batched copy with a (target slab, source slab) pair per selected
        index, so that ``a`` changes exactly as numpy's ``a[key] = value`` changes it.  ``value``: Block, Scalar or number."""
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) > a.ndim:
            raise IndexError('too many indices for block')
        key = key + (slice(None),) * (a.ndim - len(key))
        if isinstance(value, Scalar):
            value = value._blk
        # normalise: at most one axis is addressed by an explicit index list
        basic, special, out_ax = [], None, 0
        for ax, k in enumerate(key):
            if isinstance(k, (int, np.integer)):
                basic.append(int(k))
                continue
            if isinstance(k, slice):
                start, stop, step = k.indices(a.shape[ax])
                if step > 0:
                    basic.append(k)
                    out_ax += 1
                    continue
                k = np.arange(start, stop, step, dtype=np.int64)
            k = np.asarray(k)
            if k.dtype == bool:
                if k.shape != (a.shape[ax],):
                    raise IndexError('boolean index does not match the axis')
                k = np.flatnonzero(k)
            if special is not None:
                raise NotImplementedError('set_item: more than one index array / reversed slice')
            k = k.astype(np.int64)
            k = np.where(k < 0, k + a.shape[ax], k)
            if k.ndim != 1 or (len(k) and (k.min() < 0 or k.max() >= a.shape[ax])):
                raise IndexError('index array out of bounds')
            special = (ax, out_ax, k)
            basic.append(slice(None))
            out_ax += 1
        target = self.get_item(a, tuple(basic))
        if special is not None:
            ax, oax, idx = special
            tshape = target.shape[:oax] + (len(idx),) + target.shape[oax + 1:]
        else:
            tshape = target.shape
        if not isinstance(value, HipBlock):
            value = self.block_from_numpy(np.broadcast_to(np.asarray(value, complex if a.is_complex else float), tshape))
        if value.shape != tshape:
            if value.size == 1 or value.ndim <= len(tshape):     # numpy broadcasting of the value (a Scalar, a row, ...)
                value = self.block_from_numpy(np.broadcast_to(self.to_numpy(value), tshape).copy())
            else:
                raise ValueError(f'shape mismatch in set_item: {value.shape} vs {tshape}')
        if value.is_complex != a.is_complex:
            value = self.as_complex(value) if a.is_complex else value
        if special is None:
            self.copy_many([(target, value)])
            return
        # one (destination slab, source slab) pair per selected index; a repeated index keeps its LAST source (numpy)
        last = {int(i): j for j, i in enumerate(idx.tolist())}
        sel_t = [slice(None)] * target.ndim
        pairs = []
        for i, j in last.items():
            sel_t[oax] = i
            sel_v = list(sel_t)
            sel_v[oax] = j
            pairs.append((self.get_item(target, tuple(sel_t)), self.get_item(value, tuple(sel_v))))
        self.copy_many(pairs)

    # ------------------------------------------------------------------ data movement
    def copy_many(self, pairs, conj: bool = False):
        """``dst[...] = src`` (or its complex conjugate) for a list of (dst_view, src_view) pairs of equal
        shapes and dtypes: ONE launch per element size."""
        pairs = [(d, s) for d, s in pairs if d.size]
        if not pairs:
            return
        for esz in (8, 16, 1):
            sel = [(d, s) for d, s in pairs if d.buf.element_size() == esz]
            if not sel:
                continue
            arr = np.zeros(len(sel), dtype=_lib.COPY_DTYPE)
            shp, dst, sst = [], [], []
            for d, s in sel:
                if d.shape != s.shape:
                    raise ValueError(f'copy_many: shape mismatch {d.shape} vs {s.shape}')
                if s.buf.dtype != d.buf.dtype:
                    raise ValueError('copy_many: dtype mismatch (use as_complex / real / imag / to_dtype)')
                if d.ndim > _lib.CYB_MAX_NDIM:
                    raise NotImplementedError(f'blocks with more than {_lib.CYB_MAX_NDIM} axes')
                pad = _ZERO_PAD[d.ndim]
                shp.append(d.shape + pad)
                dst.append(d.strides + pad)
                sst.append(s.strides + pad)
            arr['dst'] = [d.ptr for d, _ in sel]
            arr['src'] = [s.ptr for _, s in sel]
            arr['ndim'] = [d.ndim for d, _ in sel]
            arr['conj'] = 1 if (conj and esz == 16) else 0
            arr['shape'], arr['dst_strides'], arr['src_strides'] = shp, dst, sst
            descs = arr.ctypes.data_as(C.POINTER(_lib.CopyDesc))
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_copy_strided_batched(self.ctx.handle, descs, len(sel), esz))

    def copy_2d_many(self, dst_ptr, dst_ld, src_ptr, src_ld, rows, cols, elem_size: int = 8):
        """``dst[r, c] = src[r, c]`` for a list of 2-D row-major sub-blocks given as plain arrays (base addresses, leading
        dimensions, extents): the descriptor array is filled with numpy, no per-block host objects -- the scatter of
        ``combine_legs`` over a 728-block list is one call of this."""
        n = len(dst_ptr)
        if n == 0:
            return
        arr = np.zeros(n, dtype=_lib.COPY_DTYPE)
        arr['dst'], arr['src'] = dst_ptr, src_ptr
        arr['ndim'] = 2
        arr['shape'][:, 0], arr['shape'][:, 1] = rows, cols
        arr['dst_strides'][:, 0], arr['dst_strides'][:, 1] = dst_ld, 1
        arr['src_strides'][:, 0], arr['src_strides'][:, 1] = src_ld, 1
        keep = (np.asarray(rows) > 0) & (np.asarray(cols) > 0)
        if not keep.all():
            arr = np.ascontiguousarray(arr[keep])
            if len(arr) == 0:
                return
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_copy_strided_batched(self.ctx.handle, arr.ctypes.data_as(C.POINTER(_lib.CopyDesc)), len(arr), elem_size))

    def contiguous(self, a: HipBlock) -> HipBlock:
        if a.is_contiguous():
            return a
        out = self._new_like(a)
        self.copy_many([(out, a)])
        return out

    def contiguous_many(self, blocks):
        outs = list(blocks)
        todo = [i for i, a in enumerate(blocks) if not a.is_contiguous()]
        if todo:
            pairs = []
            for kind in ('f', 'c', 'b'):  # one buffer per dtype for all the copies of this call
                sel = [i for i in todo if ('b' if blocks[i].is_bool else 'c' if blocks[i].is_complex else 'f') == kind]
                if not sel:
                    continue
                new = [self._new_bool(blocks[i].shape) for i in sel] if kind == 'b' else \
                    self._new_many([blocks[i].shape for i in sel], kind == 'c')
                for i, o in zip(sel, new):
                    outs[i] = o
                    pairs.append((o, blocks[i]))
            self.copy_many(pairs)
        return outs

    def combine_legs(self, a: HipBlock, leg_idcs_combine, cstyles=True) -> HipBlock:
        """block_backend.cpp:784-829: permute each group (reversed if not C-style) then reshape."""
        if isinstance(cstyles, bool):
            cstyles = [cstyles] * len(leg_idcs_combine)
        perm, shape, k = [], [], 0
        groups = {g[0]: (g, c) for g, c in zip(leg_idcs_combine, cstyles)}
        in_group = {i for g in leg_idcs_combine for i in g}
        while k < a.ndim:
            if k in groups:
                g, c = groups[k]
                g = list(g) if c else list(reversed(g))
                perm += g
                shape.append(math.prod(a.shape[i] for i in g))
                k = max(g) + 1
            elif k in in_group:
                k += 1
            else:
                perm.append(k)
                shape.append(a.shape[k])
                k += 1
        return self.reshape(self.permute_axes(a, perm), shape)

    def split_legs(self, a: HipBlock, idcs, dims, cstyles=True) -> HipBlock:
        """block_backend.cpp:924-982: inverse of combine_legs."""
        if isinstance(cstyles, bool):
            cstyles = [cstyles] * len(idcs)
        shape, perm = [], []
        split = {i: (list(d), c) for i, d, c in zip(idcs, dims, cstyles)}
        for k in range(a.ndim):
            if k in split:
                d, c = split[k]
                base = len(shape)
                if c:
                    shape += d
                    perm += list(range(base, base + len(d)))
                else:
                    shape += list(reversed(d))
                    perm += list(range(base + len(d) - 1, base - 1, -1))
            else:
                perm.append(len(shape))
                shape.append(a.shape[k])
        return self.permute_axes(self.reshape(a, shape), perm)

    def dagger(self, a: HipBlock) -> HipBlock:
        """block_backend.cpp:840-848: reversed axes, complex conjugate."""
        return self.conj(self.permute_axes(a, list(range(a.ndim - 1, -1, -1))))

    def conj(self, a):
        """numpy.cpp:566-573.  Real blocks are their own conjugate (a view); complex ones are copied with the
        sign of the imaginary part flipped (one launch)."""
        if not a.is_complex:
            return a
        out = self._new(a.shape, True)
        self.copy_many([(out, a)], conj=True)
        return out

    def real(self, a):
        return self._plane(a, 0) if a.is_complex else a

    def imag(self, a):
        return self._plane(a, 1) if a.is_complex else self.zeros(a.shape)

    def _as_3d(self, a: HipBlock, axis: int):
        axis = axis % a.ndim
        outer = math.prod(map(int, a.shape[:axis])) if axis else 1
        inner = math.prod(map(int, a.shape[axis + 1:])) if axis + 1 < a.ndim else 1
        return outer, a.shape[axis], inner

    def _gather_axis(self, a: HipBlock, idx: np.ndarray, axis: int) -> HipBlock:
        return self.mask_gather_many([(a, idx, axis)])[0]

    def mask_gather_many(self, items, outs=None):
        """apply_mask for a list of (block, keep_indices_or_boolmask, axis): ONE launch, one upload of all
        kept-index tables, outputs carved out of one buffer per dtype -- or written into the caller's contiguous
        blocks ``outs`` (sharded runs gather straight into their segment of the pool that is all-gathered)."""
        if not items:
            return []
        user_outs = outs
        if user_outs is not None and (len(user_outs) != len(items) or any(it[0].is_bool for it in items)):
            raise ValueError('mask_gather_many: outs must match the items one to one (numeric blocks only)')
        if any(it[0].is_bool for it in items):  # the gather kernel moves 8-byte words: boolean blocks make the round trip
            was_bool = [it[0].is_bool for it in items]
            outs = self.mask_gather_many([(self.to_dtype(a, 'float64') if a.is_bool else a, m, ax) for a, m, ax in items])
            return [self.to_dtype(o, 'bool') if b else o for o, b in zip(outs, was_bool)]
        descs = (_lib.MaskDesc * len(items))()
        srcs = self.contiguous_many([it[0] for it in items])
        idxs, geo = [], []
        for (_, mask, axis), a in zip(items, srcs):
            axis = axis % a.ndim
            if isinstance(mask, DeviceIndex):   # kept positions already on the device (truncate_select)
                idxs.append(mask)
                geo.append((axis, a.shape[:axis] + (mask.n,) + a.shape[axis + 1:]))
                continue
            mask = np.asarray(mask)
            idx = np.flatnonzero(mask) if mask.dtype == bool else mask.astype(np.int64)
            if mask.dtype == bool and mask.shape[0] != a.shape[axis]:
                raise ValueError('mask length does not match the axis')
            idxs.append(idx.astype(np.int64, copy=False))
            geo.append((axis, a.shape[:axis] + (len(idx),) + a.shape[axis + 1:]))
        outs = [None] * len(items)
        if user_outs is not None:
            for i, (o, a) in enumerate(zip(user_outs, srcs)):
                if tuple(o.shape) != tuple(geo[i][1]) or not o.is_contiguous() or o.is_complex != a.is_complex:
                    raise ValueError(f'mask_gather_many: outs[{i}] must be a contiguous block of shape {geo[i][1]}')
                outs[i] = o
        else:
            for cplx in (False, True):
                sel = [i for i, a in enumerate(srcs) if a.is_complex == cplx]
                for i, o in zip(sel, self._new_many([geo[i][1] for i in sel], cplx)):
                    outs[i] = o
        host = [x for x in idxs if not isinstance(x, DeviceIndex)]
        offs = np.concatenate([[0], np.cumsum([len(x) for x in host])]).astype(np.int64)
        didx = self.ctx.empty(int(offs[-1]), 'int64')
        if host:
            self.ctx.h2d(didx, np.concatenate(host))
        h = 0
        for i, (a, out) in enumerate(zip(srcs, outs)):
            outer, ax, inner = self._as_3d(a, geo[i][0])
            if a.is_complex:  # interleaved storage: the (re, im) pair is one more inner axis
                inner *= 2
            if isinstance(idxs[i], DeviceIndex):
                table, n_keep = idxs[i].ptr, idxs[i].n
            else:
                table, n_keep = didx.data_ptr() + 8 * int(offs[h]), len(idxs[i])
                h += 1
            descs[i].x, descs[i].out, descs[i].idx = a.ptr, out.ptr, table
            descs[i].outer, descs[i].axis, descs[i].inner, descs[i].n_keep = outer, ax, inner, n_keep
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_mask_gather_batched_f64(self.ctx.handle, descs, len(items)))
        return outs

    def apply_mask(self, block: HipBlock, mask, ax: int) -> HipBlock:
        """numpy.cpp:605-613: keep the entries of axis `ax` where the 1-D boolean mask is True."""
        mask = np.asarray(mask.to_numpy() if isinstance(mask, HipBlock) else mask).astype(bool)
        return self.mask_gather_many([(block, mask, ax)])[0]

    def enlarge_leg_many(self, items):
        """``enlarge_leg`` (numpy.cpp:700-728) for a list of (block, mask, axis): ONE zero-filled allocation per dtype, one
        upload of all position tables and ONE scatter launch -- the per-block loop of ``AbelianBackend::_mask_contract``
        with ``large_leg=False`` (abelian.cpp:2550-2553).  `mask`: 1-D bool array / block, or the positions themselves
        together with the large extent as ``(positions, n_large)``."""
        if not items:
            return []
        srcs = self.contiguous_many([it[0] for it in items])
        self._numeric_only(srcs, 'enlarge_leg')
        idxs, geo = [], []
        for (_, mask, axis), a in zip(items, srcs):
            axis = axis % a.ndim
            if isinstance(mask, tuple):
                idx, n_large = np.asarray(mask[0], dtype=np.int64), int(mask[1])
            else:
                m = np.asarray(mask.to_numpy() if isinstance(mask, HipBlock) else mask).astype(bool)
                idx, n_large = np.flatnonzero(m).astype(np.int64), len(m)
            if len(idx) != a.shape[axis]:
                raise ValueError('mask does not match the axis to enlarge')
            idxs.append(idx)
            geo.append((axis, n_large, a.shape[:axis] + (n_large,) + a.shape[axis + 1:]))
        outs = [None] * len(items)
        for cplx in (False, True):
            sel = [i for i, a in enumerate(srcs) if a.is_complex == cplx]
            for i, o in zip(sel, self._new_many([geo[i][2] for i in sel], cplx, zero=True)):
                outs[i] = o
        offs = np.concatenate([[0], np.cumsum([len(x) for x in idxs])]).astype(np.int64)
        didx = self.ctx.empty(max(int(offs[-1]), 1), 'int64')
        if offs[-1]:
            self.ctx.h2d(didx, np.concatenate(idxs))
        descs = (_lib.MaskDesc * len(items))()
        for i, (a, out) in enumerate(zip(srcs, outs)):
            outer, _, inner = self._as_3d(a, geo[i][0])
            if a.is_complex:
                inner *= 2
            descs[i].x, descs[i].out, descs[i].idx = a.ptr, out.ptr, didx.data_ptr() + 8 * int(offs[i])
            descs[i].outer, descs[i].axis, descs[i].inner, descs[i].n_keep = outer, geo[i][1], inner, len(idxs[i])
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_mask_scatter_batched_f64(self.ctx.handle, descs, len(items)))
        return outs

    def enlarge_leg(self, block: HipBlock, mask, axis: int) -> HipBlock:
        """numpy.cpp:700-728: scatter into zeros along `axis` at the True positions of mask."""
        return self.enlarge_leg_many([(block, mask, axis)])[0]

    # ------------------------------------------------------------------ BLAS-1 class ops
    @staticmethod
    def _numeric_only(blocks, what):
        """The float64 / complex128 kernels address 8-byte words: a boolean block (1-byte storage) must be converted with
        ``to_dtype`` first -- fail loudly instead of reading past its buffer."""
        for b in blocks:
            if b is not None and b.is_bool:
                raise TypeError(f'{what}: boolean block where a float64 / complex128 block is required (use to_dtype)')

    def _vec_descs(self, xs, ys=None, outs=None):
        self._numeric_only(xs, 'elementwise / reduction kernel')
        if ys is not None:
            self._numeric_only(ys, 'elementwise / reduction kernel')
        n = len(xs)
        arr = np.zeros(max(n, 1), dtype=_lib.VEC_DTYPE)
        if n:
            arr['x'][:n] = [x.ptr for x in xs]
            if ys is not None:
                arr['y'][:n] = [y.ptr if y is not None else 0 for y in ys]
            if outs is not None:
                arr['out'][:n] = [o.ptr for o in outs]
            arr['n'][:n] = [x.size for x in xs]
        return arr.ctypes.data_as(C.POINTER(_lib.VecDesc))  # the ctypes pointer keeps `arr` alive

    def _reduce(self, fn, xs, ys=None, n_results=1):
        res = self.ctx.empty(n_results)
        self.ctx.sync_stream()
        _lib.check(fn(self.ctx.handle, self._vec_descs(xs, ys), len(xs), C.c_void_p(res.data_ptr())))
        return self.ctx.d2h(res, n_results, np.float64)

    def _as_float_lists(self, blocks):
        """contiguous float64 aliases of a block list (complex blocks count twice as many elements)."""
        c = self.contiguous_many(blocks)
        return [self.reshape(self._fview(x), (-1,)) if x.is_complex else x for x in c]

    def inner_many(self, a_blocks, b_blocks):
        """sum_i <a_i, b_i> = sum conj(a) b over a block list: one launch + one 8-byte D2H
        (abelian.cpp:2159-2211).  Complex lists: the real part is the same reduction over the interleaved
        storage, the imaginary part sum(ar bi - ai br) is two more reductions over the real / imaginary planes."""
        for x, y in zip(a_blocks, b_blocks):
            if x.shape != y.shape:
                raise ValueError('inner: shape mismatch')
        if not a_blocks:
            return 0.0
        cplx = any(x.is_complex for x in a_blocks) or any(y.is_complex for y in b_blocks)
        if not cplx:
            a = self.contiguous_many(a_blocks)
            b = self.contiguous_many(b_blocks)
            return float(self._reduce(self.lib.cyb_dot_batched_f64, a, b)[0])
        a_blocks = [self.as_complex(x) for x in a_blocks]
        b_blocks = [self.as_complex(y) for y in b_blocks]
        re = float(self._reduce(self.lib.cyb_dot_batched_f64, self._as_float_lists(a_blocks), self._as_float_lists(b_blocks))[0])
        ar = self.contiguous_many([self._plane(x, 0) for x in a_blocks])
        ai = self.contiguous_many([self._plane(x, 1) for x in a_blocks])
        br = self.contiguous_many([self._plane(y, 0) for y in b_blocks])
        bi = self.contiguous_many([self._plane(y, 1) for y in b_blocks])
        im = float(self._reduce(self.lib.cyb_dot_batched_f64, ar, bi)[0]) - float(self._reduce(self.lib.cyb_dot_batched_f64, ai, br)[0])
        return complex(re, im)

    def norm_many(self, blocks) -> float:
        """2-norm of a whole block list (abelian.cpp:2781-2792)."""
        a = self._as_float_lists(blocks)
        if not a:
            return 0.0
        return float(np.sqrt(self._reduce(self.lib.cyb_dot_batched_f64, a)[0]))

    def norm(self, a: HipBlock, order=2, axis=None) -> float:
        """``np.linalg.norm(a.ravel(), ord=order)`` (numpy.cpp:898-913): the vector norms of numpy -- 2 in one reduction,
        inf / -inf / 0 / 1 / any p composed from abs, pow and the deterministic reductions.  With `axis` (numpy.cpp:904,
        ``np.linalg.norm(a, ord=order, axis=axis)``): the vector norm along that axis as a block -- 2 / 1 / 0 / any finite p
        through `sum` (one grouped GEMM with a vector of ones); the max-type orders +-inf have no per-axis reduction on the
        device path."""
        if axis is not None:
            mag = self.abs(a)
            if order is None or float(order) == 2.0:
                return self.sqrt(self.sum(self.multiply_blocks(mag, mag), axis))
            order = float(order)
            if order in (np.inf, -np.inf):
                raise NotImplementedError('HipBlockBackend.norm: per-axis max / min norms are not on the device path')
            if order == 0.0:
                return self.sum(self.to_dtype(self._compare(mag, 0.0, 5), 'float64'), axis)
            if order == 1.0:
                return self.sum(mag, axis)
            return self._pow(self.sum(self._pow(mag, order), axis), 1.0 / order)
        if order is None or order == 2:
            return self.norm_many([a])
        if a.size == 0:
            return 0.0
        order = float(order)
        mag = self.abs(a)
        if order == np.inf:
            return self.max(mag)
        if order == -np.inf:
            return self.min(mag)
        if order == 0.0:
            return float(self._count_true(self._compare(mag, 0.0, 5)))
        if order == 1.0:
            return self.sum_all(mag)
        return float(self.sum_all(self._pow(mag, order)) ** (1.0 / order))

    def inner(self, a: HipBlock, b: HipBlock, do_dagger: bool) -> float:
        """numpy.cpp:815-842. do_dagger: sum conj(a)[i...] b[i...]; else sum a[i, j, ...] b[..., j, i] WITHOUT conjugation
        (``np.tensordot(a, b, [range, reversed range])``)."""
        if a.ndim != b.ndim:
            raise ValueError('a and b must have the same number of dimensions')
        if not do_dagger:
            a = self.permute_axes(a, list(range(a.ndim - 1, -1, -1)))
            if a.is_complex:  # inner_many computes sum conj(x) y: undo the conjugation on the a side
                a = self.conj(a)
        return self.inner_many([a], [b])

    def max_abs_many(self, blocks) -> float:
        blocks = [self._cunary(x, 0) if x.is_complex else x for x in blocks]
        a = self.contiguous_many(blocks)
        if not a:
            return 0.0
        return float(self._reduce(self.lib.cyb_maxabs_batched_f64, a)[0])

    def max_abs(self, a: HipBlock) -> float:
        return self.max_abs_many([a])

    def sum_all(self, a: HipBlock) -> float:
        if a.is_bool:
            return self._count_true(a)
        ones = self.ones_block(a.shape)
        return self.inner_many([a], [ones])

    def item(self, a: HipBlock) -> float:
        if a.size != 1:
            raise ValueError('item(): block has more than one entry')
        return float(self.ctx.d2h(a.buf, 1, np.float64, a.offset)[0])

    def get_block_element(self, a: HipBlock, idcs) -> float:
        return self.item(self.get_item(a, tuple(int(i) for i in idcs)))

    def linear_combination_many(self, a_coef, vs, b_coef, ws):
        """a*v + b*w on block lists, one launch (numpy.cpp:1358-1365, abelian.cpp:2254-2302).  Real blocks
        with real coefficients and complex blocks with real coefficients both run through the float64 kernel
        (on the interleaved storage); complex coefficients use the complex kernel."""
        cplx = (any(x.is_complex for x in vs) or any(x.is_complex for x in ws) or isinstance(a_coef, complex)
                or isinstance(b_coef, complex))
        if not cplx:
            vs = self.contiguous_many(vs)
            ws = self.contiguous_many(ws)
            outs = self._new_many([v.shape for v in vs])
            if vs:
                self.ctx.sync_stream()
                _lib.check(self.lib.cyb_axpby_batched_f64(self.ctx.handle, self._vec_descs(vs, ws, outs), len(vs),
                                                          float(a_coef), float(b_coef)))
            return outs
        vs = self.contiguous_many([self.as_complex(v) for v in vs])
        ws = self.contiguous_many([self.as_complex(w) for w in ws])
        outs = self._new_many([v.shape for v in vs], True)
        if vs:
            a_c, b_c = complex(a_coef), complex(b_coef)
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_axpby_batched_c128(self.ctx.handle, self._vec_descs(vs, ws, outs), len(vs),
                                                       a_c.real, a_c.imag, b_c.real, b_c.imag))
        return outs

    def linear_combination(self, a_coef, v, b_coef, w):
        return self.linear_combination_many(a_coef, [v], b_coef, [w])[0]

    def mul_many(self, a, blocks):
        cplx = isinstance(a, complex) or any(b.is_complex for b in blocks)
        if cplx:
            bs = self.contiguous_many([self.as_complex(b) for b in blocks])
            outs = self._new_many([b.shape for b in bs], True)
            if bs:
                a_c = complex(a)
                self.ctx.sync_stream()
                _lib.check(self.lib.cyb_axpby_batched_c128(self.ctx.handle, self._vec_descs(bs, None, outs), len(bs),
                                                           a_c.real, a_c.imag, 0.0, 0.0))
            return outs
        bs = self.contiguous_many(blocks)
        outs = self._new_many([b.shape for b in bs])
        if bs:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_axpby_batched_f64(self.ctx.handle, self._vec_descs(bs, None, outs), len(bs), float(a), 0.0))
        return outs

    def mul(self, a, b: HipBlock) -> HipBlock:
        return self.mul_many(a, [b])[0]

    def _binary(self, a: HipBlock, b: HipBlock, op: int) -> HipBlock:
        if a.shape != b.shape:
            raise ValueError(f'elementwise op: shape mismatch {a.shape} vs {b.shape}')
        if a.is_complex or b.is_complex:
            if op == 0:
                return self.linear_combination(1.0, a, 1.0, b)
            if op == 1:
                return self.linear_combination(1.0, a, -1.0, b)
            a, b = self.contiguous_many([self.as_complex(a), self.as_complex(b)])
            out = self._new(a.shape, True)
            if a.size:
                self.ctx.sync_stream()
                _lib.check(self.lib.cyb_elementwise_batched_c128(self.ctx.handle, self._vec_descs([a], [b], [out]), 1, 3 + op))
            return out
        a, b = self.contiguous_many([a, b])
        out = self._new(a.shape)
        if a.size:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_binary_batched_f64(self.ctx.handle, self._vec_descs([a], [b], [out]), 1, op))
        return out

    def multiply_blocks(self, a, b):
        return self._binary(a, b, 2)

    def _cunary(self, a: HipBlock, op: int) -> HipBlock:
        """complex elementwise function; op 0 (abs) and 4 (angle) give float64 blocks."""
        a = self.contiguous(a)
        out = self._new(a.shape, op not in (0, 4))
        if a.size:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_elementwise_batched_c128(self.ctx.handle, self._vec_descs([a], None, [out]), 1, op))
        return out

    def _unary(self, a: HipBlock, op: int) -> HipBlock:
        if a.is_complex:
            if op > 3:
                raise NotImplementedError('this elementwise function is on the device path for float64 blocks')
            return self._cunary(a, op)  # abs, sqrt, exp, log share their op codes
        a = self.contiguous(a)
        out = self._new(a.shape)
        if a.size:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_unary_batched_f64(self.ctx.handle, self._vec_descs([a], None, [out]), 1, op))
        return out

    def abs(self, a):
        return self._unary(a, 0)

    def sqrt(self, a):
        return self._unary(a, 1)

    def exp(self, a):
        return self._unary(a, 2)

    def log(self, a):
        return self._unary(a, 3)

    def scale_axis_many(self, items):
        """[(block, factors_1d_block, axis)] -> scaled blocks, ONE launch (numpy.cpp:1373-1385)."""
        outs = []
        descs = (_lib.ScaleAxisDesc * max(len(items), 1))()
        blocks = self.contiguous_many([it[0] for it in items])
        facs = self.contiguous_many([it[1] for it in items])
        self._numeric_only(blocks + facs, 'scale_axis')
        for i, ((_, _, axis), a, f) in enumerate(zip(items, blocks, facs)):
            if f.is_complex:
                raise NotImplementedError('scale_axis with complex factors is not on the device path yet')
            outer, ax, inner = self._as_3d(a, axis)
            if f.size != ax:
                raise ValueError('scale_axis: factors do not match the axis')
            out = self._new(a.shape, a.is_complex)
            if a.is_complex:  # interleaved storage: the (re, im) pair is one more inner axis
                inner *= 2
            descs[i].x, descs[i].f, descs[i].out = a.ptr, f.ptr, out.ptr
            descs[i].outer, descs[i].axis, descs[i].inner = outer, ax, inner
            outs.append(out)
        if items:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_scale_axis_batched_f64(self.ctx.handle, descs, len(items)))
        return outs

    def scale_axis(self, block, factors, axis):
        if factors.is_complex:
            # (br + i bi)(fr + i fi): four real scalings of the planes, recombined (complex factors are rare: the
            # singular values and eigenvalues this path scales with are real)
            b = self.as_complex(block)
            br, bi = self.copy_block(self.real(b)), self.copy_block(self.imag(b))
            fr, fi = self.copy_block(self.real(factors)), self.copy_block(self.imag(factors))
            rr, ii, ri, ir = self.scale_axis_many([(br, fr, axis), (bi, fi, axis), (br, fi, axis), (bi, fr, axis)])
            out = self._new(b.shape, True)
            self.copy_many([(self._plane(out, 0), self.linear_combination(1.0, rr, -1.0, ii)),
                            (self._plane(out, 1), self.linear_combination(1.0, ri, 1.0, ir))])
            return out
        return self.scale_axis_many([(block, factors, axis)])[0]

    def allclose(self, a, b, rtol=1e-5, atol=1e-8) -> bool:
        diff = self.linear_combination(1.0, a, -1.0, b)
        return self.max_abs(diff) <= atol + rtol * self.max_abs(b)

    # ------------------------------------------------------------------ the hot path: GEMM
    def _matrix_view(self, a: HipBlock):
        """(ptr, rows, cols, row_stride, col_stride) of a 2-D block with a unit stride, copying if
        the view has none (the kernel reads row- or column-major views in place)."""
        if a.ndim != 2:
            raise ValueError('matrix operand must be 2-D')
        if a.is_bool:
            raise TypeError('matrix_dot: boolean block where a float64 / complex128 block is required (use to_dtype)')
        rs, cs = a.strides
        m, n = a.shape
        ok = (cs == 1 or n == 1) or (rs == 1 or m == 1)
        if not ok:
            a = self.contiguous(a)
            rs, cs = a.strides
        return a, (a.ptr, m, n, rs, cs)

    def make_gemm_plan(self, groups, outs=None, enqueue=False):
        """Plan ``out_g = sum_{(a,b) in groups[g]} a @ b`` for every group g (2-D blocks).

        ``groups`` is a list of lists of (a, b) pairs -- the K-split pairs that the reference
        accumulates with ``Block.__add__`` (abelian.cpp:1437-1446) form one group."""
        if any(a.is_complex or b.is_complex for g in groups for a, b in g):
            return self._complex_gemm(groups, outs, enqueue)
        n = len(groups)
        if outs is None:
            if any(not g for g in groups):
                raise ValueError('empty GEMM group')
            outs = self._new_many([(g[0][0].shape[0], g[0][1].shape[1]) for g in groups])
        # descriptor columns are gathered in Python lists and written through numpy views of the C structs (one
        # vectorised store per field instead of one ctypes store per field and block)
        keep, seg_rows, prob_rows = [], [], []
        s = 0
        for g, out in zip(groups, outs):
            M, N = out.shape
            if out.strides[1] != 1 and N > 1:
                raise ValueError('GEMM output must have unit column stride')
            s0 = s
            for a, b in g:
                # (the common case inline: 2-D float64 views with a unit stride -- 7280 operands in the U(1)xU(1) theta list)
                sa, sb = a.shape, b.shape
                if len(sa) == 2 and len(sb) == 2 and not a.is_bool and not b.is_bool:
                    (am, ak), (bk, bn) = sa, sb
                    (ars, acs), (brs, bcs) = a.strides, b.strides
                    if not ((acs == 1 or ak == 1) or (ars == 1 or am == 1)):
                        a, (pa, am, ak, ars, acs) = self._matrix_view(a)
                    else:
                        pa = a.ptr
                    if not ((bcs == 1 or bn == 1) or (brs == 1 or bk == 1)):
                        b, (pb, bk, bn, brs, bcs) = self._matrix_view(b)
                    else:
                        pb = b.ptr
                else:
                    a, (pa, am, ak, ars, acs) = self._matrix_view(a)
                    b, (pb, bk, bn, brs, bcs) = self._matrix_view(b)
                if am != M or bn != N or ak != bk:
                    raise ValueError(f'shapes {a.shape} and {b.shape} not aligned for output {out.shape}')
                keep.append(a)
                keep.append(b)
                seg_rows.append((pa, pb, ak, ars, acs, brs, bcs))
                s += 1
            prob_rows.append((out.ptr, M, N, out.strides[0] if M > 1 else max(N, 1), s0, s))
        nseg = s
        probs_np = np.zeros(max(n, 1), dtype=_lib.GEMM_PROB_DTYPE)
        segs_np = np.zeros(max(nseg, 1), dtype=_lib.GEMM_SEG_DTYPE)
        if n:
            cols = np.array(prob_rows, dtype=np.uint64).T
            for name, col in zip(('C', 'M', 'N', 'ldc', 'seg_begin', 'seg_end'), cols):
                probs_np[name][:n] = col
            probs_np['alpha'][:n] = 1.0
        if nseg:
            cols = np.array(seg_rows, dtype=np.int64).T
            for name, col in zip(('A', 'B', 'K', 'a_rs', 'a_cs', 'b_rs', 'b_cs'), cols):
                segs_np[name][:nseg] = col
        probs = probs_np.ctypes.data_as(C.POINTER(_lib.GemmProb))
        segs = segs_np.ctypes.data_as(C.POINTER(_lib.GemmSeg))
        keep.append((probs_np, segs_np))
        if enqueue:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_gemm_grouped_enqueue_f64(self.ctx.handle, probs, n, segs, nseg))
            return list(outs)
        handle = C.c_void_p()
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_gemm_plan_create(self.ctx.handle, C.byref(handle), probs, n, segs, nseg))
        return GemmPlan(self, handle, list(outs), keep)

    def _complex_gemm(self, groups, outs, enqueue):
        """complex128 groups through the real grouped GEMM (include/cyten_amd.h, complex section): A is read in
        place as a real M x 2K matrix, B is expanded once into the real 2K x 2N matrix [[br, bi], [-bi, br]],
        C is written in place as a real M x 2N matrix -- 8 M N K real flops, no waste.  Real operands of a mixed
        product are promoted first."""
        n_b = sum(len(g) for g in groups)
        a_list = self.contiguous_many([self.as_complex(a) for g in groups for a, _ in g])
        b_list = [self.as_complex(b) for g in groups for _, b in g]
        for b in b_list:
            if b.ndim != 2:
                raise ValueError('matrix operand must be 2-D')
        b_exp = self._new_many([(2 * b.shape[0], 2 * b.shape[1]) for b in b_list])   # one buffer for all expansions
        if n_b:
            arr = np.zeros(n_b, dtype=_lib.CEXPAND_DTYPE)   # written through a numpy view of the C structs (replayable)
            arr['src'] = [b.ptr for b in b_list]
            arr['rs'] = [b.strides[0] for b in b_list]
            arr['cs'] = [b.strides[1] for b in b_list]
            arr['K'] = [b.shape[0] for b in b_list]
            arr['N'] = [b.shape[1] for b in b_list]
            arr['dst'] = [e.ptr for e in b_exp]
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_complex_expand_batched_f64(self.ctx.handle, arr.ctypes.data_as(C.POINTER(_lib.CExpandDesc)), n_b))
        c_outs = outs
        if c_outs is None:
            c_outs = self._new_many([(g[0][0].shape[0], g[0][1].shape[1]) for g in groups], True)
        rgroups, k = [], 0
        for g in groups:
            rg = []
            for _ in g:
                a = a_list[k]
                M, K = a.shape
                rg.append((self.reshape(self._fview(a), (M, 2 * K)), b_exp[k]))
                k += 1
            rgroups.append(rg)
        f_outs = []
        for c in c_outs:
            if not (c.is_complex and c.is_contiguous()):
                raise ValueError('complex GEMM output must be a contiguous complex block')
            f_outs.append(self.reshape(self._fview(c), (c.shape[0], 2 * c.shape[1])))
        res = self.make_gemm_plan(rgroups, f_outs, enqueue)
        if enqueue:
            return list(c_outs)
        res.outs = list(c_outs)
        res._keepalive = (res._keepalive, a_list, b_exp)
        return res

    def matrix_dot_grouped(self, groups, outs=None):
        """All result blocks of one contraction in ONE asynchronous launch (no plan object, no device
        allocation, no host synchronisation: descriptors travel through the pinned upload ring)."""
        if not groups:
            return []
        return self.make_gemm_plan(groups, outs, enqueue=True)

    def matrix_dot(self, a: HipBlock, b: HipBlock) -> HipBlock:
        """As ``np.dot`` (numpy.cpp:1218-1225): matrix/vector operands."""
        if a.ndim == 2 and b.ndim == 2:
            return self.matrix_dot_grouped([[(a, b)]])[0]
        if a.ndim == 1 and b.ndim == 1:
            return self.reshape(self.matrix_dot_grouped([[(self.reshape(a, (1, -1)), self.reshape(b, (-1, 1)))]])[0], ())
        if a.ndim == 2 and b.ndim == 1:
            return self.reshape(self.matrix_dot_grouped([[(a, self.reshape(b, (-1, 1)))]])[0], (a.shape[0],))
        if a.ndim == 1 and b.ndim == 2:
            return self.reshape(self.matrix_dot_grouped([[(self.reshape(a, (1, -1)), b)]])[0], (b.shape[1],))
        raise ValueError('matrix_dot: operands must be 1-D or 2-D')

    def tdot(self, a: HipBlock, b: HipBlock, idcs_a, idcs_b) -> HipBlock:
        """As ``np.tensordot`` (numpy.cpp:1118-1129): permute (views), reshape, one GEMM."""
        idcs_a = [i % a.ndim for i in idcs_a]
        idcs_b = [i % b.ndim for i in idcs_b]
        if len(idcs_a) != len(idcs_b) or any(a.shape[i] != b.shape[j] for i, j in zip(idcs_a, idcs_b)):
            raise ValueError('tdot: shape mismatch on the contracted axes')
        keep_a = [i for i in range(a.ndim) if i not in idcs_a]
        keep_b = [j for j in range(b.ndim) if j not in idcs_b]
        K = math.prod(map(int, [a.shape[i] for i in idcs_a])) if idcs_a else 1
        a2 = self.reshape(self.permute_axes(a, keep_a + idcs_a), (-1, K)) if a.size else self.zeros((0, K))
        b2 = self.reshape(self.permute_axes(b, idcs_b + keep_b), (K, -1)) if b.size else self.zeros((K, 0))
        out_shape = [a.shape[i] for i in keep_a] + [b.shape[j] for j in keep_b]
        if K == 0 or a2.shape[0] == 0 or b2.shape[1] == 0:
            return self.zeros(out_shape)
        return self.reshape(self.matrix_dot_grouped([[(a2, b2)]])[0], out_shape)

    def outer(self, a, b):
        return self.tdot(a, b, [], [])

    def kron(self, a, b):
        """numpy.cpp kron: out[(i k),(j l)] = a[i,j] b[k,l] for 2-D blocks."""
        o = self.tdot(a, b, [], [])
        return self.reshape(self.permute_axes(o, [0, 2, 1, 3]), (a.shape[0] * b.shape[0], a.shape[1] * b.shape[1]))

    # ------------------------------------------------------------------ the hot path: decompositions
    def matrix_svd_batched(self, blocks, algorithm=None, return_info=False, outs=None, null_vectors=True, return_rank=False):
        """Thin SVD of every 2-D block of a list in one batched call.
        Returns [(U, S, Vh)], S descending (scipy.linalg.svd(full_matrices=False) conventions,
        numpy.cpp:1247-1297). All reference algorithm names are accepted and map to the
        block-Jacobi kernel."""
        if algorithm is not None and algorithm not in self.svd_algorithms:
            raise ValueError(f'SVD algorithm not supported: {algorithm}')
        self._numeric_only(blocks, 'decomposition')
        cplx = any(b.is_complex for b in blocks)
        if cplx:  # the whole list in complex arithmetic (small blocks only: csrc/csvd_small.hip)
            if outs is not None:
                raise NotImplementedError('matrix_svd_batched: preallocated outputs are for float64 blocks')
            blocks = [self.as_complex(b) for b in blocks]
        n = len(blocks)
        srcs = self.contiguous_many(blocks)
        for a in srcs:
            if a.ndim != 2:
                raise ValueError('matrix_svd: block must be 2-D')
        given = outs
        if cplx and n:
            # large blocks: the float64 block engine on the interleaved embedding (also the truncating caller's form: null
            # vectors skipped, numerical ranks reported); small ones (and lists the engine refuses): the complex Jacobi
            # kernels below, which always complete and report k
            big = [i for i, a in enumerate(srcs) if min(a.shape) >= self.COMPLEX_SVD_EMBED_MIN]
            got = self._complex_svd_embedded([srcs[i] for i in big], True, null_vectors) if big else None
            if got is not None:
                res_big, info_big, rank_big = got
                rest = [i for i in range(n) if i not in set(big)]
                res_all, info_all, rank_all = [None] * n, [0] * n, [min(a.shape) for a in srcs]
                for i, r, f, rk in zip(big, res_big, info_big, rank_big):
                    res_all[i], info_all[i], rank_all[i] = r, f, rk
                if rest:
                    rres, rinfo = self.matrix_svd_batched_complex_direct([srcs[i] for i in rest], True)
                    for i, r, f in zip(rest, rres, rinfo):
                        res_all[i], info_all[i] = r, f
                out = (res_all,) + ((info_all,) if return_info else ()) + ((rank_all,) if return_rank else ())
                return out if len(out) > 1 else res_all
        if given is None and cplx:
            cs, rs = [], []
            for a in srcs:
                m, nn = a.shape
                k = min(m, nn)
                cs += [(m, k), (k, nn)]
                rs.append((k,))
            cflat, rflat = self._new_many(cs, True), self._new_many(rs)
            outs = [(cflat[2 * i], rflat[i], cflat[2 * i + 1]) for i in range(n)]
        elif given is None:  # U, S, Vh of all blocks out of one buffer
            shapes = []
            for a in srcs:
                m, nn = a.shape
                k = min(m, nn)
                shapes += [(m, k), (k,), (k, nn)]
            flat = self._new_many(shapes)
            outs = [tuple(flat[3 * i:3 * i + 3]) for i in range(n)]
        else:
            outs = []
            for a, (U, S, Vh) in zip(srcs, given):
                m, nn = a.shape
                k = min(m, nn)
                if U.shape != (m, k) or S.shape != (k,) or Vh.shape != (k, nn) or not (
                        U.is_contiguous() and S.is_contiguous() and Vh.is_contiguous()):
                    raise ValueError('matrix_svd_batched: outs[i] must be contiguous (m,k), (k,), (k,n) blocks')
                outs.append((U, S, Vh))
        arr = np.zeros(max(n, 1), dtype=_lib.SVD_DTYPE)
        if n:
            ms = np.array([a.shape[0] for a in srcs], dtype=np.int64)
            ns = np.array([a.shape[1] for a in srcs], dtype=np.int64)
            ks = np.minimum(ms, ns)
            arr['A'][:n], arr['m'][:n], arr['n'][:n] = [a.ptr for a in srcs], ms, ns
            arr['lda'][:n] = arr['ldvh'][:n] = np.maximum(ns, 1)
            arr['ldu'][:n] = np.maximum(ks, 1)
            arr['U'][:n], arr['S'][:n], arr['Vh'][:n] = ([o[0].ptr for o in outs], [o[1].ptr for o in outs], [o[2].ptr for o in outs])
        descs = arr.ctypes.data_as(C.POINTER(_lib.SvdDesc))
        info = (C.c_int32 * max(n, 1))()
        rank = (C.c_int32 * max(n, 1))()
        if n:
            self.ctx.sync_stream()
            if cplx or (null_vectors and not return_rank):
                fn = self.lib.cyb_svd_batched_c128 if cplx else self.lib.cyb_svd_batched_f64
                _lib.check(fn(self.ctx.handle, descs, n, info if return_info else None))
                for i in range(n):
                    rank[i] = min(srcs[i].shape)
            else:   # the truncating caller's form: null vectors skipped and / or the numerical ranks reported
                _lib.check(self.lib.cyb_svd_batched_ex_f64(self.ctx.handle, descs, n, info if return_info else None,
                                                           0 if null_vectors else _lib.CYB_SVD_SKIP_NULL_VECTORS, rank))
        res = (outs,)
        if return_info:
            res += (list(info)[:n],)
        if return_rank:
            res += (list(rank)[:n],)
        return res if len(res) > 1 else outs

    def matrix_svd_batched_complex_direct(self, srcs, return_info=False):
        """The complex Jacobi kernels (`cyb_svd_batched_c128`) on contiguous complex blocks, without the embedded route."""
        keep = self.COMPLEX_SVD_EMBED_MIN
        self.COMPLEX_SVD_EMBED_MIN = 1 << 62
        try:
            return self.matrix_svd_batched(srcs, return_info=return_info)
        finally:
            self.COMPLEX_SVD_EMBED_MIN = keep

    def matrix_svd(self, a: HipBlock, algorithm=None):
        return self.matrix_svd_batched([a], algorithm)[0]

    def matrix_qr_batched(self, blocks, full=False):
        """QR of every 2-D block (scipy.linalg.qr mode 'economic'/'full', numpy.cpp:1236-1245)."""
        self._numeric_only(blocks, 'decomposition')
        cplx = any(b.is_complex for b in blocks)
        if cplx:
            blocks = [self.as_complex(b) for b in blocks]
        n = len(blocks)
        srcs = self.contiguous_many(blocks)
        if cplx and n:
            # large blocks: the real MFMA block engine on the interleaved embedding, economic and full; blocks that route gives
            # up (it verifies unitarity and reconstruction per block) and all small ones: complex Householder QR
            # (csrc/cqr_house.hip), which is backward stable for every block
            big = [i for i, a in enumerate(srcs) if a.ndim == 2 and min(a.shape) > 0
                   and (min(a.shape) >= self.COMPLEX_QR_EMBED_MIN or max(a.shape) > 128) and a.shape[0] <= self.COMPLEX_QR_EMBED_MAX_ROWS]
            if big:
                got = self._complex_qr_embedded([srcs[i] for i in big], full)
                done = {i: g for i, g in zip(big, got) if g is not None}
                rest = [i for i in range(n) if i not in done]
                if rest:
                    rest_out = self.matrix_qr_batched_direct([srcs[i] for i in rest], full, True)
                    done.update(dict(zip(rest, rest_out)))
                return [done[i] for i in range(n)]
        return self.matrix_qr_batched_direct(srcs, full, cplx)

    def matrix_qr_batched_direct(self, srcs, full, cplx):
        """The QR kernels of the C-ABI on contiguous 2-D blocks of one dtype (`cyb_qr_batched_f64` / `_c128`)."""
        n = len(srcs)
        shapes = []
        for a in srcs:
            if a.ndim != 2:
                raise ValueError('matrix_qr: block must be 2-D')
            m, nn = a.shape
            kq = m if full else min(m, nn)
            shapes += [(m, kq), (kq, nn)]
        flat = self._new_many(shapes, cplx)
        outs = [tuple(flat[2 * i:2 * i + 2]) for i in range(n)]
        arr = np.zeros(max(n, 1), dtype=_lib.QR_DTYPE)
        if n:
            ms = np.array([a.shape[0] for a in srcs], dtype=np.int64)
            ns = np.array([a.shape[1] for a in srcs], dtype=np.int64)
            kqs = ms if full else np.minimum(ms, ns)
            arr['A'][:n], arr['m'][:n], arr['n'][:n] = [a.ptr for a in srcs], ms, ns
            arr['lda'][:n] = arr['ldr'][:n] = np.maximum(ns, 1)
            arr['ldq'][:n] = np.maximum(kqs, 1)
            arr['Q'][:n], arr['R'][:n] = [o[0].ptr for o in outs], [o[1].ptr for o in outs]
            arr['full'][:n] = int(full)
        descs = arr.ctypes.data_as(C.POINTER(_lib.QrDesc))
        if n:
            self.ctx.sync_stream()
            _lib.check((self.lib.cyb_qr_batched_c128 if cplx else self.lib.cyb_qr_batched_f64)(self.ctx.handle, descs, n))
        return outs

    def matrix_qr(self, a: HipBlock, full: bool):
        return self.matrix_qr_batched([a], full)[0]

    # complex blocks with min(m, n) at least this large take the embedded route of `_complex_qr_embedded`
    COMPLEX_QR_EMBED_MIN = 48
    # ... up to this many rows (the embedding has twice as many; beyond 1536 real rows the blocked QR spreads a panel over
    # several workgroups, qr_panel_multi_kernel, which must all be resident: 256 CUs x 1536 rows)
    COMPLEX_QR_EMBED_MAX_ROWS = 65536

    def _embed_complex(self, srcs):
        """Interleaved real embeddings M(A) (a + ib -> [[a, -b], [b, a]]; 2m x 2n float64) of contiguous complex 2-D blocks:
        one buffer, ONE strided launch."""
        Ms = self._new_many([(2 * a.shape[0], 2 * a.shape[1]) for a in srcs])
        items = []
        for a, M in zip(srcs, Ms):
            m, nn = a.shape
            if m * nn == 0:
                continue
            re, im = self._plane(a, 0), self._plane(a, 1)
            for off, coeff, src in ((0, 1.0, re), (1, -1.0, im), (2 * nn, 1.0, im), (2 * nn + 1, 1.0, re)):
                items.append((HipBlock(self, M.buf, M.offset + off, (m, nn), (4 * nn, 2)), [(coeff, src)], False))
        self.lincomb_many(items)
        return Ms

    def _extract_complex_items(self, X, out, by_rows=False):
        """lincomb items that read the complex matrix out of a structured embedding X into the complex block `out` (r x c):
        from the even COLUMNS of X (real part rows 0::2, imaginary part rows 1::2 -- a real column of X is one complex
        column), or with `by_rows` from the even ROWS (real part columns 0::2, imaginary part minus columns 1::2 -- a
        real row of X is one complex row)."""
        r, c = out.shape
        if r * c == 0:
            return []
        ld = X.strides[0]
        fo = self._fview(out)
        items = []
        for plane, off, coeff in (((0, 0, 1.0), (1, 1, -1.0)) if by_rows else ((0, 0, 1.0), (1, ld, 1.0))):
            dst = HipBlock(self, fo.buf, fo.offset + plane, (r, c), (2 * c, 2))
            src = HipBlock(self, X.buf, X.offset + off, (r, c), (2 * ld, 2))
            items.append((dst, [(coeff, src)], False))
        return items

    # complex blocks with min(m, n) at least this large are decomposed on the float64 block engine through the embedding
    # (measured: 96 -> 4.7 vs 4.7 ms, 192 -> 10.0 vs 11.3 ms, 1024 -> 111 vs 220 ms, sixteen 128-blocks -> 13.7 vs 21.3 ms)
    COMPLEX_SVD_EMBED_MIN = 96
    # defect |U^H U - 1| above which a factor of the embedded route is re-orthonormalised (full-rank, mildly graded blocks
    # come out at 1e-14 ... 3e-13)
    COMPLEX_SVD_ORTHO_TOL = 2e-12

    def _complex_svd_embedded(self, srcs, return_info=False, null_vectors=True):
        """Thin SVD of complex blocks on the float64 block engine: the pipeline of the real SVD --
        blocked QR, LQ step, persistent block-Jacobi sweeps, completion from Q2 -- runs on the interleaved embeddings
        with `CYB_SVD_EMBEDDED_COMPLEX`: its QR steps preserve the structure by uniqueness, its pivot solves by
        construction (a complex 16 x 16 Hermitian Jacobi solve per pair), rows are deflated / ranked / completed as
        pairs.  Returns ([(U, S, Vh)], info, ranks) in complex / float64 blocks (ranks: numerical ranks in complex rows;
        with `null_vectors=False` the vectors beyond a block's rank are unspecified, CYB_SVD_SKIP_NULL_VECTORS), or None
        if the engine refuses the list: the caller then uses the complex Jacobi kernels."""
        n = len(srcs)
        Ms = self._embed_complex(srcs)
        shapes = []
        for a in srcs:
            m, nn = a.shape
            k = min(m, nn)
            shapes += [(2 * m, 2 * k), (2 * k,), (2 * k, 2 * nn)]
        flat = self._new_many(shapes)
        arr = np.zeros(n, dtype=_lib.SVD_DTYPE)
        ms = np.array([2 * a.shape[0] for a in srcs], dtype=np.int64)
        ns = np.array([2 * a.shape[1] for a in srcs], dtype=np.int64)
        ks = np.minimum(ms, ns)
        arr['A'], arr['m'], arr['n'] = [M.ptr for M in Ms], ms, ns
        arr['lda'] = arr['ldvh'] = np.maximum(ns, 1)
        arr['ldu'] = np.maximum(ks, 1)
        arr['U'], arr['S'], arr['Vh'] = [flat[3 * i].ptr for i in range(n)], [flat[3 * i + 1].ptr for i in range(n)], [flat[3 * i + 2].ptr for i in range(n)]
        info = (C.c_int32 * n)()
        rank = (C.c_int32 * n)()
        self.ctx.sync_stream()
        st = self.lib.cyb_svd_batched_ex_f64(self.ctx.handle, arr.ctypes.data_as(C.POINTER(_lib.SvdDesc)), n, info,
                                             _lib.CYB_SVD_EMBEDDED_COMPLEX | (0 if null_vectors else _lib.CYB_SVD_SKIP_NULL_VECTORS), rank)
        if st == _lib.CYB_ERR_UNSUPPORTED:
            return None
        if st != _lib.CYB_ERR_NOCONV:   # (blocks that did not settle are caught by the reconstruction check below)
            _lib.check(st)
        cs, rs = [], []
        for a in srcs:
            m, nn = a.shape
            k = min(m, nn)
            cs += [(m, k), (k, nn)]
            rs.append((k,))
        cflat, rflat = self._new_many(cs, True), self._new_many(rs)
        items, pairs = [], []
        for i in range(n):
            U, S, Vh = flat[3 * i], flat[3 * i + 1], flat[3 * i + 2]
            # real column 2a of U and real row 2a of Vh are ONE real singular triplet: a complex triplet whatever the
            # basis the engine left inside the two-dimensional real singular subspace
            items += self._extract_complex_items(U, cflat[2 * i]) + self._extract_complex_items(Vh, cflat[2 * i + 1], by_rows=True)
            k = rflat[i].shape[0]
            if k:
                pairs.append((rflat[i], HipBlock(self, S.buf, S.offset, (k,), (2,))))
        self.lincomb_many(items)
        self.copy_many(pairs)
        # Orthonormality in the COMPLEX sense.  The even real columns of U (rows of Vh) are orthonormal as real vectors by
        # construction; as complex vectors they are as long as the real factor is structured, and that is only as good as
        # the structure of Q1's reflectors: the ones built from a trailing block near the rounding level of the matrix --
        # the null space of a rank-deficient block, the small end of a spectrum graded over many decades -- have none
        # (defect eps * sigma_max / sigma_j).  One grouped GEMM measures the defect; where it shows, a complex QR of the
        # factor (columns in order of descending sigma) restores it: U = Q_u R_u with R_u = 1 + (terms that couple only
        # columns of such small sigma_j), so Q_u S Vh is the same matrix to eps ||A||.
        cranks = [int(rank[i]) // 2 for i in range(n)]
        # (columns that count: all of them, or -- null vectors skipped -- the leading rank)
        kk = [min(srcs[i].shape) if null_vectors else cranks[i] for i in range(n)]
        todo = [i for i in range(n) if kk[i] > 0]
        if todo:
            us = [self.subblock(cflat[2 * i], 0, srcs[i].shape[0], 0, kk[i]) for i in todo]
            vs = [self.subblock(cflat[2 * i + 1], 0, kk[i], 0, srcs[i].shape[1]) for i in todo]
            uh = [self.conj(self.permute_axes(u, [1, 0])) for u in us]
            vt = [self.conj(self.permute_axes(v, [1, 0])) for v in vs]
            uc = self.contiguous_many(us)
            grams = self.matrix_dot_grouped([[(uh[j], uc[j])] for j in range(len(todo))] + [[(vs[j], vt[j])] for j in range(len(todo))])
            eyes = {}
            bad_u, bad_v = [], []
            for j, i in enumerate(todo):
                k = kk[i]
                if k not in eyes:
                    eyes[k] = self.eye_matrix(k, dtype='complex128')
                for g, lst in ((grams[j], bad_u), (grams[len(todo) + j], bad_v)):
                    if self.max_abs(self.linear_combination(1.0, g, -1.0, eyes[k])) > self.COMPLEX_SVD_ORTHO_TOL:
                        lst.append(j)
            if bad_u or bad_v:
                qs = [q for q, _ in self.matrix_qr_batched([uc[j] for j in bad_u] + [vt[j] for j in bad_v], False)]
                self.copy_many([(us[j], qs[t]) for t, j in enumerate(bad_u)])
                self.copy_many([(vs[j], self.permute_axes(qs[len(bad_u) + t], [1, 0])) for t, j in enumerate(bad_v)], conj=True)
        res, info = [(cflat[2 * i], rflat[i], cflat[2 * i + 1]) for i in range(n)], list(info)
        # Reconstruction check.  The route rests on the QR steps leaving R = M(R_c) structured, which needs the leading
        # columns of the block to be independent: a numerically dependent column in the MIDDLE (zero columns, a product of
        # block-sparse factors) gets an unstructured reflector pair and the rows of R after it are no partners any more
        # (singular values off by 1e-3, or no convergence; scripts/svd_fuzz.py seeds 52 / 53).  One grouped GEMM per list
        # finds those blocks; they go to the complex Jacobi kernels, which make no such assumption.
        if todo:
            us = [self.subblock(cflat[2 * i], 0, srcs[i].shape[0], 0, kk[i]) for i in todo]
            vs = [self.subblock(cflat[2 * i + 1], 0, kk[i], 0, srcs[i].shape[1]) for i in todo]
            ss = [HipBlock(self, rflat[i].buf, rflat[i].offset, (kk[i],), (1,)) for i in todo]
            usc = self.scale_axis_many([(u, sv, 1) for u, sv in zip(us, ss)])
            recon = self.matrix_dot_grouped([[(usc[j], vs[j])] for j in range(len(todo))])
            diffs = self.linear_combination_many(1.0, recon, -1.0, [srcs[i] for i in todo])
            failing = [i for j, i in enumerate(todo)
                       if not self.max_abs(diffs[j]) <= self.COMPLEX_SVD_RECON_TOL * np.sqrt(max(srcs[i].shape)) * self.max_abs(srcs[i])]
            if failing:
                fres, finfo = self.matrix_svd_batched_complex_direct([srcs[i] for i in failing], True)
                for i, r, f in zip(failing, fres, finfo):
                    res[i], info[i], cranks[i] = r, f, min(srcs[i].shape)
        elif st == _lib.CYB_ERR_NOCONV:
            _lib.check(st)
        return res, info, cranks

    # |U S Vh - A|_max above this (times sqrt(max(m, n)) max|A|) sends a block of the embedded route to the complex kernels
    # (structured blocks come out at 1e-15 ... 1e-14 on this scale)
    COMPLEX_SVD_RECON_TOL = 1e-12

    # unitarity defect |Q^H Q - 1| above which the factor of the embedded QR is re-orthonormalised
    COMPLEX_QR_ORTHO_TOL = 1e-12

    def _complex_qr_embedded(self, srcs, full=False, _depth=0):
        """QR of complex blocks on the real block engine: the REAL blocked Householder QR of the
        interleaved embedding M(A) -- entry a + ib -> [[a, -b], [b, a]], 2m x 2n -- IS the complex QR once the diagonal of
        R is made positive (a QR with fixed diagonal signs is unique, and M(R_c) is upper triangular in the interleaved
        column order), so the MFMA strip kernel and the register-resident panel kernels serve complex blocks unchanged.
        Q_c is the even real columns of Q (real column 2a IS complex column a), R_c the even rows of R.

        That argument needs full column rank.  Where the block is numerically rank deficient -- or only has a part at the
        level eps |A| / sigma, e.g. a low-rank block plus noise -- the reflectors built from the trailing block carry no
        (or only part of the) structure: the even columns are then still orthonormal as REAL vectors and A = Q R still holds,
        but they are not orthonormal in the complex sense (defect eps |A| / sigma_j, up to O(1)).  One grouped GEMM measures
        the defect; above `COMPLEX_QR_ORTHO_TOL` the factor is factored once more, Q_c = Q' S (a well-conditioned block:
        its embedded QR is structured to rounding), and A = Q' (S R_c) with S R_c upper triangular -- "twice is enough".
        `full`: the extra m - k columns are the even columns of the real full Q's trailing part, made orthonormal with the
        rest by the same second pass.  `scripts/complex_embedding_model.py` is the numpy check of the argument."""
        n = len(srcs)
        Ms = self._embed_complex(srcs)
        qrs = self.matrix_qr_batched(Ms, full)
        # diagonal of every R to the host: signs for the uniqueness fix
        diags = [self.contiguous(HipBlock(self, R.buf, R.offset, (min(R.shape),), (R.strides[0] + 1,))) if min(R.shape) else None
                 for _, R in qrs]
        fix_q, fix_r = [], []
        for (Q, R), d in zip(qrs, diags):
            sq = np.ones(Q.shape[1])
            sr = np.ones(R.shape[0])
            if d is not None:
                dn = self.to_numpy(d)
                sgn = np.where(dn < 0, -1.0, 1.0)
                sq[:len(sgn)] = sgn
                sr[:len(sgn)] = sgn
            fix_q.append((Q, self.as_block(sq), 1))
            fix_r.append((R, self.as_block(sr), 0))
        Qs = self.scale_axis_many(fix_q)
        Rs = self.scale_axis_many(fix_r)
        shapes = []
        for a in srcs:
            m, nn = a.shape
            kq = m if full else min(m, nn)
            shapes += [(m, kq), (kq, nn)]
        flat = self._new_many(shapes, True)
        items = []
        for i in range(n):
            items += self._extract_complex_items(Qs[i], flat[2 * i]) + self._extract_complex_items(Rs[i], flat[2 * i + 1])
        self.lincomb_many(items)
        outs = [(flat[2 * i], flat[2 * i + 1]) for i in range(n)]
        # ---- complex unitarity of the extracted factors; second pass where it is not there
        todo = [i for i in range(n) if outs[i][0].shape[1] > 0 and outs[i][0].shape[0] > 0]
        if todo and _depth < 2:
            qh = [self.conj(self.permute_axes(outs[i][0], [1, 0])) for i in todo]
            grams = self.matrix_dot_grouped([[(qh[j], outs[i][0])] for j, i in enumerate(todo)])
            eyes = {}
            diffs = []
            for j, i in enumerate(todo):
                k = outs[i][0].shape[1]
                if k not in eyes:
                    eyes[k] = self.eye_matrix(k, dtype='complex128')
                diffs.append(self.linear_combination(1.0, grams[j], -1.0, eyes[k]))
            bad = []
            if self.max_abs_many(diffs) > self.COMPLEX_QR_ORTHO_TOL:   # (one read-back for the list; per block only if needed)
                bad = [i for i, df in zip(todo, diffs) if self.max_abs(df) > self.COMPLEX_QR_ORTHO_TOL]
            if bad:
                second = self._complex_qr_embedded([outs[i][0] for i in bad], False, _depth + 1)
                newr = self.matrix_dot_grouped([[(S, outs[i][1])] for i, (_, S) in zip(bad, second)])
                for i, (Q2, _), R2 in zip(bad, second, newr):
                    outs[i] = (Q2, R2)
                if _depth == 0:   # the re-factored blocks are measured once more; what is still not unitary goes to Householder
                    qh2 = [self.conj(self.permute_axes(outs[i][0], [1, 0])) for i in bad]
                    grams2 = self.matrix_dot_grouped([[(qh2[j], outs[i][0])] for j, i in enumerate(bad)])
                    for j, i in enumerate(bad):
                        k = outs[i][0].shape[1]
                        if not self.max_abs(self.linear_combination(1.0, grams2[j], -1.0, eyes[k])) <= 1e-11:
                            outs[i] = None
                    todo = [i for i in todo if outs[i] is not None]
        if _depth == 0 and todo:
            # The structure argument also fails when a numerically DEPENDENT column sits in the middle of the block (an
            # unstructured reflector pair there leaves a complement that is not invariant either, and every later column pair
            # inherits it): A = Q_c R_c then no longer holds.  One more grouped GEMM checks the reconstruction; such blocks go
            # back to the caller (None), which uses the Gram-Schmidt kernels with their completion of dependent columns.
            prods = self.matrix_dot_grouped([[(outs[i][0], outs[i][1])] for i in todo])
            for j, i in enumerate(todo):
                scale = self.max_abs(srcs[i])
                if scale > 0.0 and self.max_abs(self.linear_combination(1.0, prods[j], -1.0, srcs[i])) > 1e-11 * scale * max(srcs[i].shape):
                    outs[i] = None
        return outs

    def matrix_lq_batched(self, blocks, full=False):
        """block_backend.cpp:1033-1040: q, r = qr(a^T); return r^T, q^T (views)."""
        qrs = self.matrix_qr_batched([self.permute_axes(a, [1, 0]) for a in blocks], full)
        return [(self.permute_axes(r, [1, 0]), self.permute_axes(q, [1, 0])) for q, r in qrs]

    def matrix_lq(self, a: HipBlock, full: bool):
        return self.matrix_lq_batched([a], full)[0]

    def _argsort_perm(self, w: np.ndarray, sort):
        """block_backend.cpp:759-781."""
        if sort in ('m<', 'SM'):
            key = np.abs(w)
        elif sort in ('m>', 'LM'):
            key = -np.abs(w)
        elif sort in ('<', 'SR', 'SA'):
            key = w
        elif sort in ('>', 'LR', 'LA'):
            key = -w
        else:
            raise ValueError(f"Unknown sort option: '{sort}'")
        return np.argsort(key, kind='stable')

    # complex Hermitian blocks at least this large are diagonalised on the float64 block engine through the embedding
    # (measured: 128 -> 4.5 vs 5.5 ms, 448 -> 24 vs 40 ms, 1024 -> 85 vs 181 ms, sixteen 256-blocks -> 13 vs 68 ms; the in-LDS
    #  kernel of csvd_small.hip serves n <= 64 in 1.6 ms)
    COMPLEX_EIGH_EMBED_MIN = 96

    def _complex_eigh_embedded(self, srcs, return_info=False):
        """np.linalg.eigh of complex Hermitian blocks on the float64 block engine: the one-sided block-Jacobi iteration
        of the real path on the rows of M(H) + shift (exactly structured: no QR step is involved) with the structured
        pivot solves of `CYB_EIGH_EMBEDDED_COMPLEX`.  Every eigenvalue comes out twice; real column 2a of the
        eigenvector matrix is complex eigenvector a.  Returns ([(w, V)], info), or None where the engine refuses the list."""
        n = len(srcs)
        Ms = self._embed_complex(srcs)
        flat = self._new_many([sh for a in srcs for sh in ((2 * a.shape[0],), (2 * a.shape[0], 2 * a.shape[0]))])
        arr = np.zeros(n, dtype=_lib.EIGH_DTYPE)
        ks = np.array([2 * a.shape[0] for a in srcs], dtype=np.int64)
        arr['A'], arr['n'] = [M.ptr for M in Ms], ks
        arr['lda'] = arr['ldv'] = np.maximum(ks, 1)
        arr['W'], arr['V'] = [flat[2 * i].ptr for i in range(n)], [flat[2 * i + 1].ptr for i in range(n)]
        info = (C.c_int32 * n)()
        self.ctx.sync_stream()
        st = self.lib.cyb_eigh_batched_ex_f64(self.ctx.handle, arr.ctypes.data_as(C.POINTER(_lib.EighDesc)), n, info,
                                              _lib.CYB_EIGH_EMBEDDED_COMPLEX)
        if st == _lib.CYB_ERR_UNSUPPORTED:
            return None
        _lib.check(st)
        wflat = self._new_many([(a.shape[0],) for a in srcs])
        vflat = self._new_many([(a.shape[0], a.shape[0]) for a in srcs], True)
        items, pairs = [], []
        for i in range(n):
            k = srcs[i].shape[0]
            items += self._extract_complex_items(flat[2 * i + 1], vflat[i])
            if k:
                pairs.append((wflat[i], HipBlock(self, flat[2 * i].buf, flat[2 * i].offset, (k,), (2,))))
        self.lincomb_many(items)
        self.copy_many(pairs)
        return list(zip(wflat, vflat)), list(info)

    def eigh_batched(self, blocks, sort=None, vectors=True, return_info=False, _embed=True):
        """Hermitian EVD of every block: [(w ascending, V)] (np.linalg.eigh, numpy.cpp:658-680)."""
        self._numeric_only(blocks, 'decomposition')
        cplx = any(b.is_complex for b in blocks)
        want_vectors = vectors
        if cplx:  # eigenvectors are always computed on the complex paths
            blocks = [self.as_complex(b) for b in blocks]
            vectors = True
        n = len(blocks)
        srcs = self.contiguous_many(blocks)
        for a in srcs:
            if a.ndim != 2 or a.shape[0] != a.shape[1]:
                raise ValueError('eigh: block must be a square matrix')
        if cplx and n and _embed:
            # large blocks: the float64 block engine on the interleaved embedding; small ones (and lists the engine refuses):
            # the complex Jacobi kernels (csrc/csvd_small.hip, csrc/csvd_large.hip)
            big = [i for i, a in enumerate(srcs) if a.shape[0] >= self.COMPLEX_EIGH_EMBED_MIN]
            got = self._complex_eigh_embedded([srcs[i] for i in big], True) if big else None
            if got is not None:
                outs, info_all = [None] * n, [0] * n
                for i, r, f in zip(big, *got):
                    outs[i], info_all[i] = r, f
                rest = [i for i in range(n) if outs[i] is None]
                if rest:
                    rres, rinfo = self.eigh_batched([srcs[i] for i in rest], None, True, True, _embed=False)
                    for i, r, f in zip(rest, rres, rinfo):
                        outs[i], info_all[i] = r, f
                return self._eigh_finish(outs, info_all, sort, want_vectors, return_info)
        shapes = []
        for a in srcs:
            shapes += [(a.shape[0],), (a.shape[0], a.shape[0])] if (vectors and not cplx) else [(a.shape[0],)]
        flat = self._new_many(shapes)
        if cplx:
            vflat = self._new_many([(a.shape[0], a.shape[0]) for a in srcs], True)
            outs = list(zip(flat, vflat))
        else:
            outs = [(flat[2 * i], flat[2 * i + 1]) for i in range(n)] if vectors else [(w, None) for w in flat]
        arr = np.zeros(max(n, 1), dtype=_lib.EIGH_DTYPE)
        if n:
            ks = np.array([a.shape[0] for a in srcs], dtype=np.int64)
            arr['A'][:n], arr['n'][:n] = [a.ptr for a in srcs], ks
            arr['lda'][:n] = arr['ldv'][:n] = np.maximum(ks, 1)
            arr['W'][:n] = [o[0].ptr for o in outs]
            if vectors:
                arr['V'][:n] = [o[1].ptr for o in outs]
        descs = arr.ctypes.data_as(C.POINTER(_lib.EighDesc))
        info = (C.c_int32 * max(n, 1))()
        if n:
            self.ctx.sync_stream()
            fn = self.lib.cyb_eigh_batched_c128 if cplx else self.lib.cyb_eigh_batched_f64
            _lib.check(fn(self.ctx.handle, descs, n, info if return_info else None))
        return self._eigh_finish(outs, list(info)[:n], sort, want_vectors or not cplx, return_info)

    def _eigh_finish(self, outs, info, sort, want_vectors, return_info):
        if not want_vectors:
            outs = [(w, None) for w, _ in outs]
        if sort is not None:
            res = []
            for W, V in outs:
                perm = self._argsort_perm(self.to_numpy(W), sort)
                W2 = self._gather_axis(W, perm, 0)
                V2 = self._gather_axis(V, perm, 1) if V is not None else None
                res.append((W2, V2))
            outs = res
        if return_info:
            return outs, info
        return outs

    def eigh(self, block: HipBlock, sort=None):
        return self.eigh_batched([block], sort)[0]

    def eigvalsh(self, block: HipBlock, sort=None):
        return self.eigh_batched([block], sort, vectors=False)[0][0]

    # ------------------------------------------------------------------ small helpers of the API
    def block_from_diagonal(self, diag: HipBlock) -> HipBlock:
        n = diag.size
        out = self.zeros((n, n), dtype=diag.dtype)
        view = HipBlock(self, out.buf, out.offset, (n,), (n + 1,))
        self.copy_many([(view, diag)])
        return out

    def get_diagonal(self, a: HipBlock, tol=None) -> HipBlock:
        n = a.shape[0]
        view = HipBlock(self, a.buf, a.offset, (n,), (a.strides[0] + a.strides[1],))
        if tol is not None:
            off = self.linear_combination(1.0, a, -1.0, self.block_from_diagonal(view))
            if self.max_abs(off) > tol:
                raise ValueError('Not a diagonal block.')
        return self.copy_block(view)

    def trace_full(self, a: HipBlock) -> float:
        return self.sum_all(self.get_diagonal(a))

    def tile(self, a: HipBlock, repeats: int) -> HipBlock:
        out = self._new((a.size * repeats,), a.is_complex)
        self.copy_many([(HipBlock(self, out.buf, out.offset + r * a.size, (a.size,), (1,)), a) for r in range(repeats)])
        return out

    # ------------------------------------------------------------------ linear combinations of views
    def lincomb_many(self, items):
        """``dst[...] = sum_t coeff_t * src_t`` (or ``+=`` with accumulate) for a list of
        ``(dst_view, [(coeff, src_view), ...], accumulate)``: ONE launch.  Views are arbitrary strided views of equal shape
        (a permuted source is just a view); destinations must not overlap.  float64 destinations take float64 sources and
        real coefficients; complex128 destinations take complex or float64 sources and complex coefficients
        (`cyb_lincomb_strided_batched_c128`: anyonic R / C symbols).  This is the device part of the tree-block updates of
        ``TreePairMapping::transform_tensor`` (fusion_tree_mapping.cpp:447-497)."""
        items = [it for it in items if it[0].size]
        if not items:
            return
        cplx = items[0][0].is_complex
        n_terms = sum(len(it[1]) for it in items)
        descs = np.zeros(len(items), dtype=_lib.LINCOMB_DTYPE)
        terms = np.zeros(max(n_terms, 1), dtype=_lib.LINTERM_C128_DTYPE if cplx else _lib.LINTERM_DTYPE)
        shp, dst, t_ptr, t_c, t_ss, t_real, tb = [], [], [], [], [], [], []
        t = 0
        for dst_v, tl, acc in items:
            if dst_v.is_bool or dst_v.is_complex != cplx:
                raise NotImplementedError('lincomb_many: the destinations of one call are all float64 or all complex128 views')
            if dst_v.ndim > _lib.CYB_MAX_NDIM:
                raise NotImplementedError(f'blocks with more than {_lib.CYB_MAX_NDIM} axes')
            pad = _ZERO_PAD[dst_v.ndim]
            shp.append(dst_v.shape + pad)
            dst.append(dst_v.strides + pad)
            tb.append((t, t + len(tl)))
            for c, src in tl:
                if src.shape != dst_v.shape:
                    raise ValueError(f'lincomb_many: shape mismatch {src.shape} vs {dst_v.shape}')
                if src.is_bool or (not cplx and (src.is_complex or (isinstance(c, complex) and c.imag != 0.0))):
                    raise NotImplementedError('lincomb_many: a float64 destination takes float64 views and real coefficients')
                t_ptr.append(src.ptr)
                t_c.append(complex(c) if cplx else float(c.real if isinstance(c, complex) else c))
                t_ss.append(src.strides + pad)
                t_real.append(0 if src.is_complex else 1)
            t += len(tl)
        descs['dst'] = [it[0].ptr for it in items]
        descs['ndim'] = [it[0].ndim for it in items]
        descs['accumulate'] = [1 if it[2] else 0 for it in items]
        descs['term_begin'], descs['term_end'] = [b for b, _ in tb], [e for _, e in tb]
        descs['shape'], descs['dst_strides'] = shp, dst
        if n_terms:
            terms['src'][:n_terms], terms['src_strides'][:n_terms] = t_ptr, t_ss
            if cplx:
                cc = np.asarray(t_c, dtype=np.complex128)
                terms['coeff_re'][:n_terms], terms['coeff_im'][:n_terms], terms['src_real'][:n_terms] = cc.real, cc.imag, t_real
            else:
                terms['coeff'][:n_terms] = t_c
        self.ctx.sync_stream()
        if cplx:
            _lib.check(self.lib.cyb_lincomb_strided_batched_c128(
                self.ctx.handle, descs.ctypes.data_as(C.POINTER(_lib.LincombDesc)), len(items),
                terms.ctypes.data_as(C.POINTER(_lib.LincombTermC128)), n_terms))
        else:
            _lib.check(self.lib.cyb_lincomb_strided_batched_f64(
                self.ctx.handle, descs.ctypes.data_as(C.POINTER(_lib.LincombDesc)), len(items),
                terms.ctypes.data_as(C.POINTER(_lib.LincombTerm)), n_terms))

    def _transform_blocks_fast(self, old_blocks, new, updates) -> bool:
        """`transform_blocks` without a view object per tree block and term: every old and new block is a plain row-major
        2-D block here, so the strides of a tree-block view follow from its leading dimension and the tree-block axes alone
        (row axes: C-strides of `dims1` times ld, column axes: C-strides of `dims2`), and the descriptor arrays are filled
        directly.  The SU(2)xU(1) F-move of cfg4 (296 tree blocks, 330 terms): 21 -> ~3 ms per tensor, all of it host time.
        Returns False (nothing done) for inputs outside this case."""
        maxd = _lib.CYB_MAX_NDIM
        blocks = list(old_blocks) + list(new)
        if any(b.ndim != 2 or b.is_bool or (b.size and (b.strides[1] != 1)) for b in blocks):
            return False
        cplx = bool(new) and new[0].is_complex           # complex destinations: complex or real sources, complex coefficients
        if any(b.is_complex != cplx for b in new) or (not cplx and any(b.is_complex for b in old_blocks)):
            return False
        n_up = len(updates)
        n_terms = sum(len(u[7]) for u in updates)
        descs = np.zeros(max(n_up, 1), dtype=_lib.LINCOMB_DTYPE)
        terms = np.zeros(max(n_terms, 1), dtype=_lib.LINTERM_C128_DTYPE if cplx else _lib.LINTERM_DTYPE)
        d_ptr, d_nd, d_tb, d_te, d_shape, d_str = [], [], [], [], [], []
        t_ptr, t_c, t_str, t_real = [], [], [], []
        esz_new = 16 if cplx else 8
        old_esz = [16 if b.is_complex else 8 for b in old_blocks]
        nptr = [b.ptr for b in new]
        nld = [b.strides[0] if b.size else 1 for b in new]
        optr = [b.ptr for b in old_blocks]
        old_ld = [b.strides[0] if b.size else 1 for b in old_blocks]
        t = 0
        for b, rows, cols, dims1, idcs1, dims2, idcs2, tl in updates:
            dims = tuple(int(x) for x in dims1) + tuple(int(x) for x in dims2)
            nd = len(dims)
            perm = [int(i) for i in idcs1] + [int(i) for i in idcs2]
            if nd > maxd or nd == 0:
                return False
            pshape = tuple(dims[i] for i in perm)
            n_row = len(idcs1)
            m_new = math.prod(pshape[:n_row])
            n_new = math.prod(pshape[n_row:])
            if (rows[1] - rows[0], cols[1] - cols[0]) != (m_new, n_new):
                raise ValueError('transform_blocks: the permuted tree block does not fit its slice')
            if m_new * n_new == 0:
                continue
            ld = nld[b]
            dst_str = tuple(x * ld for x in _c_strides(pshape[:n_row])) + _c_strides(pshape[n_row:])
            pad = _ZERO_PAD[nd]
            d_ptr.append(nptr[b] + esz_new * (rows[0] * ld + cols[0]))
            d_nd.append(nd)
            d_shape.append(pshape + pad)
            d_str.append(dst_str + pad)
            d_tb.append(t)
            n1 = len(dims1)
            rs, cs = _c_strides(dims[:n1]), _c_strides(dims[n1:])
            for coeff, k, rk, ck in tl:
                if not cplx and isinstance(coeff, complex):
                    if coeff.imag != 0.0:
                        return False
                    coeff = coeff.real
                if (rk[1] - rk[0], ck[1] - ck[0]) != (math.prod(dims[:n1]), math.prod(dims[n1:])):
                    raise ValueError('transform_blocks: a source slice does not have the tree-block shape')
                lk = old_ld[k]
                sst = tuple(x * lk for x in rs) + cs
                t_ptr.append(optr[k] + old_esz[k] * (rk[0] * lk + ck[0]))
                t_c.append(complex(coeff) if cplx else float(coeff))
                t_real.append(8 // old_esz[k] if cplx else 0)     # (1: a float64 source under a complex mapping)
                t_str.append(tuple(sst[i] for i in perm) + pad)
                t += 1
            d_te.append(t)
        n_items = len(d_ptr)
        if n_items == 0:
            return True
        descs = descs[:n_items]
        descs['dst'], descs['ndim'], descs['accumulate'] = d_ptr, d_nd, 0
        descs['term_begin'], descs['term_end'] = d_tb, d_te
        descs['shape'], descs['dst_strides'] = d_shape, d_str
        if t:
            terms['src'][:t], terms['src_strides'][:t] = t_ptr, t_str
            if cplx:
                cc = np.asarray(t_c, dtype=np.complex128)
                terms['coeff_re'][:t], terms['coeff_im'][:t], terms['src_real'][:t] = cc.real, cc.imag, t_real
            else:
                terms['coeff'][:t] = t_c
        self.ctx.sync_stream()
        if cplx:
            _lib.check(self.lib.cyb_lincomb_strided_batched_c128(
                self.ctx.handle, descs.ctypes.data_as(C.POINTER(_lib.LincombDesc)), n_items,
                terms.ctypes.data_as(C.POINTER(_lib.LincombTermC128)), t))
        else:
            _lib.check(self.lib.cyb_lincomb_strided_batched_f64(
                self.ctx.handle, descs.ctypes.data_as(C.POINTER(_lib.LincombDesc)), n_items,
                terms.ctypes.data_as(C.POINTER(_lib.LincombTerm)), t))
        return True

    def transform_blocks(self, old_blocks, new_shapes, updates):
        """The block arithmetic of ``TreePairMapping::transform_tensor`` (fusion_tree_mapping.cpp:391-513) for mapping
        data computed by the (host) fusion-tree layer: returns new 2-D blocks of `new_shapes`, zero except for

            new[b][rows, cols] = permute_combined_matrix(sum_i coeff_i * old[k_i][rows_i, cols_i], dims1, idcs1, dims2, idcs2)

        for every ``(b, rows, cols, dims1, idcs1, dims2, idcs2, [(coeff, k, rows_k, cols_k), ...])`` in `updates`
        (slices as (start, stop)).  One zero-filled allocation and ONE launch for the whole tensor instead of
        ``zeros`` + (``get_item`` + ``mul`` + ``+``) per term + ``permute_combined_matrix`` + ``set_item`` per tree pair.
        The result is complex128 if an old block or a coefficient is (the reference's ``dtype = to_complex(dtype)`` for a
        mapping that is not real, fusion_tree_mapping.cpp:433-436); float64 old blocks then enter as real sources."""
        cplx = any(b.is_complex for b in old_blocks) or any(
            isinstance(c, complex) and c.imag != 0.0 for u in updates for (c, _, _, _) in u[7])
        new = self.zeros_many(new_shapes, dtype='complex128' if cplx else None)
        fast = self._transform_blocks_fast(old_blocks, new, updates)
        if fast:
            return new
        items = []
        for b, rows, cols, dims1, idcs1, dims2, idcs2, terms in updates:
            dims = list(dims1) + list(dims2)
            perm = list(idcs1) + list(idcs2)
            pshape = [dims[i] for i in perm]
            target = self.get_item(new[b], (slice(*rows), slice(*cols)))
            m_new = math.prod(pshape[:len(idcs1)])
            if target.shape != (m_new, math.prod(pshape) // max(m_new, 1)):
                raise ValueError('transform_blocks: the permuted tree block does not fit its slice')
            st = _nocopy_reshape_strides(target.shape, target.strides, tuple(pshape))
            if st is None:
                raise ValueError('transform_blocks: target slice cannot be viewed with the tree-block axes')
            tview = HipBlock(self, target.buf, target.offset, pshape, st)
            srcs = []
            for coeff, k, rk, ck in terms:
                sub = self.get_item(old_blocks[k], (slice(*rk), slice(*ck)))
                sst = _nocopy_reshape_strides(sub.shape, sub.strides, tuple(dims))
                if sst is None:
                    sub = self.contiguous(sub)
                    sst = _c_strides(dims)
                v = HipBlock(self, sub.buf, sub.offset, dims, sst)
                srcs.append((coeff, self.permute_axes(v, perm)))
            items.append((tview, srcs, False))
        self.lincomb_many(items)
        return new

    # ------------------------------------------------------------------ truncation on the device
    TRUNCATE_MAX = 65536

    def truncate_select(self, S_blocks, chi_max=None, chi_min=1, degeneracy_tol=0.0, trunc_cut=0.0, svd_min=None,
                        minimize_error=True, qdims=None):
        """``_truncate_singular_values_selection`` (tensor_backend.cpp:139-242) for the per-sector singular values
        `S_blocks` without pulling them to the host: returns (tables, mask, err, new_norm) with ``tables[s]`` a
        :class:`DeviceIndex` of the kept positions of sector s (for ``mask_gather_many``) and `mask` a boolean block
        over the concatenated values.  The host reads 16 + 8 * n_sectors bytes (err, new_norm, kept counts).
        `qdims`: the quantum-dimension weights of the non-abelian path (marginal error d * S^2, tensor_backend.cpp:158-164) --
        one number per sector, or one per value as the reference passes them (constant within a sector,
        fusion_tree_backend.cpp:2280-2303); err and new_norm are then the weighted sums.  Raises NotImplementedError for
        weights that vary inside a sector and for more than TRUNCATE_MAX values."""
        S_blocks = list(S_blocks)
        weights = None
        if qdims is not None:
            sizes = [s.size for s in S_blocks]
            q = np.asarray(qdims, dtype=np.float64).reshape(-1)
            if len(q) == sum(sizes) and len(q) != len(sizes):
                offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
                weights = np.ones(len(sizes))
                for i in range(len(sizes)):
                    seg = q[offs[i]:offs[i + 1]]
                    if len(seg):
                        if np.any(seg != seg[0]):
                            raise NotImplementedError('truncate_select: weights that vary inside a sector are host work')
                        weights[i] = seg[0]
            elif len(q) == len(sizes):
                weights = np.ascontiguousarray(q, dtype=np.float64).copy()
            else:
                raise ValueError('truncate_select: qdims must hold one weight per sector or one per singular value')
            if not np.all(weights > 0):
                raise ValueError('truncate_select: quantum dimensions are positive')
        S_blocks = self.contiguous_many(list(S_blocks))
        n_sec = len(S_blocks)
        n = sum(s.size for s in S_blocks)
        if n == 0:
            raise ValueError('truncate_select: no singular values')
        if n > self.TRUNCATE_MAX:
            raise NotImplementedError(f'truncate_select: {n} values exceed the {self.TRUNCATE_MAX} one workgroup sorts')
        opts = _lib.TruncOpts(-1 if chi_max is None else int(chi_max), int(chi_min), float(degeneracy_tol), float(trunc_cut),
                              0.0 if svd_min is None else float(svd_min), 0 if svd_min is None else 1, 1 if minimize_error else 0)
        idx = self.ctx.empty(n, 'int64')
        mask = self._new_bool((n,))
        res = self.ctx.empty(2 + n_sec)
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_truncate_select_weighted_f64(
            self.ctx.handle, self._vec_descs(S_blocks), n_sec, None if weights is None else weights.ctypes.data_as(C.c_void_p), C.byref(opts),
            C.c_void_p(idx.data_ptr()), C.c_void_p(mask.ptr), C.c_void_p(res.data_ptr())))
        raw = self.ctx.d2h(res, 2 + n_sec, np.float64)
        counts = raw[2:].view(np.int64)
        offs = np.concatenate([[0], np.cumsum([s.size for s in S_blocks])])
        tables = [DeviceIndex(idx.data_ptr() + 8 * int(offs[s]), int(counts[s]), idx) for s in range(n_sec)]
        return tables, mask, float(raw[0]), float(raw[1])

    # ------------------------------------------------------------------ rest of the operator API (block_backend.h:243-488)
    def as_scalar(self, value, dtype=None):
        """``BlockBackend::as_scalar`` (block_backend.h:243-251; numpy.cpp:330-405): the value as a 0-d DEVICE block wrapped
        in :class:`Scalar`.  Accepts Python / numpy numbers, a Scalar, or a one-element block."""
        nominal = None
        if dtype is not None and _norm_dtype(dtype) in _NOMINAL:
            nominal = _norm_dtype(dtype)
            dtype = _STORAGE_OF[nominal]
        if nominal is not None:
            sc = self.as_scalar(value, dtype)
            return Scalar(self.to_dtype(sc._blk, nominal))
        if isinstance(value, Scalar):
            value = value._blk
        if isinstance(value, HipBlock):
            if value.size != 1:
                raise ValueError('as_scalar: block has more than one entry')
            blk = self.reshape(value, ())
            if dtype is not None and np.dtype(dtype) != blk.dtype:
                blk = self.to_dtype(blk, dtype)
            return Scalar(blk)
        arr = np.asarray(value)
        if arr.size != 1:
            raise ValueError('as_scalar: value has more than one entry')
        if dtype is None:
            dtype = np.complex128 if arr.dtype.kind == 'c' else np.bool_ if arr.dtype.kind == 'b' else np.float64
        if np.dtype(dtype).kind != 'c' and arr.dtype.kind == 'c':
            arr = arr.real
        return Scalar(self.block_from_numpy(np.asarray(arr, dtype=dtype).reshape(()), dtype=dtype))

    def to_dtype(self, a: HipBlock, dtype) -> HipBlock:
        """numpy.cpp:1131-1138 (np.asarray(a, dtype)) for the six members of dtypes.h:12-21.  Device storage is float64,
        complex128 or bool; float32 / complex64 / int64 blocks are held in double words with their values rounded to the
        nominal type (cast-on-store) and carry it as their dtype (`_DtypePolicy`)."""
        want = _norm_dtype(dtype)
        if want in _NOMINAL:
            base = self.to_dtype(a, _STORAGE_OF[want])
            if base is a or base.buf is a.buf:      # (never round the caller's data in place)
                base = self.copy_block(base)
            return self._retag(base, want, True)
        if getattr(a, '_nom', None) is not None:     # widening a float32 / complex64 / int64 block is exact: the same words
            a = self._retag(a, None, False)
        kind = want.kind
        if kind == 'c':
            if a.is_bool:
                a = self.to_dtype(a, 'float64')
            return self.as_complex(a)
        if kind == 'b':
            if a.is_bool:
                return a
            if a.is_complex:  # non-zero real or imaginary part
                re, im = self.copy_block(self.real(a)), self.copy_block(self.imag(a))
                return self._compare(self.linear_combination(1.0, self.multiply_blocks(re, re), 1.0, self.multiply_blocks(im, im)), 0.0, 5)
            return self._compare(a, 0.0, 5)
        if a.is_complex:  # numpy discards the imaginary part (with a ComplexWarning)
            return self.copy_block(self.real(a))
        if a.is_bool:
            c = self.contiguous(a)
            out = self._new(a.shape)
            if a.size:
                self.ctx.sync_stream()
                _lib.check(self.lib.cyb_convert_u8_f64(self.ctx.handle, C.c_void_p(c.ptr), C.c_void_p(out.ptr), a.size))
            return out
        return a

    def _compare(self, a: HipBlock, other, op: int):
        """Block::operator< <= > >= == != (numpy.cpp:229-263): a boolean block."""
        if isinstance(other, HipBlock):
            if other.shape != a.shape:
                raise ValueError(f'comparison: shape mismatch {a.shape} vs {other.shape}')
            if a.is_complex or other.is_complex or a.is_bool or other.is_bool:
                raise NotImplementedError('comparisons are on the device path for float64 blocks')
            x, y = self.contiguous_many([a, other])
            yp, scalar = C.c_void_p(y.ptr), 0.0
        elif isinstance(other, (int, float, np.integer, np.floating)) and not isinstance(other, bool):
            if a.is_complex or a.is_bool:
                raise NotImplementedError('comparisons are on the device path for float64 blocks')
            x, yp, scalar = self.contiguous(a), None, float(other)
        else:
            return NotImplemented
        out = self._new_bool(a.shape)
        if a.size:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_compare_f64(self.ctx.handle, C.c_void_p(x.ptr), yp, scalar, C.c_void_p(out.ptr), a.size, op))
        return out

    def _count_true(self, a: HipBlock) -> int:
        if not a.is_bool:
            raise ValueError('a boolean block is required')
        c = self.contiguous(a)
        res = self.ctx.empty(1, 'int64')
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_count_nonzero_u8(self.ctx.handle, C.c_void_p(c.ptr), c.size, C.c_void_p(res.data_ptr())))
        return int(self.ctx.d2h(res, 1, np.int64)[0])

    def any(self, a: HipBlock) -> bool:
        """numpy.cpp:596-603."""
        return self._count_true(a) > 0

    def all(self, a: HipBlock) -> bool:
        """numpy.cpp:568-575."""
        return self._count_true(a) == a.size

    def _unary_param(self, a: HipBlock, op: int, param: float) -> HipBlock:
        if a.is_complex or a.is_bool:
            raise NotImplementedError('this elementwise function is on the device path for float64 blocks')
        a = self.contiguous(a)
        out = self._new(a.shape)
        if a.size:
            self.ctx.sync_stream()
            _lib.check(self.lib.cyb_unary_param_batched_f64(self.ctx.handle, self._vec_descs([a], None, [out]), 1, op, float(param)))
        return out

    def cutoff_inverse(self, a: HipBlock, cutoff: float) -> HipBlock:
        """``1 / a`` where ``abs(a) >= cutoff``, otherwise 0 (numpy.cpp:645-656)."""
        return self._unary_param(a, 0, cutoff)

    def stable_log(self, block: HipBlock, cutoff: float) -> HipBlock:
        """``log(a)`` where ``a > cutoff``, otherwise 0 (numpy.cpp:1088-1098)."""
        return self._unary_param(block, 1, cutoff)

    def angle(self, a: HipBlock) -> HipBlock:
        """numpy.cpp:587-594."""
        return self._cunary(a, 4) if a.is_complex else self._unary_param(a, 3, 0.0)

    def _pow(self, a: HipBlock, exponent) -> HipBlock:
        if isinstance(exponent, HipBlock):
            if a.is_complex or exponent.is_complex:
                raise NotImplementedError('Block::pow with complex blocks is not on the device path yet')
            return self._binary(self.to_dtype(a, 'float64') if a.is_bool else a,
                                self.to_dtype(exponent, 'float64') if exponent.is_bool else exponent, 4)
        return self._unary_param(a, 2, float(exponent))

    def _extremum(self, a: HipBlock, mode: int):
        if a.is_complex or a.is_bool:
            raise NotImplementedError('max / min / argmax are on the device path for float64 blocks')
        if a.size == 0:
            raise ValueError('zero-size block has no extremum')
        c = self.contiguous(a)
        res = self.ctx.empty(2)
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_extremum_f64(self.ctx.handle, C.c_void_p(c.ptr), c.size, mode, C.c_void_p(res.data_ptr())))
        raw = self.ctx.d2h(res, 2, np.float64)
        return float(raw[0]), int(raw[1:2].view(np.int64)[0])

    def max(self, a: HipBlock) -> float:
        """numpy.cpp:871-878."""
        return self._extremum(a, 0)[0]

    def min(self, a: HipBlock) -> float:
        """numpy.cpp:889-896."""
        return -self._extremum(a, 1)[0]

    def abs_argmax(self, block: HipBlock):
        """Indices (one per axis) of the entry of largest magnitude, first occurrence (numpy.cpp:533-550)."""
        return [int(i) for i in np.unravel_index(self._extremum(block, 2)[1], block.shape)]

    def argmin(self, block: HipBlock):
        """Indices of the smallest entry, first occurrence in C order (numpy.cpp:552-566)."""
        return [int(i) for i in np.unravel_index(self._extremum(block, 1)[1], block.shape)]

    def sum(self, a: HipBlock, ax: int) -> HipBlock:
        """np.sum(a, axis=ax) (numpy.cpp:1100-1107): the axis is moved last and contracted with a vector of ones by
        the grouped GEMM (one launch)."""
        if a.is_bool:
            a = self.to_dtype(a, 'float64')
        ax = ax % a.ndim
        rest = [k for k in range(a.ndim) if k != ax]
        out_shape = [a.shape[k] for k in rest]
        n = a.shape[ax]
        if n == 0 or a.size == 0:
            return self.zeros(out_shape, dtype=a.dtype)
        m = self.contiguous(self.permute_axes(a, rest + [ax]))
        if a.is_complex:
            re = self.sum(self._plane(m, 0), m.ndim - 1)
            im = self.sum(self._plane(m, 1), m.ndim - 1)
            out = self._new(out_shape, True)
            self.copy_many([(self._plane(out, 0), re), (self._plane(out, 1), im)])
            return out
        rows = m.size // n
        res = self.matrix_dot_grouped([[(self.reshape(m, (rows, n)), self.ones_block((n, 1)))]])[0]
        return self.reshape(res, out_shape)

    def trace_partial(self, a: HipBlock, idcs1, idcs2, remaining_idcs) -> HipBlock:
        """numpy.cpp:1166-1195: transpose to remaining + idcs1 + idcs2, fuse each group to one axis of extent T and
        take the trace over the last two axes.  The diagonal is a strided view (stride T + 1); the sum runs on device."""
        idcs1, idcs2, remaining = [i % a.ndim for i in idcs1], [i % a.ndim for i in idcs2], [i % a.ndim for i in remaining_idcs]
        t = self.permute_axes(a, remaining + idcs1 + idcs2)
        T = math.prod(a.shape[i] for i in idcs1)
        if T != math.prod(a.shape[i] for i in idcs2):
            raise ValueError('trace_partial: traced legs do not match')
        rshape = [a.shape[i] for i in remaining]
        t = self.contiguous(self.reshape(t, rshape + [T, T]))
        R = math.prod(rshape)
        diag = HipBlock(self, t.buf, t.offset, (R, T), (T * T, T + 1))
        return self.reshape(self.sum(diag, 1), rshape)

    def apply_leg_permutations(self, block: HipBlock, perms) -> HipBlock:
        """``block[np.ix_(*perms)]`` (numpy.cpp:1345-1356): one index gather per axis."""
        if len(perms) != block.ndim:
            raise ValueError('apply_leg_permutations: one permutation per axis is required')
        out = block
        for ax, p in enumerate(perms):
            p = np.asarray(p, dtype=np.int64)
            if p.ndim != 1:
                raise ValueError('permutations must be 1-D')
            if not np.array_equal(p, np.arange(block.shape[ax])):
                out = self._gather_axis(out, p, ax)
        return out

    def apply_basis_perm(self, block: HipBlock, legs, inv: bool = False) -> HipBlock:
        """block_backend.cpp:720-737: the legs' ``basis_perm`` (or ``inverse_basis_perm``) on every axis."""
        perms = []
        for leg in legs:
            if leg is None:
                raise ValueError('apply_basis_perm: leg must not be None')
            perms.append(np.asarray(leg.inverse_basis_perm if inv else leg.basis_perm, dtype=np.int64))
        return self.apply_leg_permutations(block, perms)

    def _argsort(self, block: HipBlock, axis: int = 0) -> np.ndarray:
        """np.argsort(block, axis) (numpy.cpp:614-621).  Index blocks are host int64 arrays in this mirror: every
        caller in the reference turns the result into a host index vector at once (abelian.cpp eigh / argsort)."""
        if block.is_complex:
            raise ValueError('_argsort needs a real block')
        return np.argsort(self.to_numpy(block), axis=axis, kind='stable')

    def argsort(self, block: HipBlock, sort=None, axis: int = 0) -> np.ndarray:
        """block_backend.cpp:759-781."""
        if sort is None:
            work = block
        elif sort in ('m<', 'SM'):
            work = self.abs(block)
        elif sort in ('m>', 'LM'):
            work = self.mul(-1.0, self.abs(block))
        elif sort in ('<', 'SR', 'SA'):
            work = self.real(block)
        elif sort in ('>', 'LR', 'LA'):
            work = self.mul(-1.0, self.real(block))
        elif sort == 'SI':
            work = self.imag(block)
        elif sort == 'LI':
            work = self.mul(-1.0, self.imag(block))
        else:
            raise ValueError(f"Unknown sort option: '{sort}'")
        return self._argsort(work, axis)

    def block_from_mask(self, mask, dtype=None) -> HipBlock:
        """(N, M) block with a single 1 per row at the True positions of the length-M mask (numpy.cpp:748-766):
        the identity scattered along its column axis."""
        m = np.asarray(mask.to_numpy() if isinstance(mask, HipBlock) else mask).astype(bool)
        if m.ndim != 1:
            raise ValueError('block_from_mask: 1-D mask required')
        res = self.enlarge_leg(self.eye_matrix(int(m.sum())), m, 1)
        return res if dtype is None else self.to_dtype(res, dtype)

    def get_block_mask_element(self, a, large_leg_idx: int, small_leg_idx: int, sum_block: int = 0) -> bool:
        """block_backend.cpp:739-757."""
        if not (isinstance(a, HipBlock) and a.is_bool):
            raise ValueError('a must be a boolean block')
        dim0 = a.shape[0]
        offset = (large_leg_idx // dim0) * sum_block
        large_leg_idx %= dim0
        m = self.to_numpy(a)
        if not m[large_leg_idx]:
            return False
        return small_leg_idx == offset + int(m[:large_leg_idx].sum())

    def matrix_exp(self, matrix: HipBlock) -> HipBlock:
        """scipy.linalg.expm (numpy.cpp:1227-1234) as scaling and squaring of a degree-18 Taylor polynomial in Horner
        form: every step is one grouped-GEMM launch.  ||A / 2^s||_1 <= 1/2 makes the truncation error < 2e-23."""
        if matrix.ndim != 2 or matrix.shape[0] != matrix.shape[1]:
            raise ValueError('matrix_exp: square 2-D block required')
        n = matrix.shape[0]
        if n == 0:
            return self.copy_block(matrix)
        absA = self.abs(matrix) if not matrix.is_complex else \
            self.sqrt(self.linear_combination(1.0, self.multiply_blocks(self.copy_block(self.real(matrix)), self.copy_block(self.real(matrix))),
                                              1.0, self.multiply_blocks(self.copy_block(self.imag(matrix)), self.copy_block(self.imag(matrix)))))
        norm1 = self.max(self.sum(absA, 0))
        s = 0 if norm1 <= 0.5 else int(math.ceil(math.log2(norm1 / 0.5)))
        A = self.mul(0.5 ** s, matrix)
        eye = self.eye_matrix(n) if not matrix.is_complex else self.as_complex(self.eye_matrix(n))
        P = eye
        for k in range(18, 0, -1):  # P = I + (A / k) P
            P = self.linear_combination(1.0, eye, 1.0 / k, self.matrix_dot(A, P))
        for _ in range(s):
            P = self.matrix_dot(P, P)
        return P

    def permute_combined_matrix(self, block: HipBlock, dims1, idcs1, dims2, idcs2) -> HipBlock:
        """block_backend.cpp:857-884."""
        b = self.reshape(block, list(dims1) + list(dims2))
        b = self.permute_axes(b, list(idcs1) + list(idcs2))
        M = math.prod(b.shape[:len(idcs1)])
        return self.reshape(b, (M, b.size // max(M, 1)))

    def permute_combined_idx(self, block: HipBlock, axis: int, dims, idcs) -> HipBlock:
        """block_backend.cpp:886-921."""
        if block.ndim != 2:
            raise RuntimeError('permute_combined_idx: block must be 2D')
        M, N = block.shape
        if axis in (-2, 0):
            b = self.reshape(block, list(dims) + [N])
            b = self.permute_axes(b, list(idcs) + [len(idcs)])
            return self.reshape(b, (M, N))
        if axis in (-1, 1):
            b = self.reshape(block, [M] + list(dims))
            b = self.permute_axes(b, [0] + [1 + i for i in idcs])
            return self.reshape(b, (M, N))
        raise ValueError('Invalid axis.')

    def tensor_outer(self, a: HipBlock, b: HipBlock, K: int) -> HipBlock:
        """block_backend.cpp:994-1010."""
        res = self.outer(a, b)
        N, M = a.ndim, b.ndim
        return self.permute_axes(res, list(range(K)) + [N + i for i in range(M)] + list(range(K, N)))

    def random_uniform(self, dims, dtype=None, device=None, seed=None) -> HipBlock:
        """Uniform on [-1, 1) (numpy.cpp:965-988), real and imaginary part independently for complex dtypes."""
        if seed is None:
            seed = int(np.random.default_rng().integers(0, 2 ** 63 - 1))
        cplx = dtype is not None and _norm_dtype(dtype).kind == 'c'
        blk = self._new(dims, cplx)
        n = blk.size * (2 if cplx else 1)
        self.ctx.sync_stream()
        _lib.check(self.lib.cyb_random_uniform_f64(self.ctx.handle, C.c_void_p(blk.ptr), n, int(seed), -1.0, 1.0))
        return blk

    def real_if_close(self, a: HipBlock, tol: float) -> HipBlock:
        """np.real_if_close (numpy.cpp:999-1006): the real part if every |imag| < tol * eps(float64)."""
        if not a.is_complex:
            return a
        if self.max_abs(self.copy_block(self.imag(a))) < tol * np.finfo(np.float64).eps:
            return self.copy_block(self.real(a))
        return a

    def _block_repr_lines(self, a: HipBlock, indent: str, max_width: int, max_lines: int):
        """numpy.cpp:1017-1055."""
        with np.printoptions(linewidth=max_width - len(indent)):
            lines = [f'{indent}{line}' for line in str(self.to_numpy(a)).split('\n')]
        if len(lines) > max_lines:
            first = (max_lines - 1) // 2
            last = max_lines - 1 - first
            lines = lines[:first] + [f'{indent}...'] + lines[-last:]
        return lines

    def block_from_hdf5(self, hdf5_loader, h5gr, subpath) -> HipBlock:
        """``Block::from_hdf5`` (numpy.cpp:285-293): load ``subpath + 'arr'`` through the caller's loader and upload it."""
        blk = self.block_from_numpy(np.asarray(hdf5_loader.load(subpath + 'arr')))
        if hasattr(hdf5_loader, 'memorize_load'):
            hdf5_loader.memorize_load(h5gr, blk)
        return blk



# ---------------------------------------------------------------------------------------------------------------------------
# dtype policy: compute in float64 / complex128, cast on store
# ---------------------------------------------------------------------------------------------------------------------------
# methods whose result keeps the int64 type when all block operands are int64 (numpy: movement and +, -, *, abs, sums,
# extrema of integers stay integers; division, roots, norms etc. give float64)
_INT_KEEPS = frozenset({'permute_axes', 'reshape', 'add_axis', 'squeeze_axes', 'get_item', 'copy_block', 'contiguous', 'contiguous_many',
                        'combine_legs', 'split_legs', 'apply_mask', 'enlarge_leg', 'enlarge_leg_many', 'mask_gather_many', 'tile',
                        'get_diagonal', 'block_from_diagonal', 'abs', 'sum', 'multiply_blocks', 'apply_leg_permutations',
                        'apply_basis_perm', 'permute_combined_matrix', 'permute_combined_idx', 'subblock', 'dagger', 'conj', 'real',
                        'outer', 'kron', 'tensor_outer'})
_INT_BINARY_OPS = (0, 1, 2)     # `_binary` op codes add / sub / mul
# methods the policy does not touch: they take no blocks, return no blocks, or set the dtype themselves
_POLICY_SKIP = frozenset({'to_dtype', 'to_numpy', 'as_scalar', 'block_from_numpy', 'as_block', 'get_dtype', 'get_shape', 'get_device', 'is_real',
                          'synchronize', 'test_block_sanity', 'is_correct_block_type', 'as_device', 'possible_svd_algorithms',
                          'get_backend_name', 'concatenate_to_numpy', 'item', 'get_block_element', 'block_from_hdf5', 'make_gemm_plan',
                          'truncate_select', 'argsort', 'abs_argmax', 'argmin', 'any', 'all', 'allclose', 'get_block_mask_element'})
_CREATE_WITH_DTYPE = frozenset({'zeros', 'zeros_many', 'ones_block', 'eye_matrix', 'eye_block', 'random_normal', 'random_uniform',
                                'block_from_mask'})


def _collect_blocks(obj, acc, depth=0):
    if isinstance(obj, HipBlock):
        acc.append(obj)
    elif isinstance(obj, Scalar):
        acc.append(obj._blk)
    elif isinstance(obj, (list, tuple)) and depth < 4:
        for x in obj:
            _collect_blocks(x, acc, depth + 1)


def _map_blocks(obj, fn, depth=0):
    if isinstance(obj, HipBlock):
        return fn(obj)
    if isinstance(obj, Scalar):
        return Scalar(fn(obj._blk))
    if isinstance(obj, list) and depth < 4:
        return [_map_blocks(x, fn, depth + 1) for x in obj]
    if isinstance(obj, tuple) and depth < 4:
        return tuple(_map_blocks(x, fn, depth + 1) for x in obj)
    return obj


def _with_dtype_policy(name, fn):
    import inspect
    params = list(inspect.signature(fn).parameters)
    dtype_pos = params.index('dtype') - 1 if 'dtype' in params else None      # position among the arguments after self

    @functools.wraps(fn)
    def wrapped(self, *args, **kw):
        want = None
        if name in _CREATE_WITH_DTYPE:                     # the caller names the dtype: storage type for the kernel, tag afterwards
            d = kw.get('dtype', args[dtype_pos] if dtype_pos is not None and dtype_pos < len(args) else None)
            if d is not None and _norm_dtype(d) in _NOMINAL:
                want = _norm_dtype(d)
                if 'dtype' in kw:
                    kw['dtype'] = _STORAGE_OF[want]
                else:
                    args = args[:dtype_pos] + (_STORAGE_OF[want],) + args[dtype_pos + 1:]
        if want is None and (self._n_tagged == 0 or self._policy_depth):
            return fn(self, *args, **kw)
        ins = []
        _collect_blocks(args, ins)
        _collect_blocks(list(kw.values()), ins)
        if want is None and all(getattr(b, '_nom', None) is None for b in ins):
            return fn(self, *args, **kw)
        self._policy_depth += 1
        try:
            out = fn(self, *args, **kw)
        finally:
            self._policy_depth -= 1
        if want is None:
            kinds = [b.dtype for b in ins]
            floats = [d for d in kinds if d.kind in 'fc']
            ints = [d for d in kinds if d.kind == 'i']
            if floats and not ints and all(d in _NOMINAL for d in floats):
                want = 'single'
            elif ints and not floats and (name in _INT_KEEPS or (name == '_binary' and args[-1] in _INT_BINARY_OPS)):
                want = np.dtype('int64')
            else:
                if name == 'set_item' and ins and getattr(ins[0], '_nom', None) is not None:
                    self._retag(ins[0], ins[0]._nom, True)          # numpy casts the value to the array's dtype on assignment
                return out
        if name == 'set_item':
            if ins and getattr(ins[0], '_nom', None) is not None:
                self._retag(ins[0], ins[0]._nom, True)
            return out
        in_bufs = {id(b.buf) for b in ins if getattr(b, '_nom', None) is not None}

        def tag(blk):
            if blk.is_bool:
                return blk
            nominal = want if want != 'single' else np.dtype('complex64' if blk.is_complex else 'float32')
            if nominal == np.dtype('int64') and blk.is_complex:
                return blk
            return self._retag(blk, nominal, id(blk.buf) not in in_bufs)   # views of tagged operands hold rounded values already
        return _map_blocks(out, tag)
    return wrapped


for _name, _fn in list(vars(HipBlockBackend).items()):
    if callable(_fn) and not isinstance(_fn, (staticmethod, classmethod, type)) and (
            (not _name.startswith('_') and _name not in _POLICY_SKIP) or _name in ('_binary', '_pow', '_compare')):
        setattr(HipBlockBackend, _name, _with_dtype_policy(_name, _fn))
del _name, _fn
