"""Mechanical checks of the cyten integration surface that need neither cyten nor a GPU:

* every pure virtual (`= 0`) of the reference's ``BlockBackend`` / ``BlockBackend::Block`` (parsed from the TEXT of
  include/cyten/block_backend/block_backend.h when /root/reference exists) has a same-named counterpart with a compatible
  number of parameters on ``HipBlockBackend`` / ``HipBlock`` -- the mirror a C++ subclass forwards to;
* every attribute of the Array-API namespace and of its array objects that the reference's ``ArrayApiBlockBackend`` uses
  (parsed from src/block_backend/array_api.cpp) exists on ``HipArrayNamespace`` / ``HipArray`` -- the route by which an
  UNMODIFIED cyten runs on the device (integration/cyten_hip.py);
* the operations the reference base leaves NotImplemented are exactly the ones the registration subclass overrides.

Skipped where the reference tree is absent (the GPU box)."""
import inspect
import os
import re

import pytest

from cyten_amd.block_backend import HipBlock, HipBlockBackend, Scalar
from integration import cyten_hip
from integration.hip_array_api import HipArray, HipArrayNamespace

REF = '/root/reference'
need_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference sources not present')

# C++ operator -> the Python special method that carries it on the mirror
OPERATORS = {'operator+': '__add__', 'operator-': '__sub__', 'operator*': '__mul__', 'operator/': '__truediv__',
             'operator<': '__lt__', 'operator<=': '__le__', 'operator>': '__gt__', 'operator>=': '__ge__',
             'operator==': '__eq__', 'operator!=': '__ne__'}
# reference virtuals whose counterpart carries another name on the host mirror (with the reason)
RENAMED = {'get_backend': 'get_backend', '_item_as_complex128': None, '_item_as_int64': None}   # host accessors live on Scalar


def _pure_virtuals(text):
    """[(name, n_params)] of the `virtual ... name(args) ... = 0;` declarations in a class body"""
    out = []
    for m in re.finditer(r'virtual\s+[^;{}]*?\b(operator\s*[^\s(]+|[A-Za-z_]\w*)\s*\(([^;{}]*?)\)\s*(?:const\s*)?=\s*0\s*;', text, re.S):
        name = re.sub(r'\s+', '', m.group(1))
        args = m.group(2).strip()
        depth, n = 0, (1 if args else 0)
        for ch in args:
            depth += ch in '<(' 
            depth -= ch in '>)'
            n += (ch == ',' and depth == 0)
        out.append((name, n))
    return out


def _accepts(fn, n):
    """can `fn` (a method) be called with n positional arguments besides self?"""
    sig = inspect.signature(fn)
    params = [p for p in sig.parameters.values() if p.name != 'self']
    if any(p.kind == p.VAR_POSITIONAL for p in params):
        return True
    pos = [p for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    required = [p for p in pos if p.default is p.empty]
    return len(required) <= n <= len(pos)


@need_ref
def test_every_pure_virtual_of_the_reference_header_has_a_counterpart():
    text = open(os.path.join(REF, 'include/cyten/block_backend/block_backend.h')).read()
    i_block = text.index('class Block : public std::enable_shared_from_this<Block>')
    i_scalar = text.index('/// Holds a single scalar value', i_block)
    block_virtuals = _pure_virtuals(text[i_block:i_scalar])
    backend_virtuals = _pure_virtuals(text[i_scalar:])
    assert len(block_virtuals) >= 20 and len(backend_virtuals) >= 70          # the parser found the interface
    missing = []
    for name, n in block_virtuals:
        if name in OPERATORS:
            target = getattr(HipBlock, OPERATORS[name], None)
        elif name in ('_item_as_complex128', '_item_as_int64'):
            target = getattr(Scalar, {'_item_as_complex128': 'as_complex128', '_item_as_int64': 'as_int64'}[name])
            n = 0
        elif name in ('get_item', 'set_item'):
            target = getattr(HipBlockBackend, name)                            # (block, key[, value]) on the backend mirror
            n += 1
        else:
            target = getattr(HipBlock, name, None)
        if target is None:
            missing.append(('Block', name))
        elif callable(target) and not isinstance(target, property) and not _accepts(target, n):
            missing.append(('Block', name, f'arity {n}'))
    for name, n in backend_virtuals:
        target = getattr(HipBlockBackend, name, None)
        if target is None:
            missing.append(('BlockBackend', name))
        elif name != 'as_scalar' and not _accepts(target, n):                  # as_scalar: nine C++ overloads, one Python method
            missing.append(('BlockBackend', name, f'arity {n}'))
    assert not missing, missing


@need_ref
def test_array_api_namespace_provides_what_the_reference_backend_calls():
    src = open(os.path.join(REF, 'src/block_backend/array_api.cpp')).read()
    api_names = set(re.findall(r'api_\.attr\("(\w+)"\)(?!\.attr)', src)) | set(re.findall(r'api\(\)\.attr\("(\w+)"\)', src))
    linalg_names = set(re.findall(r'api_\.attr\("linalg"\)\.attr\("(\w+)"\)', src))
    array_names = set(re.findall(r'(?:arr_|obj\([a-z_]*\)|\ba|\bb)\.attr\("(\w+)"\)', src))
    assert {'matmul', 'tensordot', 'asarray', 'zeros', 'permute_dims', 'reshape'} <= api_names and {'svd', 'qr', 'eigh'} <= linalg_names
    lacking = sorted(n for n in api_names - {'linalg'} if not hasattr(HipArrayNamespace, n))
    lacking += sorted('linalg.' + n for n in linalg_names if not hasattr(__import__('integration.hip_array_api', fromlist=['_Linalg'])._Linalg, n))
    lacking += sorted('array.' + n for n in array_names if not hasattr(HipArray, n))
    assert not lacking, lacking


@need_ref
def test_cold_overrides_are_exactly_what_the_reference_base_leaves_open():
    src = open(os.path.join(REF, 'src/block_backend/array_api.cpp')).read()
    left_open = set(re.findall(r'NotImplemented\("ArrayApiBlockBackend does not support (\w+)\.', src)) | {'tile'}
    assert left_open == set(cyten_hip.COLD_OVERRIDES)
    code = inspect.getsource(cyten_hip._make_backend_class)
    for name in cyten_hip.COLD_OVERRIDES:       # override name == method name (the trampoline looks it up by the C++ name)
        assert re.search(rf'def {name}\(self', code), name


def test_registration_is_lazy_and_documented():
    assert 'pytest.param' in cyten_hip.CONFTEST_SNIPPET and '_block_backend_params' in cyten_hip.CONFTEST_SNIPPET
    src = inspect.getsource(cyten_hip)
    assert '_tensor_backend_cache' in src and 'import cyten' in inspect.getsource(cyten_hip.register)
    assert not re.search(r'^\s*(import|from)\s+cyten\b', src.split('def _make_backend_class')[0], re.M)   # no top-level import
