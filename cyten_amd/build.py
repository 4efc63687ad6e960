"""Build recipe for libcyten_amd.so (hand-written HIP for gfx950) and the oracle's C pieces.

``python -m cyten_amd.build`` compiles every ``csrc/*.hip`` translation unit with hipcc for
``--offload-arch=gfx950`` and links them into ``cyten_amd/lib/libcyten_amd.so`` *in-tree* (the
.so travels to the GPU box with the repo snapshot; it is git-ignored).  hipcc cross-compiles
without a GPU, so this also is the "does it build" check of ``__graft_entry__.build()``.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / 'csrc'
LIB_DIR = PKG_DIR / 'lib'
OBJ_DIR = PKG_DIR / 'lib' / 'obj'
LIB_PATH = LIB_DIR / 'libcyten_amd.so'
INCLUDE = PKG_DIR.parent / 'include'

ARCH = 'gfx950'
HIPCC_FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function',
               '-ffp-contract=fast']


def _hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: cannot build libcyten_amd.so')
    return exe


def _digest(paths) -> str:
    h = hashlib.sha256()
    h.update(' '.join(HIPCC_FLAGS).encode())
    for p in sorted(paths):
        h.update(str(p.name).encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def sources():
    return sorted(CSRC.glob('*.hip'))


def headers():
    return sorted(CSRC.glob('*.h')) + sorted(INCLUDE.glob('*.h'))


def build(force: bool = False, verbose: bool = True) -> Path:
    """Compile all HIP sources for gfx950 and link the shared library. Returns its path."""
    LIB_DIR.mkdir(exist_ok=True)
    OBJ_DIR.mkdir(exist_ok=True)
    srcs = sources()
    hdrs = headers()
    stamp = LIB_DIR / 'build.sha256'
    digest = _digest(srcs + hdrs)
    if not force and LIB_PATH.exists() and stamp.exists() and stamp.read_text() == digest:
        if verbose:
            print(f'[cyten_amd.build] up to date: {LIB_PATH}')
        return LIB_PATH
    hipcc = _hipcc()
    hdr_digest = _digest(hdrs)

    def compile_one(src: Path) -> Path:
        obj = OBJ_DIR / (src.stem + '.o')
        ostamp = OBJ_DIR / (src.stem + '.sha256')
        d = _digest([src]) + hdr_digest
        if not force and obj.exists() and ostamp.exists() and ostamp.read_text() == d:
            return obj
        cmd = [hipcc, *HIPCC_FLAGS, '-c', str(src), '-o', str(obj)]
        if verbose:
            print('[cyten_amd.build]', ' '.join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src.name}:\n{res.stdout}\n{res.stderr}')
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
        ostamp.write_text(d)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(srcs)))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', str(LIB_PATH), *map(str, objs)]
    if verbose:
        print('[cyten_amd.build]', ' '.join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'link failed:\n{res.stdout}\n{res.stderr}')
    stamp.write_text(digest)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB_PATH)
