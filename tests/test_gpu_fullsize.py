"""BASELINE-size checks of the decomposition path through size-independent properties, evaluated ON the
device (the oracle's LAPACK loop needs seconds per block at these sizes): chi = 4096 U(1) theta, all 15
coupled-charge blocks up to 1442 x 1442, fp64 tolerance 1e-10 relative to the block norm."""
import numpy as np
import pytest

from cyten_amd import abelian as ab
from cyten_amd import workloads as wl
from helpers import to_device_tensor

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_chi4096_theta_svd_properties(bb):
    A, B = wl.config_u1_mps(4096)
    a, b = to_device_tensor(bb, A), to_device_tensor(bb, B)
    theta = ab.compose(bb, a, b, 1)
    mv = ab.combine_legs_to_matrix(bb, theta, 2)
    shapes = [tuple(m.shape) for m in mv.blocks]
    assert len(shapes) == 15 and max(max(s) for s in shapes) == 1442
    U, S, Vh = ab.svd(bb, mv)
    # (i) reconstruction  U diag(S) Vh == M, one grouped GEMM for all sectors
    US = bb.scale_axis_many([(u, s, 1) for u, s in zip(U, S)])
    rec = bb.matrix_dot_grouped([[(us, vh)] for us, vh in zip(US, Vh)])
    diff = bb.linear_combination_many(1.0, rec, -1.0, mv.blocks)
    total2 = 0.0
    for d, m, s in zip(diff, mv.blocks, S):
        nrm = bb.norm(m)
        assert bb.max_abs(d) <= TOL * nrm
        # (ii) sum of squared singular values == squared Frobenius norm
        s_np = bb.to_numpy(s)
        assert abs(np.sum(s_np ** 2) - nrm ** 2) <= TOL * nrm ** 2
        # (iii) descending, non-negative
        assert np.all(s_np >= 0) and np.all(np.diff(s_np) <= 1e-12 * s_np[0])
        total2 += nrm ** 2
    # (iv) isometries: U^T U = I, Vh Vh^T = I (grouped GEMMs on transposed views)
    gram_u = bb.matrix_dot_grouped([[(bb.permute_axes(u, [1, 0]), u)] for u in U])
    gram_v = bb.matrix_dot_grouped([[(vh, bb.permute_axes(vh, [1, 0]))] for vh in Vh])
    for g in gram_u + gram_v:
        eye = bb.eye_matrix(g.shape[0])
        assert bb.max_abs(bb.linear_combination(1.0, g, -1.0, eye)) <= TOL
    # (v) every block of theta = A.B factors through ONE sector of the shared bond: its numerical rank is
    #     that sector's multiplicity (random blocks: exactly), about half the block size
    bond_mults = set(int(m) for m in wl.u1_leg(4096, 2.0).mults)
    for s, shp in zip(S, shapes):
        s_np = bb.to_numpy(s)
        rank = int(np.sum(s_np > 1e-9 * s_np[0]))
        assert rank in bond_mults and rank <= min(shp)
        assert np.all(s_np[rank:] <= 1e-10 * s_np[0])
    # (vi) norm bookkeeping of the truncation at chi_max = 4096
    _, Ut, St, Vt, err, new_norm = ab.truncated_svd(bb, theta, 2, chi_max=4096)
    assert sum(s.size for s in St) == 4096
    assert abs(err + new_norm - total2) <= TOL * total2
