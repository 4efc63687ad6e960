"""Parity of the grouped fp64 MFMA GEMM (through the C-ABI) with the oracle's np.dot loop."""
import numpy as np
import pytest

from oracle import block_ops as ops

pytestmark = pytest.mark.gpu

TOL = 1e-10  # BASELINE.json: fp64 within 1e-10 of the numpy backend on the same inputs


def _run_groups(bb, groups_np, views=None):
    groups = []
    for g in groups_np:
        groups.append([(bb.as_block(a), bb.as_block(b)) for a, b in g])
    outs = bb.matrix_dot_grouped(groups)
    return [bb.to_numpy(o) for o in outs]


def _ref_groups(groups_np):
    res = []
    for g in groups_np:
        acc = ops.matrix_dot(*g[0])
        for a, b in g[1:]:
            acc = acc + ops.matrix_dot(a, b)
        res.append(acc)
    return res


def _check(outs, refs):
    for o, r in zip(outs, refs):
        assert o.shape == r.shape
        scale = max(1.0, np.abs(r).max())
        assert np.abs(o - r).max() <= TOL * scale


@pytest.mark.parametrize('shape', [(1, 1, 1), (16, 16, 4), (17, 5, 3), (33, 47, 29), (64, 64, 64), (100, 90, 77),
                                   (128, 128, 16), (129, 127, 17), (212, 212, 180), (300, 1, 50), (1, 300, 50),
                                   (257, 130, 65), (474, 474, 474),
                                   # strip classes (one extent below 40, the other long) and ragged edges
                                   (5, 3000, 5), (3000, 5, 5), (30, 2500, 20), (2500, 30, 20), (19, 256, 70), (300, 39, 33),
                                   (263, 307, 90), (130, 193, 48), (144, 160, 31), (385, 129, 17), (96, 600, 64)])
def test_single_gemm_shapes(bb, rng, shape):
    M, N, K = shape
    g = [[(rng.standard_normal((M, K)), rng.standard_normal((K, N)))]]
    _check(_run_groups(bb, g), _ref_groups(g))


def test_ragged_block_list_with_ksplit_accumulation(bb, rng):
    """Many problems of all tile classes in ONE call, several with K-split pairs (the reference's
    `block = block + matrix_dot(...)` chain, abelian.cpp:1437-1446)."""
    groups = []
    for _ in range(60):
        M, N = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        nseg = int(rng.integers(1, 5))
        groups.append([(rng.standard_normal((M, k)), rng.standard_normal((k, N)))
                       for k in rng.integers(1, 90, size=nseg)])
    _check(_run_groups(bb, groups), _ref_groups(groups))


def test_strided_operand_views(bb, rng):
    """Transposed / permuted operands are read in place (no copy) when they have a unit stride."""
    a = rng.standard_normal((70, 45))
    b = rng.standard_normal((45, 91))
    A_t = bb.permute_axes(bb.as_block(np.ascontiguousarray(a.T)), [1, 0])   # column-major view of a
    B_t = bb.permute_axes(bb.as_block(np.ascontiguousarray(b.T)), [1, 0])
    for x, y in [(A_t, bb.as_block(b)), (bb.as_block(a), B_t), (A_t, B_t)]:
        out = bb.to_numpy(bb.matrix_dot(x, y))
        assert np.abs(out - a @ b).max() <= TOL * np.abs(a @ b).max()
    # a row-sliced operand (offset + larger leading dimension) and an output written into a slice
    big = bb.as_block(rng.standard_normal((100, 60)))
    sub = bb.get_item(big, (slice(10, 80), slice(5, 50)))
    ref = bb.to_numpy(big)[10:80, 5:50] @ b
    assert np.abs(bb.to_numpy(bb.matrix_dot(sub, bb.as_block(b))) - ref).max() <= TOL * np.abs(ref).max()
    # a view with no unit stride at all falls back to one strided copy, still correct
    t3 = bb.as_block(rng.standard_normal((6, 7, 8)))
    v = bb.reshape(bb.permute_axes(t3, [1, 0, 2]), (7, 48))
    ref = bb.to_numpy(t3).transpose(1, 0, 2).reshape(7, 48) @ np.ones((48, 3))
    assert np.abs(bb.to_numpy(bb.matrix_dot(v, bb.ones_block((48, 3)))) - ref).max() <= TOL * np.abs(ref).max()


def test_np_dot_vector_forms_and_tdot(bb, rng):
    a, v, w = rng.standard_normal((9, 13)), rng.standard_normal(13), rng.standard_normal(9)
    np.testing.assert_allclose(bb.to_numpy(bb.matrix_dot(bb.as_block(a), bb.as_block(v))), a @ v, atol=1e-12)
    np.testing.assert_allclose(bb.to_numpy(bb.matrix_dot(bb.as_block(w), bb.as_block(a))), w @ a, atol=1e-12)
    np.testing.assert_allclose(bb.to_numpy(bb.matrix_dot(bb.as_block(v), bb.as_block(v))), v @ v, atol=1e-12)
    x, y = rng.standard_normal((4, 5, 6, 3)), rng.standard_normal((6, 2, 5))
    out = bb.tdot(bb.as_block(x), bb.as_block(y), [1, 2], [2, 0])
    np.testing.assert_allclose(bb.to_numpy(out), ops.tdot(x, y, [1, 2], [2, 0]), atol=1e-12)
    assert bb.tdot(bb.as_block(x), bb.as_block(y), [], []).shape == (4, 5, 6, 3, 6, 2, 5)


def test_empty_and_invalid_inputs(bb, rng):
    assert bb.matrix_dot_grouped([]) == []
    z = bb.tdot(bb.zeros((0, 4)), bb.zeros((4, 3)), [1], [0])
    assert z.shape == (0, 3)
    with pytest.raises(ValueError):
        bb.matrix_dot(bb.as_block(rng.standard_normal((3, 4))), bb.as_block(rng.standard_normal((5, 2))))
    with pytest.raises(ValueError):
        bb.tdot(bb.as_block(rng.standard_normal((3, 4))), bb.as_block(rng.standard_normal((5, 2))), [1], [0])


def test_full_size_theta_gemm_properties(bb):
    """chi = 4096 U(1) theta (BASELINE headline size): too big for an element-wise CPU check of
    every block inside the time budget, so use size-independent properties: (i) linearity in a
    random probe, C x == A (B x), evaluated through the 16-wide tile class (an independent code
    path from the 128-wide tiles that produced C); (ii) the largest block element-wise on the host."""
    from cyten_amd import abelian as ab, workloads as wl
    from helpers import to_device_tensor
    A, B = wl.config_u1_mps(4096)
    a, b = to_device_tensor(bb, A), to_device_tensor(bb, B)
    plan = ab.compose_plan(a, b, 1)
    assert len(plan.pairs) == 54 and abs(plan.flops - 10.80e9) < 0.05e9   # 54 GEMMs, 10.80 GFLOP
    theta = ab.compose(bb, a, b, 1)
    rng = np.random.default_rng(1)
    big = int(np.argmax([np.prod(s) for s in plan.res_shapes]))
    for g in {0, len(plan.pairs) // 2, big}:
        C = bb.reshape(theta.blocks[g], (plan.res_shapes[g][0] * plan.res_shapes[g][1], -1))
        x = bb.as_block(rng.standard_normal((C.shape[1], 1)))
        lhs = bb.to_numpy(bb.matrix_dot(C, x))
        rhs = np.zeros_like(lhs)
        for i, j in plan.pairs[g]:
            a2 = bb.reshape(a.blocks[i], (C.shape[0], -1))
            b2 = bb.reshape(b.blocks[j], (a2.shape[1], -1))
            rhs += bb.to_numpy(bb.matrix_dot(a2, bb.matrix_dot(b2, x)))
        assert np.abs(lhs - rhs).max() <= 1e-9 * np.abs(rhs).max()
    i, j = plan.pairs[big][0]
    ref = sum(A.blocks[i].reshape(-1, A.blocks[i].shape[-1]) @ B.blocks[j].reshape(B.blocks[j].shape[0], -1)
              for i, j in plan.pairs[big])
    got = bb.to_numpy(theta.blocks[big]).reshape(ref.shape)
    assert np.abs(got - ref).max() <= TOL * np.abs(ref).max()


@pytest.mark.parametrize('M,Ks,N', [(10, [10], 8192), (10, [10], 9001), (16, [32], 8193), (1, [1], 8192), (5, [3, 10, 7], 10007),
                                    (16, [10, 10], 40960 + 5), (7, [32, 1], 8200), (10, [10], 8191)])
def test_streaming_class_for_skinny_products(bb, rng, M, Ks, N):
    """The HBM-bound products of an MPO-sized operator with a long operand (tile class 10 of gemm_grouped.hip):
    small M and K segments, N in the thousands to millions, ragged last tile, odd N, several K segments, operands as
    sub-views with odd row strides, mixed in one launch with ordinary problems."""
    groups, views = [], []
    for K in Ks:
        a = rng.standard_normal((M, K + 3))[:, 1:K + 1]                 # column-offset view of A
        b = rng.standard_normal((K + 1, N + 5))[1:, 3:N + 3]            # B rows start at odd element offsets
        groups.append((a, b))
    other = [(rng.standard_normal((70, 40)), rng.standard_normal((40, 90)))]
    dev_groups = [[(bb.get_item(bb.as_block(np.pad(a, ((0, 0), (1, 2)))), (slice(None), slice(1, 1 + a.shape[1]))),
                    bb.get_item(bb.as_block(np.pad(b, ((1, 0), (3, 2)))), (slice(1, None), slice(3, 3 + N)))) for a, b in groups],
                  [(bb.as_block(other[0][0]), bb.as_block(other[0][1]))]]
    outs = bb.matrix_dot_grouped(dev_groups)
    want = sum(a @ b for a, b in groups)
    got = bb.to_numpy(outs[0])
    assert got.shape == (M, N)
    assert np.abs(got - want).max() <= TOL * max(1.0, np.abs(want).max())
    assert np.abs(bb.to_numpy(outs[1]) - other[0][0] @ other[0][1]).max() <= TOL * 100
