"""Randomised soak of the round-3 surfaces on the device: the remaining AbelianBackend callers (partial_compose, mask_contract,
two-leg qr / lq, weighted truncation), the FusionTree callers and tree moves, complex linear combinations and the six-dtype
policy.  It re-runs the GPU tests of those surfaces as plain functions with OTHER seeds: the case generators of
tests/abelian_caller_cases.py are shifted (`SEED_SHIFT`), every `rng` fixture is a fresh generator, so the same assertions
(against oracle/abelian_ref.py, oracle/fusion_tree_ref.py, oracle/block_ops.py, numpy in the nominal dtype, the reference-held
tree-move expectations) meet other charges, multiplicities, fill patterns and data.
    python scripts/callers_fuzz.py [n_rounds=20] [seed=0]"""
import itertools
import sys
import time
import traceback

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np

import abelian_caller_cases as cases
import test_gpu_abelian_callers as t_ab
import test_gpu_dtypes as t_dt
import test_gpu_fusion_tree as t_ft
import test_gpu_tree_moves as t_tm
from cyten_amd.block_backend import HipBlockBackend

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bb = HipBlockBackend('cuda:0')


def calls(r):
    """(name, thunk) for one round; `rng()` hands every call its own generator"""
    k = [0]

    def rng():
        k[0] += 1
        return np.random.default_rng([seed, r, k[0]])
    out = []
    for case in range(9):
        out.append((f'partial_compose[{case}]', lambda case=case: t_ab.test_partial_compose(bb, case)))
    for leg_idx, cplx in itertools.product([0, 2, 4], [False, True]):
        out.append((f'mask_contract[{leg_idx},{cplx}]',
                    lambda leg_idx=leg_idx, cplx=cplx: t_ab.test_mask_contract_is_one_batched_gather_or_scatter(bb, rng(), leg_idx, cplx)))
    out.append(('enlarge_leg_many', lambda: t_ab.test_enlarge_leg_many_matches_numpy(bb, rng())))
    for case, lq in itertools.product(range(3), [False, True]):
        out.append((f'two_leg_qr_lq[{case},{lq}]', lambda case=case, lq=lq: t_ab.test_two_leg_qr_lq_with_identity_blocks_for_absent_sectors(bb, case, lq)))
    out.append(('weighted_truncation', lambda: t_ab.test_weighted_truncation_on_the_device(bb, rng())))
    for fn in (t_dt.test_to_dtype_accepts_the_six_dtypes_and_matches_numpy, t_dt.test_blocks_from_single_precision_arrays_keep_their_dtype,
               t_dt.test_promotion_follows_numpy, t_dt.test_views_of_single_precision_blocks_keep_dtype_and_memory,
               t_dt.test_hot_path_in_single_precision, t_dt.test_assignment_into_a_single_precision_block_rounds_like_numpy,
               t_dt.test_float64_path_is_untouched_by_the_policy):
        out.append((fn.__name__[5:], lambda fn=fn: fn(bb, rng())))
    for case in t_ft.CASES:
        out.append((f"ft_transform[{case['name']}]",
                    lambda case=case: t_ft.test_device_transform_tensor_reproduces_the_reference_held_tree_moves(bb, case, rng())))
    for (pc, pd), cplx in itertools.product([((1, 0, 2), (0, 1)), ((0, 2, 1), (1, 0)), ((2, 0, 1), (1, 0))], [False, True]):
        out.append((f'ft_abelian_perm[{pc},{pd},{cplx}]',
                    lambda pc=pc, pd=pd, cplx=cplx: t_ft.test_device_transform_tensor_is_the_dense_leg_permutation_for_abelian_trees(bb, rng(), pc, pd, cplx)))
    out.append(('ft_compose_decomp', lambda: t_ft.test_device_compose_and_decompositions(bb, rng())))
    out.append(('ft_truncation_qdims', lambda: t_ft.test_device_truncation_with_quantum_dimensions(bb, rng())))
    for case in t_tm.CASES:
        out.append((f"tree_move[{case['name']}]", lambda case=case: t_tm.test_device_transform_blocks_reproduces_the_reference_expectation(bb, case, rng())))
    out.append(('lincomb_complex_views', lambda: t_tm.test_device_lincomb_complex_views(bb, rng())))
    return out


t0 = time.time()
n_calls, fails = 0, []
for r in range(n_rounds):
    cases.SEED_SHIFT = 7919 * seed + 101 * r + 1
    for name, thunk in calls(r):
        n_calls += 1
        try:
            thunk()
        except KeyboardInterrupt:
            raise
        except BaseException as exc:                  # noqa: BLE001  (a soak records and goes on; pytest's Failed is a BaseException)
            fails.append((r, name, repr(exc)[:300]))
            print(f'[callers-fuzz] FAIL round {r} seed {seed} shift {cases.SEED_SHIFT}: {name}: {exc!r}'[:600], flush=True)
            traceback.print_exc(limit=4)
    if r % 5 == 4:
        print(f'[callers-fuzz] round {r + 1}/{n_rounds}: {n_calls} checks, {len(fails)} failures, {time.time() - t0:.0f} s', flush=True)
print(f'[callers-fuzz] seed {seed}: {n_rounds} rounds, {n_calls} checks, {len(fails)} failures in {time.time() - t0:.0f} s')
for f in fails[:40]:
    print('   ', f)
sys.exit(1 if fails else 0)
