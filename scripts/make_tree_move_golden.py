"""Writes tests/golden/ref_tree_move_cases.json: the literal expectations the reference's own tests hold for single tree
moves of the FusionTreeBackend -- C symbols (braids) and B symbols (bends) -- transcribed as DATA.

Provenance: /root/reference/tests/python_tests/backends/test_fusion_tree_backend.py
  * test_c_symbol_fibonacci_anyons  :36-188   (Fibonacci anyons, all multiplicities 1: every tree is one row / column)
  * test_c_symbol_su3_3             :401-617  (SU(3)_3: domain legs of multiplicity 2 -> column tree blocks of width 4;
                                               the rows / columns whose expectation the reference computes through
                                               `sym._c_symbol` instead of writing it out (:499-511, :575-581) are left out:
                                               those cases are marked "partial")
  * test_b_symbol_fibonacci_anyons  :634-786  (bends: entries move between coupled-sector blocks)
Each statement says `new[nb][dst] = sum_t coeff_t * old[ob_t][src_t]` with index lists along ONE axis (the other axis
taken whole) or single elements -- what the reference writes as `expect[..][rows, :] = blocks[..][rows', :] * symbol`.
No reference source is executed; the symbol values are the closed forms the reference test states (:59-73, :424, :488-489).
The tree-block structure a case needs to be replayed through `transform_blocks` (which rows / columns form one tree
block, multiplicity axes and their permutation) is recorded next to the statements.

    python scripts/make_tree_move_golden.py
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

phi = (1 + 5 ** 0.5) / 2
R_1 = np.exp(-4j * np.pi / 5)
R_tau = np.exp(3j * np.pi / 5)
SYM = {
    '1': 1.0, '-1': -1.0,
    'R_1': R_1, 'R_tau': R_tau,
    'C_tttt11': phi ** -1 * R_1.conjugate(),
    'C_ttttt1': phi ** -0.5 * R_tau * R_1.conjugate(),
    'C_tttt1t': phi ** -0.5 * R_tau.conjugate(),
    'C_tttttt': -1 * phi ** -1,
    'sqrt_phi': phi ** 0.5, 'inv_sqrt_phi': phi ** -0.5,
    'r8_0': -1j, 'r8_1': 1j,
}
f2 = np.array([[-0.5, -(3 ** 0.5) / 2], [3 ** 0.5 / 2, -0.5]])
f1 = f2.T
r8 = [-1j, 1j]
# composite coefficients of the SU(3)_3 case (two F moves around an R move), named by the row / column they feed
SYM.update({
    'f2f2_00': f2[0, 0] * f2[0, 0] + f2[0, 1] * f2[1, 0], 'f2f2_10': f2[1, 0] * f2[0, 0] + f2[1, 1] * f2[1, 0],
    'f2f2_01': f2[0, 0] * f2[0, 1] + f2[0, 1] * f2[1, 1], 'f2f2_11': f2[1, 0] * f2[0, 1] + f2[1, 1] * f2[1, 1],
    'f1f1_00': f1[0, 0] * f1[0, 0] + f1[0, 1] * f1[1, 0], 'f1f1_10': f1[1, 0] * f1[0, 0] + f1[1, 1] * f1[1, 0],
    'f1f1_01': f1[0, 0] * f1[0, 1] + f1[0, 1] * f1[1, 1], 'f1f1_11': f1[1, 0] * f1[0, 1] + f1[1, 1] * f1[1, 1],
    'f1rf2_00': f1[0, 0] * r8[0] * f2[0, 0] + f1[0, 1] * r8[1] * f2[1, 0], 'f1rf2_10': f1[1, 0] * r8[0] * f2[0, 0] + f1[1, 1] * r8[1] * f2[1, 0],
    'f1rf2_01': f1[0, 0] * r8[0] * f2[0, 1] + f1[0, 1] * r8[1] * f2[1, 1], 'f1rf2_11': f1[1, 0] * r8[0] * f2[0, 1] + f1[1, 1] * r8[1] * f2[1, 1],
    'f2rf1_00': f2[0, 0] * r8[0] * f1[0, 0] + f2[0, 1] * r8[1] * f1[1, 0], 'f2rf1_10': f2[1, 0] * r8[0] * f1[0, 0] + f2[1, 1] * r8[1] * f1[1, 0],
    'f2rf1_01': f2[0, 0] * r8[0] * f1[0, 1] + f2[0, 1] * r8[1] * f1[1, 1], 'f2rf1_11': f2[1, 0] * r8[0] * f1[0, 1] + f2[1, 1] * r8[1] * f1[1, 1],
})


def S(nb, dst, *terms):
    """one statement along the case's axis: new[nb][dst] = sum coeff * old[ob][src]; a term is (symbol, src) or (symbol, ob, src)"""
    out = []
    for t in terms:
        sym, ob, src = (t[0], nb, t[1]) if len(t) == 2 else t
        out.append({'coeff': sym, 'ob': ob, 'src': list(src)})
    return {'nb': nb, 'dst': list(dst), 'terms': out}


def scale(nb, idx, sym):
    return S(nb, idx, (sym, idx))


def rng_(a, b):
    return list(range(a, b))


exc = [0, 2, 1, 3]


def ex(base):
    return [base + i for i in exc]


CASES = []

# ---- Fibonacci anyons, C symbols (:36-188).  Blocks (8, 3) and (13, 5); all multiplicities 1.
fib = {'old_shapes': [[8, 3], [13, 5]], 'new_shapes': [[8, 3], [13, 5]], 'row_tree': {'width': 1, 'dims': [1, 1, 1, 1]},
       'col_tree': {'width': 1, 'dims': [1, 1, 1]}}
CASES.append(dict(fib, name='fib_c_exchange_legs_0_1', source=':75-97', axis=0, statements=[
    scale(0, [0, 1, 2], '1'), scale(0, [3, 6], 'R_1'), scale(0, [4, 5, 7], 'R_tau'),
    scale(1, [0, 1, 2, 3, 4], '1'), scale(1, [6, 8, 10], 'R_1'), scale(1, [5, 7, 9, 11, 12], 'R_tau')]))
CASES.append(dict(fib, name='fib_c_exchange_legs_5_6', source=':99-120', axis=1, statements=[
    scale(0, [0], '1'), scale(0, [1], 'R_1'), scale(0, [2], 'R_tau'),
    scale(1, [0, 1], '1'), scale(1, [3], 'R_1'), scale(1, [2, 4], 'R_tau')]))
CASES.append(dict(fib, name='fib_c_exchange_legs_2_3', source=':122-160', axis=0, statements=[
    S(0, [0], ('1', [1])), S(0, [1], ('1', [0])), scale(0, [2], 'R_tau'), scale(0, [3], '1'), S(0, [4], ('1', [5])), S(0, [5], ('1', [4])),
    scale(0, [6], 'R_1'), scale(0, [7], 'R_tau'),
    scale(1, [0], '1'), S(1, [1], ('1', [2])), S(1, [2], ('1', [1])),
    S(1, [3], ('C_tttt11', [3]), ('C_ttttt1', [4])), S(1, [4], ('C_tttt1t', [3]), ('C_tttttt', [4])),
    scale(1, [5], '1'), S(1, [6], ('1', [8])), S(1, [7], ('1', [9])), S(1, [8], ('1', [6])), S(1, [9], ('1', [7])), scale(1, [10], 'R_tau'),
    S(1, [11], ('C_tttt11', [11]), ('C_ttttt1', [12])), S(1, [12], ('C_tttt1t', [11]), ('C_tttttt', [12]))]))
CASES.append(dict(fib, name='fib_c_exchange_legs_4_5', source=':162-188', axis=1, statements=[
    scale(0, [0], 'R_1'), scale(0, [1], '1'), scale(0, [2], 'R_tau'),
    scale(1, [0], '1'), scale(1, [1], 'R_tau'), scale(1, [2], '1'),
    S(1, [3], ('C_tttt11', [3]), ('C_ttttt1', [4])), S(1, [4], ('C_tttt1t', [3]), ('C_tttttt', [4]))]))

# ---- SU(3)_3, C symbols (:401-617).  Blocks (6, 12), (16, 36), (5, 12), (5, 12); codomain multiplicities 1, domain [1, 2, 2]:
#      a column tree block is 4 wide with multiplicity axes (1, 2, 2).
su3 = {'old_shapes': [[6, 12], [16, 36], [5, 12], [5, 12]], 'new_shapes': [[6, 12], [16, 36], [5, 12], [5, 12]],
       'row_tree': {'width': 1, 'dims': [1, 1, 1]}, 'col_tree': {'width': 4, 'dims': [1, 2, 2]}}
st = []
for i in (0, 2, 3):
    st += [scale(i, [0], 'r8_0'), scale(i, [1], 'r8_1'), scale(i, [2], '-1'), S(i, [3, 4], ('1', [4, 3]))]
st += [scale(0, [5], '1'), scale(1, [0, 5, 6], '-1'), scale(1, [1, 3, 7], 'r8_0'), scale(1, [2, 4, 8], 'r8_1'),
       S(1, rng_(9, 16), ('1', [12, 13, 14, 9, 10, 11, 15]))]
CASES.append(dict(su3, name='su3_3_c_exchange_legs_0_1', source=':421-451', axis=0, statements=st))
st = []
for i in (0, 2, 3):
    st += [scale(i, rng_(0, 4), 'r8_0'), scale(i, rng_(4, 8), 'r8_1'), scale(i, rng_(8, 12), '1')]
st += [scale(1, rng_(0, 4), '-1'), scale(1, rng_(4, 8), 'r8_0'), scale(1, rng_(8, 12), 'r8_1'), scale(1, rng_(12, 16), 'r8_0'),
       scale(1, rng_(16, 20), 'r8_1'), scale(1, rng_(20, 28), '-1'), scale(1, rng_(28, 36), '1')]
CASES.append(dict(su3, name='su3_3_c_exchange_legs_4_5', source=':453-481', axis=1, statements=st))
CASES.append(dict(su3, name='su3_3_c_exchange_legs_1_2', source=':483-563', axis=0, partial='rows 0-6 of block 1 are computed through sym._c_symbol in the reference (:499-511), not written out',
                  statements=[
    scale(0, [0], 'r8_0'), scale(0, [1], 'r8_1'), S(0, [2, 3], ('1', [3, 2])), scale(0, [4], '-1'), scale(0, [5], '1'),
    S(1, [7], ('f2f2_00', [9]), ('f2f2_10', [10])), S(1, [8], ('f2f2_01', [9]), ('f2f2_11', [10])),
    S(1, [9], ('f1f1_00', [7]), ('f1f1_10', [8])), S(1, [10], ('f1f1_01', [7]), ('f1f1_11', [8])),
    scale(1, [11], '1'),
    S(1, [12], ('f1rf2_00', [12]), ('f1rf2_10', [13])), S(1, [13], ('f1rf2_01', [12]), ('f1rf2_11', [13])),
    S(1, [14, 15], ('-1', [15, 14])),
    S(2, [0], ('f1rf2_00', [0]), ('f1rf2_10', [1])), S(2, [1], ('f1rf2_01', [0]), ('f1rf2_11', [1])), S(2, [2, 3], ('-1', [3, 2])), scale(2, [4], '-1'),
    S(3, [0], ('f2rf1_00', [0]), ('f2rf1_10', [1])), S(3, [1], ('f2rf1_01', [0]), ('f2rf1_11', [1])), S(3, [2, 3], ('-1', [3, 2])), scale(3, [4], '-1')]))
CASES.append(dict(su3, name='su3_3_c_exchange_legs_3_4', source=':565-617', axis=1, partial='columns 0-27 of block 1 are computed through sym._c_symbol in the reference (:575-581), not written out',
                  statements=[
    S(0, rng_(0, 4), ('r8_0', ex(0))), S(0, rng_(4, 8), ('r8_1', ex(4))), S(0, rng_(8, 12), ('-1', ex(8))),
    S(1, rng_(28, 32), ('f1rf2_00', ex(28)), ('f1rf2_10', ex(32))), S(1, rng_(32, 36), ('f1rf2_01', ex(28)), ('f1rf2_11', ex(32))),
    S(2, rng_(0, 4), ('f1rf2_00', ex(0)), ('f1rf2_10', ex(4))), S(2, rng_(4, 8), ('f1rf2_01', ex(0)), ('f1rf2_11', ex(4))), S(2, rng_(8, 12), ('-1', ex(8))),
    S(3, rng_(0, 4), ('f2rf1_00', ex(0)), ('f2rf1_10', ex(4))), S(3, rng_(4, 8), ('f2rf1_01', ex(0)), ('f2rf1_11', ex(4))), S(3, rng_(8, 12), ('-1', ex(8)))]))


# ---- Fibonacci anyons, B symbols (:634-786): single entries move between the blocks (all multiplicities 1)
def E(nb, r, c, sym, ob, rr, cc):
    return {'nb': nb, 'dst': [r, c], 'terms': [{'coeff': sym, 'ob': ob, 'src': [rr, cc]}]}


CASES.append({'name': 'fib_b_bend_up_single_domain_leg', 'source': ':666-689', 'axis': 'element', 'old_shapes': [[1, 2]], 'new_shapes': [[2, 1]],
              'statements': [E(0, 0, 0, '1', 0, 0, 0), E(0, 1, 0, '1', 0, 0, 1)]})
up = [E(0, 0, 0, '1', 0, 0, 1), E(0, 1, 0, 'sqrt_phi', 1, 0, 3), E(0, 2, 0, '1', 0, 1, 1), E(0, 3, 0, 'sqrt_phi', 1, 1, 3), E(0, 4, 0, 'sqrt_phi', 1, 2, 3)]
for col, (c_a, c_b) in enumerate([(0, 1), (2, 4)]):      # new column 0 reads old columns (0 | 1), new column 1 reads (2 | 4)
    up += [E(1, 0, col, '1', 1, 0, c_a), E(1, 1, col, 'inv_sqrt_phi', 0, 0, c_a), E(1, 2, col, '1', 1, 0, c_b), E(1, 3, col, '1', 1, 1, c_a),
           E(1, 4, col, '1', 1, 2, c_a), E(1, 5, col, '1', 1, 1, c_b), E(1, 6, col, 'inv_sqrt_phi', 0, 1, c_a), E(1, 7, col, '1', 1, 2, c_b)]
CASES.append({'name': 'fib_b_bend_up', 'source': ':691-744', 'axis': 'element', 'old_shapes': [[2, 3], [3, 5]], 'new_shapes': [[5, 1], [8, 2]],
              'statements': up})
down = [E(0, 0, c, 'sqrt_phi', 1, 1, c) for c in range(5)]
for row, src_row in enumerate([0, 2]):                     # new row 0 reads old rows 0, new row 1 reads old rows 2 (block 1) / 1 (block 0)
    r0 = row
    down += [E(1, row, 0, '1', 1, src_row, 0), E(1, row, 1, 'inv_sqrt_phi', 0, r0, 0), E(1, row, 2, '1', 1, src_row, 1),
             E(1, row, 3, 'inv_sqrt_phi', 0, r0, 1), E(1, row, 4, '1', 1, src_row, 2), E(1, row, 5, '1', 1, src_row, 3),
             E(1, row, 6, 'inv_sqrt_phi', 0, r0, 2), E(1, row, 7, '1', 1, src_row, 4)]
CASES.append({'name': 'fib_b_bend_down', 'source': ':746-786', 'axis': 'element', 'old_shapes': [[2, 3], [3, 5]], 'new_shapes': [[1, 5], [2, 8]],
              'statements': down})

# the leg permutation of each move as the reference passes it to permute_legs (codomain_idcs, domain_idcs over the flat legs =
# codomain + reversed domain), the leg counts, and the multiplicities of one tree's uncoupled sectors before / after the move
LEGS = {
    'fib_c_exchange_legs_0_1': ([1, 0, 2, 3], [6, 5, 4], ':96'), 'fib_c_exchange_legs_5_6': ([0, 1, 2, 3], [5, 6, 4], ':119'),
    'fib_c_exchange_legs_2_3': ([0, 1, 3, 2], [6, 5, 4], ':159'), 'fib_c_exchange_legs_4_5': ([0, 1, 2, 3], [6, 4, 5], ':187'),
    'su3_3_c_exchange_legs_0_1': ([1, 0, 2], [5, 4, 3], ':450'), 'su3_3_c_exchange_legs_4_5': ([0, 1, 2], [4, 5, 3], ':480'),
    'su3_3_c_exchange_legs_1_2': ([0, 2, 1], [5, 4, 3], ':562'), 'su3_3_c_exchange_legs_3_4': ([0, 1, 2], [5, 3, 4], ':616'),
    'fib_b_bend_up_single_domain_leg': ([0], [], ':688'), 'fib_b_bend_up': ([0, 1, 2, 3], [5, 4], ':743'), 'fib_b_bend_down': ([0, 1], [5, 4, 3, 2], ':785'),
}
OLD_JK = {'fib_c': (4, 3), 'su3_3': (3, 3), 'fib_b_bend_up_single_domain_leg': (0, 1), 'fib_b_bend_up': (3, 3), 'fib_b_bend_down': (3, 3)}
for c in CASES:
    cod, dom, where = LEGS[c['name']]
    J, K = OLD_JK.get(c['name']) or OLD_JK[c['name'][:5]]
    old_row = c.get('row_tree', {}).get('dims', [1] * J)
    old_col = c.get('col_tree', {}).get('dims', [1] * K)
    flat = list(old_row) + list(old_col)[::-1]                 # multiplicities over the flat legs
    c['legs'] = {'J': J, 'K': K, 'codomain_idcs': cod, 'domain_idcs': dom, 'permute_legs_call': where,
                 'old_row_dims': list(old_row), 'old_col_dims': list(old_col),
                 'new_row_dims': [flat[i] for i in cod], 'new_col_dims': [flat[i] for i in dom]}

symbols = {k: {'re': float(np.real(v)), 'im': float(np.imag(v))} for k, v in SYM.items()}
doc = {
    '_doc': 'Literal expectations of single tree moves held by the reference tests (tests/python_tests/backends/'
            'test_fusion_tree_backend.py), transcribed as data by scripts/make_tree_move_golden.py: new[nb][dst] = sum_t '
            'symbols[coeff_t] * old[ob_t][src_t]; axis 0: dst / src are row lists (all columns), axis 1: column lists (all rows), '
            '"element": [row, col] pairs.  Inputs are seeded random complex blocks of old_shapes, as in the reference '
            '(random_uniform, :51-53).  row_tree / col_tree: width of one tree block along that axis and its multiplicity axes.',
    'symbols': symbols,
    'cases': CASES,
}
path = os.path.join(ROOT, 'tests', 'golden', 'ref_tree_move_cases.json')
with open(path, 'w') as f:
    json.dump(doc, f, indent=1)
print(path, len(CASES), 'cases', sum(len(c['statements']) for c in CASES), 'statements')
