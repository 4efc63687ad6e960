"""numpy model of the block one-sided Jacobi iteration (csrc/jacobi_engine.hip): outer sweeps needed when the inner
eigensolve of the 32x32 Gram is (a) one full cyclic sweep (31 rounds, what the kernel does), (b) cross-block pairs only
(16 rounds), (c) cross-only except for the first visit of a block in every outer sweep.  Development aid."""
import sys
import numpy as np

JB = 16


def rot(a, d, b):
    if abs(b) < 1e-300:
        return 1.0, 0.0
    delta = d - a
    h = np.hypot(delta, 2 * b)
    c2 = 0.5 + 0.5 * abs(delta) / h
    c = np.sqrt(c2)
    s = (1.0 if (delta >= 0) == (b >= 0) else -1.0) * abs(b) / (h * c)
    return c, s


def circle(n, r, k):
    if k == 0:
        return n - 1, r
    m = n - 1
    return (r + k) % m, (r - k) % m


def inner(G, mode):
    """one inner sweep on the symmetric G (2JB x 2JB); returns Q with Q^T G Q closer to diagonal"""
    n = G.shape[0]
    Q = np.eye(n)
    G = G.copy()
    if mode == 'full':
        rounds = [[circle(n, r, k) for k in range(n // 2)] for r in range(n - 1)]
    else:  # cross pairs only: (i, JB + (i + r) % JB)
        rounds = [[(i, JB + (i + r) % JB) for i in range(JB)] for r in range(JB)]
    for pairs in rounds:
        R = np.eye(n)
        for p, q in pairs:
            p, q = min(p, q), max(p, q)
            c, s = rot(G[p, p], G[q, q], G[p, q])
            R[p, p], R[q, q], R[p, q], R[q, p] = c, c, s, -s
        G = R.T @ G @ R
        Q = Q @ R
    return Q


def run(W, policy, tol=1e-13, max_sweeps=40):
    W = W.copy()
    nb = W.shape[0] // JB
    for sweep in range(max_sweeps):
        off = 0.0
        visited = set()
        for r in range(nb - 1):
            for k in range(nb // 2):
                P, Qb = circle(nb, r, k)
                P, Qb = min(P, Qb), max(P, Qb)
                idx = np.r_[P * JB:(P + 1) * JB, Qb * JB:(Qb + 1) * JB]
                X = W[idx]
                G = X @ X.T
                d = np.sqrt(np.maximum(np.diag(G), 1e-300))
                C = np.abs(G) / np.outer(d, d)
                np.fill_diagonal(C, 0)
                off = max(off, C.max())
                if policy == 'full':
                    mode = 'full'
                elif policy == 'cross':
                    mode = 'cross'
                elif policy.startswith('every'):   # full in every k-th round of the sweep, cross-only in between
                    mode = 'full' if r % int(policy[5:]) == 0 else 'cross'
                else:  # first visit of either block in this sweep: full
                    mode = 'full' if (P not in visited or Qb not in visited) else 'cross'
                visited.update((P, Qb))
                Qm = inner(G, mode)
                X = Qm.T @ X
                order = np.argsort(-np.sum(X * X, axis=1), kind='stable')
                W[idx] = X[order]
        if off <= tol ** 0.5 / 10:
            return sweep + 1, off
    return -1, off


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(0)
    for name, A in (('full rank', rng.standard_normal((n, n))), ('theta-like rank n/2', rng.standard_normal((n, n // 2)) @ rng.standard_normal((n // 2, n))),
                    ('graded', rng.standard_normal((n, n)) * np.logspace(0, -8, n))):
        R = np.linalg.qr(A)[1]
        keep = np.linalg.norm(R, axis=1) > 1e-12 * np.linalg.norm(A)
        W = R[keep]
        W = W[:(W.shape[0] // (2 * JB)) * 2 * JB]
        pols = sys.argv[3].split(',') if len(sys.argv) > 3 else ('full', 'cross', 'first-full')
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        for policy in pols:
            sw, off = run(W, policy)
            print(f'[model] n={n} {name}: {policy}: {sw} sweeps (last off {off:.1e})', flush=True)


def second_qr_experiment(n=256):
    """sweeps of the block iteration on R (rows of the first QR factor, what the pipeline does) against L of a second
    factorisation R = L Q2 (Drmac-Veselic preconditioning without pivoting)."""
    rng = np.random.default_rng(1)
    for name, A in (('full rank', rng.standard_normal((n, n))), ('theta-like rank n/2', rng.standard_normal((n, n // 2)) @ rng.standard_normal((n // 2, n))),
                    ('graded', rng.standard_normal((n, n)) * np.logspace(0, -8, n))):
        R = np.linalg.qr(A)[1]
        keep = np.linalg.norm(R, axis=1) > 1e-12 * np.linalg.norm(A)
        W = R[keep]
        W = W[:(W.shape[0] // (2 * JB)) * 2 * JB]
        L = np.linalg.qr(W.T)[1].T
        for label, M in (('rows of R', W), ('rows of L (second QR)', L), ('columns of L', L.T.copy())):
            sw, off = run(M, 'full')
            print(f'[model2] n={n} {name}: {label}: {sw} sweeps', flush=True)


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == 'second':
    second_qr_experiment(int(sys.argv[1]))
