"""numpy check of the two claims behind the complex-on-the-block-engine plan:
(i)  a REAL Householder QR of the interleaved embedding M(A) of a complex block is the complex QR, up to a sign per column;
(ii) a block one-sided Jacobi iteration on M(R) whose 32 x 32 pivot rotations are M(Q_c) of the 16 x 16 Hermitian pivot
     problem converges to the complex SVD (singular values once, complex-orthonormal vectors), unlike the plain real one."""
import numpy as np

rng = np.random.default_rng(1)


def embed(A):
    m, n = A.shape
    M = np.zeros((2 * m, 2 * n))
    M[0::2, 0::2], M[0::2, 1::2] = A.real, -A.imag
    M[1::2, 0::2], M[1::2, 1::2] = A.imag, A.real
    return M


def extract(M):
    return M[0::2, 0::2] + 1j * M[1::2, 0::2]


def is_structured(M, tol=1e-12):
    return (np.abs(M[0::2, 0::2] - M[1::2, 1::2]).max() <= tol * np.abs(M).max()
            and np.abs(M[0::2, 1::2] + M[1::2, 0::2]).max() <= tol * np.abs(M).max())


# ---- (i) QR
A = rng.standard_normal((60, 40)) + 1j * rng.standard_normal((60, 40))
Q, R = np.linalg.qr(embed(A))                       # LAPACK real Householder QR
sg = np.sign(np.diag(R))
sg[sg == 0] = 1
Q, R = Q * sg, R * sg[:, None]                      # positive diagonal: now unique
print('(i) real QR of M(A): R structured', is_structured(R), ' Q structured', is_structured(Q),
      ' |A - Qc Rc|', np.abs(extract(Q) @ extract(R) - A).max(), ' |Qc^H Qc - 1|', np.abs(extract(Q).conj().T @ extract(Q) - np.eye(40)).max(),
      ' imag(diag Rc)', np.abs(np.diag(extract(R)).imag).max())


# ---- (ii) structured block Jacobi on the rows of M(R): 8 complex rows (16 real) per block
def jacobi(W, structured, sweeps=12, cb=8):
    W = W.copy()
    r = W.shape[0] // 2                             # complex rows
    nb = r // cb
    for sw in range(sweeps):
        off = 0.0
        for p in range(nb):
            for q in range(p + 1, nb):
                idx = np.r_[2 * cb * p:2 * cb * (p + 1), 2 * cb * q:2 * cb * (q + 1)]
                X = W[idx]
                G = X @ X.T
                d = np.sqrt(np.diag(G))
                off = max(off, np.abs(G / np.outer(d, d) - np.eye(len(d))).max())
                if structured:
                    Gc = extract(G)                 # 16 x 16 Hermitian (the Gram is M(Gc) up to rounding)
                    Gc = 0.5 * (Gc + Gc.conj().T)
                    w, Qc = np.linalg.eigh(Gc)
                    Qm = embed(Qc[:, ::-1])
                else:
                    w, Qm = np.linalg.eigh(0.5 * (G + G.T))
                    Qm = Qm[:, ::-1]
                W[idx] = Qm.T @ X
        if off < 1e-13:
            return W, sw + 1
    return W, sweeps


B = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
Rc = np.linalg.qr(B, mode='r')
sref = np.linalg.svd(B, compute_uv=False)
for structured in (True, False):
    W, ns = jacobi(embed(Rc), structured)
    s_all = np.sort(np.linalg.norm(W, axis=1))[::-1]
    if structured:
        Wc = extract(W)
        s = np.sort(np.linalg.norm(Wc, axis=1))[::-1]
        V = Wc / np.linalg.norm(Wc, axis=1)[:, None]
        print(f'(ii) structured pivots: {ns} sweeps, rows structured {is_structured(W, 1e-10)}, |s - s_ref| {np.abs(s - sref).max():.1e}, '
              f'|V V^H - 1| {np.abs(V @ V.conj().T - np.eye(64)).max():.1e}')
    else:
        Wc = extract(W)
        V = Wc / np.maximum(np.linalg.norm(Wc, axis=1), 1e-300)[:, None]
        print(f'(ii) plain real pivots:  {ns} sweeps, rows structured {is_structured(W, 1e-10)}, every value twice: '
              f'{np.abs(s_all[0::2] - sref).max():.1e} / {np.abs(s_all[1::2] - sref).max():.1e}; extracted complex rows orthonormal? '
              f'{np.abs(V @ V.conj().T - np.eye(64)).max():.1e}')
