"""numpy model of a CholeskyQR2 + Householder-reconstruction panel inside a blocked QR (32-column panels, fall-back to the
Householder panel at a Cholesky pivot ratio of 1e5): how many panels take the fast path and what it costs in accuracy, on a
theta-like Gaussian product, a graded block and the sectors of a DMRG theta (gpurun_out/theta_center.npz if present).
DESIGN.md section 8, item 1: modelled and rejected."""
import numpy as np, sys
np.set_printoptions(linewidth=200)
def house_panel(P):
    """reference: Householder panel -> V (unit lower trapezoidal), T, R (b x b upper)"""
    m,b=P.shape; A=P.copy(); V=np.zeros((m,b)); taus=np.zeros(b)
    for j in range(b):
        x=A[j:,j]; alpha=x[0]; xn=np.linalg.norm(x[1:])
        if xn==0: tau=0; v=np.zeros(m-j); v[0]=1; beta=alpha
        else:
            beta=-np.copysign(np.hypot(alpha,xn),alpha); tau=(beta-alpha)/beta; v=x/(alpha-beta); v[0]=1
        A[j:,j:]-=tau*np.outer(v,v@A[j:,j:]); V[j:,j]=v; taus[j]=tau
    T=np.zeros((b,b))
    for j in range(b):
        T[j,j]=taus[j]
        if j: T[:j,j]=-taus[j]*T[:j,:j]@(V[:,:j].T@V[:,j])
    return V,T,np.triu(A[:b])
def chol_panel(P, kmax=1e5):
    """CholeskyQR2 + Householder reconstruction. Returns (V,T,R) or None (fallback)."""
    m,b=P.shape
    G=P.T@P
    try: R1=np.linalg.cholesky(G).T
    except np.linalg.LinAlgError: return None
    d=np.abs(np.diag(R1)); 
    if d.min()<=0 or d.max()/d.min()>kmax: return None
    Q1=P@np.linalg.inv(R1)
    G2=Q1.T@Q1
    try: R2=np.linalg.cholesky(G2).T
    except np.linalg.LinAlgError: return None
    M=np.linalg.inv(R1)@np.linalg.inv(R2)
    Q=P@M
    R=R2@R1
    # modified LU of Q - [S;0]
    A=Q[:b].copy(); S=np.zeros(b); L=np.eye(b); U=np.zeros((b,b))
    for j in range(b):
        S[j]=-1.0 if A[j,j]>=0 else 1.0
        A[j,j]-=S[j]
        U[j,j:]=A[j,j:]
        L[j+1:,j]=A[j+1:,j]/A[j,j]
        A[j+1:,j+1:]-=np.outer(L[j+1:,j],A[j,j+1:])
    Y1=L; Uinv=np.linalg.inv(U)
    Y=np.vstack([Y1,(Q[b:])@Uinv])
    T=-U@np.diag(S)@np.linalg.inv(Y1).T
    return Y,T,np.diag(S)@R
def blocked_qr(A,panel,b=32):
    m,n=A.shape; A=A.copy(); k=min(m,n); Q=np.eye(m); nfb=0; npan=0
    for j0 in range(0,k,b):
        pw=min(b,k-j0); P=A[j0:,j0:j0+pw]; npan+=1
        res=None
        if panel=='chol' and pw==b and m-j0>=2*b: res=chol_panel(P)
        if res is None:
            res=house_panel(P); nfb+=(panel=='chol')
        V,T,R=res
        A[j0:,j0:j0+pw]=0; A[j0:j0+pw,j0:j0+pw]=R
        A[j0:,j0+pw:]-=V@(T.T@(V.T@A[j0:,j0+pw:]))
        Q[:,j0:]=Q[:,j0:]-(Q[:,j0:]@V)@T@V.T
    return Q,np.triu(A),nfb,npan
rng=np.random.default_rng(0)
cases={}
n=512; cases['theta-like 512 rank 256']=rng.standard_normal((n,n//2))@rng.standard_normal((n//2,n))
cases['gauss 400x300']=rng.standard_normal((400,300))
q1,_=np.linalg.qr(rng.standard_normal((140,140))); q2,_=np.linalg.qr(rng.standard_normal((140,140)))
cases['graded 140 (14 decades)']=(q1*np.logspace(0,-14,140))@q2
try:
    d=np.load('/root/repo/gpurun_out/theta_center.npz'); cases['dmrg b3 140']=d['b3']; cases['dmrg b2 84']=d['b2']
except Exception as e: print(e)
cases['ones 200']=np.ones((200,200))
for name,A in cases.items():
    for panel in ('house','chol'):
        Q,R,nfb,npan=blocked_qr(A,panel)
        nrm=np.linalg.norm(A)
        print(f'{name:28s} {panel:5s} recon {np.abs(Q@R[:A.shape[0]]-A).max()/nrm:.1e} ortho {np.abs(Q.T@Q-np.eye(len(Q))).max():.1e} fallback {nfb}/{npan}')
    # second stage: LQ of R_g (rows of R above threshold)
