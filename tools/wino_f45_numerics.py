"""Paper estimate for a nested F(4,5) x F(4,3) Winograd conv: would 3.0 multiplies per output instead of the
4.5 of the production F(2,5) x F(4,3) kernel survive fp32?  Simulates both nestings the way the kernels compute - filter transform
in double then rounded to fp32, input transform in fp32, products accumulated sequentially in fp32 over 256 input channels (the
MFMA's fp32 accumulate), output transform in fp32 - against the float64 direct convolution, and tries other interpolation points.
CPU only (numpy).  Output -> profiles/r05_wino_f45_estimate.txt (with the performance model)."""
import numpy as np
from fractions import Fraction as Fr
def cook_toom(m, r, pts):
    """F(m,r): returns AT [m x n], G [n x r], BT [n x n] (n = m+r-1) with points pts (n-1 finite + infinity), exact rationals -> float64"""
    n = m + r - 1
    assert len(pts) == n - 1
    # Lagrange / Toom-Cook per Lavin: AT[i][j] = pts[j]^i (last col for inf: only i=m-1), G[j][k] = pts[j]^k / N_j, BT from polynomial product
    P = [Fr(p) for p in pts]
    # N_j = prod_{k!=j} (p_j - p_k)
    N = []
    for j in range(n - 1):
        v = Fr(1)
        for k in range(n - 1):
            if k != j: v *= (P[j] - P[k])
        N.append(v)
    AT = [[P[j] ** i for j in range(n - 1)] + [Fr(1) if i == m - 1 else Fr(0)] for i in range(m)]
    G = [[P[j] ** k / N[j] for k in range(r)] for j in range(n - 1)] + [[Fr(1) if k == r - 1 else Fr(0) for k in range(r)]]
    # BT: rows j<n-1: coefficients of prod_{k!=j}(x - p_k) ; last row: coefficients of prod_k (x - p_k)
    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b): out[i + j] += x * y
        return out
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for k in range(n - 1):
            if k != j: poly = polymul(poly, [-P[k], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for k in range(n - 1): poly = polymul(poly, [-P[k], Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in row] for row in M])
    return f(AT), f(G), f(BT)
def check(m, r, pts):
    AT, G, BT = cook_toom(m, r, pts)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()
p6 = [0, 1, -1, 2, -2]
p8 = [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)]
print("check F(2,5)", check(2, 5, p6), "F(4,3)", check(4, 3, p6), "F(4,5)", check(4, 5, p8))
def nested_err(mf, pts_f, Cin=256, trials=3, seed=1):
    ATf, Gf, BTf = cook_toom(mf, 5, pts_f)
    ATt, Gt, BTt = cook_toom(4, 3, p6)
    nf, nt = mf + 4, 6
    rng = np.random.default_rng(seed)
    errs = []
    for _ in range(trials):
        D = rng.standard_normal((Cin, nf, nt)).astype(np.float32)          # input patches (unit variance like GELU outputs ~)
        W = (rng.standard_normal((Cin, 5, 3)) / np.sqrt(Cin * 15)).astype(np.float32)
        # weights transformed in double, rounded to fp32
        V = np.einsum('fk,ckw,tw->cft', Gf, W.astype(np.float64), Gt).astype(np.float32)
        # input transform in fp32
        U = np.einsum('fa,cab->cfb', BTf.astype(np.float32), D).astype(np.float32)
        U = np.einsum('cfb,tb->cft', U, BTt.astype(np.float32)).astype(np.float32)
        # products accumulated in fp32 sequentially over channels
        M = np.zeros((nf, nt), np.float32)
        for c in range(Cin):
            M = (M + U[c] * V[c]).astype(np.float32)
        Y = (ATf.astype(np.float32) @ M).astype(np.float32)
        Y = (Y @ ATt.astype(np.float32).T).astype(np.float32)
        ref = np.zeros((mf, 4))
        for i in range(mf):
            for j in range(4):
                ref[i, j] = np.sum(D[:, i:i + 5, j:j + 3].astype(np.float64) * W.astype(np.float64))
        errs.append((np.abs(Y - ref).max(), np.sqrt(np.mean(ref ** 2))))
    return errs
for name, mf, pts in (("F(2,5)xF(4,3)", 2, p6), ("F(4,5)xF(4,3)", 4, p8)):
    e = nested_err(mf, pts, trials=20)
    print(name, "max abs err / rms(ref): mean", np.mean([a / b for a, b in e]), "max", np.max([a / b for a, b in e]))
# alternative points for F(4,5): 0, +-1, +-2, +-1/2 vs 0,+-1,+-1/2,+-2 same; try +-1/2,+-1,+-3/2? 
for alt in ([0, 1, -1, Fr(1,2), Fr(-1,2), 2, -2], [0, 1, -1, Fr(1,2), Fr(-1,2), Fr(3,2), Fr(-3,2)], [0,1,-1,2,-2,3,-3], [0, 1, -1, Fr(1,2), Fr(-1,2), Fr(1,4), Fr(-1,4)]):
    e = nested_err(4, alt, trials=10)
    print("F(4,5) pts", [str(a) for a in alt], np.mean([a / b for a, b in e]))
