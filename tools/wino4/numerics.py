"""Error study for Winograd F(4x4,3x3) in fp32 (CPU, numpy): transform matrices from interpolation points, error vs a float64
direct convolution relative to the output's absmax, for several point sets / scalings; F(2x2,3x3) and direct fp32 beside it."""
import numpy as np
from fractions import Fraction as Fr

def cook_toom(points, m, r):
    """F(m, r) with n = m + r - 1 points (the last one is infinity).  Returns AT (m x n), G (n x r), BT (n x n) as Fractions:
    y = AT [ (G g) . (BT d) ]"""
    n = m + r - 1
    pts = [Fr(p) for p in points]          # n - 1 finite points
    assert len(pts) == n - 1
    # A^T: rows i = 0..m-1, columns = points: p^i, infinity column: 1 for i = m-1
    AT = [[pts[j] ** i for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    # G: rows = points: p^k / N_j, where N_j = prod_{l != j}(p_j - p_l); infinity row: [0..0 1]
    G = []
    for j in range(n - 1):
        N = Fr(1)
        for l in range(n - 1):
            if l != j:
                N *= (pts[j] - pts[l])
        G.append([pts[j] ** k / N for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    # B^T from the polynomial identity: rows j<n-1: coefficients of prod_{l != j}(x - p_l) ; last row: coefficients of prod_l (x - p_l)
    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for k, y in enumerate(b):
                out[i + k] += x * y
        return out
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for l in range(n - 1):
            if l != j:
                poly = polymul(poly, [-pts[l], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for l in range(n - 1):
        poly = polymul(poly, [-pts[l], Fr(1)])
    BT.append(poly)
    return AT, G, BT

def tofloat(M):
    return np.array([[float(x) for x in row] for row in M], dtype=np.float64)

def check_exact(AT, G, BT, m, r):
    rng = np.random.default_rng(0)
    g = rng.standard_normal(r); d = rng.standard_normal(m + r - 1)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()

def rescale(AT, G, BT, s):
    """row scaling of BT by s_j, G by 1/s_j (any nonzero s keeps the identity)"""
    s = np.asarray(s, dtype=np.float64)
    return AT, G / s[:, None], BT * s[:, None]

def rescale3(AT, G, BT, sa, sb):
    """y = AT diag(sa)^-1 ... : column scaling of AT by 1/sa, G rows by sa*..: AT' = AT / sa, G' = G * sa / sb, BT' = BT * sb"""
    sa = np.asarray(sa, dtype=np.float64); sb = np.asarray(sb, dtype=np.float64)
    return AT / sa[None, :], G * (sa / sb)[:, None], BT * sb[:, None]

def conv_direct64(x, w):
    # x [H+2][W+2][C], w [K][C][3][3] -> y [H][W][K]
    H, W = x.shape[0] - 2, x.shape[1] - 2
    y = np.zeros((H, W, w.shape[0]))
    for a in range(3):
        for b in range(3):
            y += x[a:a + H, b:b + W, :] @ w[:, :, a, b].T
    return y

def conv_direct32(x, w):
    x = x.astype(np.float32); w = w.astype(np.float32)
    H, W = x.shape[0] - 2, x.shape[1] - 2
    y = np.zeros((H, W, w.shape[0]), np.float32)
    for a in range(3):
        for b in range(3):
            y += x[a:a + H, b:b + W, :] @ w[:, :, a, b].T
    return y

def conv_wino32(x, w, AT, G, BT, m, filt64=True):
    """fp32 data path: input transform, per-position GEMM (fp32), output transform all in float32; filter transform in
    float64 then rounded (filt64) or in float32"""
    n = m + 2
    H, W = x.shape[0] - 2, x.shape[1] - 2
    assert H % m == 0 and W % m == 0
    x32 = x.astype(np.float32)
    AT32, BT32 = AT.astype(np.float32), BT.astype(np.float32)
    if filt64:
        U = np.einsum('ia,kcab,jb->ijck', G, w, G).astype(np.float32)          # [n][n][C][K]
    else:
        G32 = G.astype(np.float32)
        U = np.einsum('ia,kcab,jb->ijck', G32, w.astype(np.float32), G32).astype(np.float32)
    th, tw = H // m, W // m
    # tiles
    d = np.zeros((th, tw, n, n, x.shape[2]), np.float32)
    for i in range(th):
        for j in range(tw):
            d[i, j] = x32[i * m:i * m + n, j * m:j * m + n, :]
    # V = BT d B  (float32 arithmetic, sequential adds like the kernel: emulate with float32 einsum in two steps)
    t = np.einsum('pi,tuijc->tupjc', BT32, d).astype(np.float32)
    V = np.einsum('qj,tupjc->tupqc', BT32, t).astype(np.float32)
    M = np.einsum('tupqc,pqck->tupqk', V, U).astype(np.float32)     # float32 matmul (numpy accumulates pairwise in f32)
    s = np.einsum('ap,tupqk->tuaqk', AT32, M).astype(np.float32)
    Y = np.einsum('bq,tuaqk->tuabk', AT32, s).astype(np.float32)
    y = np.zeros((H, W, w.shape[0]), np.float32)
    for i in range(th):
        for j in range(tw):
            y[i * m:(i + 1) * m, j * m:(j + 1) * m, :] = Y[i, j]
    return y

def study(C, K, seed=0, relu_input=True):
    rng = np.random.default_rng(seed)
    H = W = 16
    x = rng.standard_normal((H + 2, W + 2, C))
    if relu_input:
        x = np.maximum(x, 0) * 1.3 + 0.1 * rng.standard_normal(x.shape)      # post-ReLU-BN-ish: biased positive
    w = rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (9 * C))
    ref = conv_direct64(x, w)
    amax = np.abs(ref).max()
    out = {}
    out['direct32'] = np.abs(conv_direct32(x, w) - ref).max() / amax
    AT, G, BT = map(tofloat, cook_toom([0, 1, -1], 2, 3))
    out['F2 std'] = np.abs(conv_wino32(x, w, AT, G, BT, 2) - ref).max() / amax
    for name, pts in [('F4 {0,1,-1,2,-2}', [0, 1, -1, 2, -2]),
                      ('F4 {0,1,-1,1/2,-1/2}', [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
                      ('F4 {0,1,-1,1/2,-2}', [0, 1, -1, Fr(1, 2), -2]),
                      ('F4 {0,1,-1,2,-1/2}', [0, 1, -1, 2, Fr(-1, 2)]),
                      ('F4 {0,1/2,-1/2,3/2,-3/2}', [0, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)]),
                      ('F4 {0,1,-1,3/2,-3/2}', [0, 1, -1, Fr(3, 2), Fr(-3, 2)]),
                      ('F4 {0,3/4,-3/4,3/2,-3/2}', [0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)]),
                      ]:
        ATf, Gf, BTf = cook_toom(pts, 4, 3)
        AT, G, BT = map(tofloat, (ATf, Gf, BTf))
        assert check_exact(AT, G, BT, 4, 3) < 1e-9, name
        out[name] = np.abs(conv_wino32(x, w, AT, G, BT, 4) - ref).max() / amax
        # row-normalised BT (each row scaled to max |coef| = 1 -> power of two scaling keeps exactness)
        s = 1.0 / np.abs(BT).max(axis=1)
        s = 2.0 ** np.round(np.log2(s))
        AT2, G2, BT2 = rescale(AT, G, BT, s)
        out[name + ' rowscaled'] = np.abs(conv_wino32(x, w, AT2, G2, BT2, 4) - ref).max() / amax
    return out

if __name__ == '__main__':
    for C, K in [(32, 32), (64, 64), (128, 128), (256, 256)]:
        res = {}
        for seed in range(3):
            r = study(C, K, seed)
            for k, v in r.items():
                res.setdefault(k, []).append(v)
        print('C=%d K=%d' % (C, K))
        for k, v in res.items():
            print('   %-36s max %.2e  mean %.2e' % (k, max(v), np.mean(v)))
