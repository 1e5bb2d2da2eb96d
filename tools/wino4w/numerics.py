"""Error study for a weight gradient in the F(4x4,3x3) Winograd domain (CPU, numpy; round 5):
    dw = G^T [ sum_tiles (B^T d B) (.) (A e A^T) ] G     d: 6x6 input tile, e: 4x4 tile of dy, A = (A^T)^T of the forward F(4x4,3x3)
with the interpolation points of csrc/wino4.hip (0, +-3/4, +-3/2, inf), fp32 transforms and fp32 accumulation over the tiles
in the order an MFMA chain would sum them (sequential in chunks of two tiles), against a float64 direct weight gradient -- and the
F(2x2,3x3) form (csrc/wino.hip) beside it under the same accumulation.  Prints max error relative to the gradient's absmax."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wino4.numerics import cook_toom, tofloat


def wgrad_direct64(x, dy):
    # x [N][H+2][W+2][Ci] (zero padded), dy [N][H][W][Co] -> dw [Co][Ci][3][3]
    N, H, W, Co = dy.shape
    dw = np.zeros((Co, x.shape[3], 3, 3))
    for a in range(3):
        for b in range(3):
            dw[:, :, a, b] = np.einsum("nhwo,nhwi->oi", dy, x[:, a:a + H, b:b + W, :])
    return dw


def wgrad_wino32(x, dy, m, pts, chunk=2):
    AT, G, BT = cook_toom(pts, m, 3)
    AT, G, BT = tofloat(AT), tofloat(G), tofloat(BT)
    n = m + 2
    N, H, W, Co = dy.shape
    Ci = x.shape[3]
    A32, BT32 = AT.T.astype(np.float32), BT.astype(np.float32)
    x32, dy32 = x.astype(np.float32), dy.astype(np.float32)
    V, E = [], []
    for s in range(N):
        for ty in range(H // m):
            for tx in range(W // m):
                d = x32[s, m * ty:m * ty + n, m * tx:m * tx + n, :]                      # [n][n][Ci]
                e = dy32[s, m * ty:m * ty + m, m * tx:m * tx + m, :]                     # [m][m][Co]
                v = np.einsum("ai,ijc,bj->abc", BT32, d, BT32).astype(np.float32)       # B^T d B
                u = np.einsum("ai,ijc,bj->abc", A32, e, A32).astype(np.float32)         # A e A^T
                V.append(v)
                E.append(u)
    V, E = np.stack(V), np.stack(E)                                                      # [T][n][n][C]
    acc = np.zeros((n, n, Ci, Co), np.float32)
    for t0 in range(0, V.shape[0], chunk):                                               # an MFMA adds two tiles per instruction
        acc = (acc + np.einsum("tabi,tabo->abio", V[t0:t0 + chunk], E[t0:t0 + chunk]).astype(np.float32)).astype(np.float32)
    dw = np.einsum("ak,abio,bl->oikl", G, acc.astype(np.float64), G)                     # G^T dU G in double (as the finish kernel could)
    return dw, float(np.abs(V).max()), float(np.abs(E).max())


def main():
    rng = np.random.default_rng(0)
    for (N, H, W, Ci, Co) in [(2, 64, 32, 8, 8), (8, 64, 32, 8, 8), (16, 120, 16, 8, 8)]:
        x = np.zeros((N, H + 2, W + 2, Ci))
        x[:, 1:-1, 1:-1, :] = rng.standard_normal((N, H, W, Ci)) * 0.7 + 0.3            # post-BatchNorm-like activations (non-zero mean)
        dy = rng.standard_normal((N, H, W, Co)) * 1e-3
        ref = wgrad_direct64(x, dy)
        am = np.abs(ref).max()
        d32 = wgrad_direct64(x.astype(np.float32).astype(np.float64), dy.astype(np.float32).astype(np.float64))
        line = "N=%d %dx%d tiles(4x4)=%d: " % (N, H, W, N * H * W // 16)
        for name, m, pts in (("F(2x2)", 2, [0, 1, -1]), ("F(4x4) 0,+-3/4,+-3/2", 4, [0, "3/4", "-3/4", "3/2", "-3/2"]),
                             ("F(4x4) 0,+-1,+-2", 4, [0, 1, -1, 2, -2]), ("F(4x4) 0,+-1/2,+-1", 4, [0, "1/2", "-1/2", 1, -1])):
            dw, vmax, emax = wgrad_wino32(x, dy, m, pts)
            line += "%s %.2e (|V| %.1f |E| %.1e)   " % (name, np.abs(dw - ref).max() / am, vmax, emax)
        print(line, flush=True)


if __name__ == "__main__":
    main()
