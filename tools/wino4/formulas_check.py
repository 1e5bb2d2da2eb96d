"""Checks (CPU, numpy float64) of the closed-form F(4x4,3x3) transforms used by csrc/wino4.hip (points 0, +-3/4, +-3/2, inf)
and of the LDS bank behaviour of its half-transformed image layout."""
import numpy as np
a, b = 0.75, 1.5
A2, B2 = a * a, b * b
S2, P2 = A2 + B2, A2 * B2

def bt6(c):
    c0, c1, c2, c3, c4, c5 = c
    t0 = P2 * c0 + (-S2 * c2 + c4)
    E12 = -B2 * c2 + c4; O12 = -B2 * c1 + c3
    E34 = -A2 * c2 + c4; O34 = -A2 * c1 + c3
    return [t0, E12 + a * O12, E12 - a * O12, E34 + b * O34, E34 - b * O34, P2 * c1 + (-S2 * c3 + c5)]

def G():
    Na = 2 * A2 * (A2 - B2); Nb = 2 * B2 * (B2 - A2)
    return np.array([[1 / P2, 0, 0], [1 / Na, a / Na, A2 / Na], [1 / Na, -a / Na, A2 / Na],
                     [1 / Nb, b / Nb, B2 / Nb], [1 / Nb, -b / Nb, B2 / Nb], [0, 0, 1]])

def at4(m):
    m0, m1, m2, m3, m4, m5 = m
    s12, d12, s34, d34 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
    return [m0 + s12 + s34, a * d12 + b * d34, A2 * s12 + B2 * s34, a ** 3 * d12 + b ** 3 * d34 + m5]

rng = np.random.default_rng(0)
d = rng.standard_normal((6, 6)); g = rng.standard_normal((3, 3))
# V[xi][nu] = sum_i sum_j BT[xi][i] d[i][j] BT[nu][j]:  W-direction (j) first, then H-direction (i)
C = np.array([bt6(d[i, :]) for i in range(6)])        # C[i][nu]
V = np.array([bt6(C[:, nu]) for nu in range(6)]).T     # V[xi][nu]
Gm = G()
U = Gm @ g @ Gm.T                                      # U[xi][nu], g[ky][kx]
M = U * V
Q = np.array([at4(M[:, nu]) for nu in range(6)]).T     # Q[p][nu]
Y = np.array([at4(Q[p, :]) for p in range(4)])         # Y[p][q]
ref = np.array([[sum(d[p + ky, q + kx] * g[ky, kx] for ky in range(3) for kx in range(3)) for q in range(4)] for p in range(4)])
print('forward identity err', np.abs(Y - ref).max())

# weight gradient: dw[ky][kx] = sum_{p,q} d[p+ky][q+kx] e[p][q] = G^T [ (A e A^T) (.) (BT d B) ] G
e = rng.standard_normal((4, 4))
AT = np.array([[1, 1, 1, 1, 1, 0], [0, a, -a, b, -b, 0], [0, A2, A2, B2, B2, 0], [0, a ** 3, -a ** 3, b ** 3, -b ** 3, 1]])
E = AT.T @ e @ AT
dw = Gm.T @ (E * V) @ Gm
refw = np.array([[sum(d[p + ky, q + kx] * e[p, q] for p in range(4) for q in range(4)) for kx in range(3)] for ky in range(3)])
print('wgrad identity err', np.abs(dw - refw).max())
print('G =\n', Gm)

# ---- LDS layout of the half-transformed image C[nu][q][y][tc] (16-byte slots) and the ds_read_b128 / ds_write_b128 lane groups
def blkrow(y):
    q, r = y >> 2, y & 3
    return 4 * q + ((r + q) & 3)

RD_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
for TC in (4, 8):
    TR = 32 // TC
    PR = 4 * TR + 2
    PS = PR * TC + 2                     # plane stride in slots, == 2 mod 8
    assert PS % 8 == 2, PS
    worst = 1
    for i in range(6):
        for grp in RD_GROUPS:
            slots = []
            for li in grp:
                tr, tc = li // TC, li % TC
                slots.append((blkrow(4 * tr + i) * TC + tc) % 16)
            worst = max(worst, max(slots.count(s) for s in set(slots)))
    print('TC=%d: plane %d slots (%d B per buffer); ds_read_b128 worst conflict %d-way' % (TC, PS, 12 * PS * 16, worst))
    # writes: lane = (eo, q, tc, y...) ; 8 consecutive lanes per group, bank window 8 slots (128 B)
    worstw = 1
    for rnd_y0 in range(0, PR):
        for (nuE, nuO) in ((0, 5), (1, 2), (3, 4)):
            for g8 in range(0, 4 * TC, 8):
                slots = []
                for l in range(g8, g8 + 8):
                    eo, q, tc = l & 1, (l >> 1) & 1, (l >> 2) % TC
                    y = rnd_y0 + (l >> 2) // TC
                    if y >= PR:
                        continue
                    nu = nuO if eo else nuE
                    slots.append(((nu * 2 + q) * PS + blkrow(y) * TC + tc) % 8)
                worstw = max(worstw, max(slots.count(s) for s in set(slots)))
    print('        ds_write_b128 worst conflict %d-way' % worstw)
