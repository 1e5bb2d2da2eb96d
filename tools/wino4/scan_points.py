import numpy as np, sys
from fractions import Fraction as Fr
sys.path.insert(0, '/root/repo/tools/wino4')
from numerics import cook_toom, tofloat, conv_wino32, conv_direct64

def run(C, K, pts, seeds=(0, 1)):
    ATf, Gf, BTf = cook_toom(pts, 4, 3)
    AT, G, BT = map(tofloat, (ATf, Gf, BTf))
    errs = []
    for seed in seeds:
        rng = np.random.default_rng(seed)
        H = W = 16
        x = rng.standard_normal((H + 2, W + 2, C))
        x = np.maximum(x, 0) * 1.3 + 0.1 * rng.standard_normal(x.shape)
        w = rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (9 * C))
        ref = conv_direct64(x, w)
        errs.append(np.abs(conv_wino32(x, w, AT, G, BT, 4) - ref).max() / np.abs(ref).max())
    return max(errs)

cands = []
for a in [Fr(1, 2), Fr(5, 8), Fr(3, 4), Fr(7, 8), Fr(1)]:
    for b in [Fr(1), Fr(5, 4), Fr(3, 2), Fr(7, 4), Fr(2)]:
        if b > a:
            cands.append((a, b))
for a, b in cands:
    pts = [0, a, -a, b, -b]
    e64 = run(64, 64, pts)
    e256 = run(256, 256, pts)
    print('a=%-5s b=%-5s  C64 %.2e   C256 %.2e' % (a, b, e64, e256))
