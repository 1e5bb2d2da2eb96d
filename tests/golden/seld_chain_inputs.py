"""Inputs of the end-to-end SELD chain fixture (``seld_chain.npz``): the 4-channel int16 clips are regenerated from their
seeds by this function -- in ``make_golden.py`` (build container, fed to the REAL reference) and in the tests (fed to the code
under test) -- instead of being stored (6.7 MB); the fixture keeps each clip's CRC32 so a differing regeneration is caught.
Our own code; there is no reference counterpart."""
import zlib

import numpy as np

CLIPS = (("fold6_room1_mix001", 4101, 240000), ("fold6_room1_mix002", 4102, 240000 + 77), ("fold6_room2_mix003", 4103, 360000))


def chain_clip(seed, n_samples):
    """(n_samples, 4) int16: a noise floor plus a handful of tone / noise bursts with per-channel gains (a crude FOA
    encoding of a direction), so that the features change over the clip.  Integer noise from the legacy RandomState stream
    (stable across numpy versions); the bursts go through float64 sin / round."""
    rs = np.random.RandomState(seed)
    pcm = rs.randint(-300, 301, size=(n_samples, 4)).astype(np.float64)
    t = np.arange(n_samples) / 24000.0
    nb = 6 + n_samples // 60000
    for _ in range(nb):
        start = int(rs.randint(0, max(1, n_samples - 24000)))
        length = int(rs.randint(12000, 72000))
        end = min(n_samples, start + length)
        az, el = rs.uniform(-np.pi, np.pi), rs.uniform(-1.0, 1.0)
        gains = np.array([1.0, np.sin(az) * np.cos(el), np.sin(el), np.cos(az) * np.cos(el)])      # W, Y, Z, X
        amp = float(rs.uniform(1500.0, 9000.0))
        if rs.rand() < 0.5:
            sig = np.sin(2.0 * np.pi * float(rs.uniform(200.0, 6000.0)) * t[start:end])
        else:
            sig = rs.standard_normal(end - start)
        env = np.minimum(1.0, np.minimum(np.arange(end - start), np.arange(end - start)[::-1]) / 1200.0)
        pcm[start:end] += amp * (sig * env)[:, None] * gains[None, :]
    return np.clip(np.round(pcm), -32768, 32767).astype(np.int16)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
