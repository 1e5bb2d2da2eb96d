"""K1 front end: raw 4-channel FOA audio -> 7-channel (log-mel x4 + mel-scale intensity vector x3) features
on the GPU.  Host-side counterpart of ``FeatureLabelProcessor.get_feature``
(/root/reference/src/datasets.py:187-207, :252-292): this class only builds the constant tables
(twiddles, Hann window, sparse mel filter bank -- all in float64 on the host, once) and launches
``adyolo_feat_stft_mel`` + ``adyolo_feat_finish`` (csrc/features.hip).

The mel filter bank follows librosa==0.8.1 ``filters.mel(sr=24000, n_fft=1200, n_mels=64)`` semantics
(Slaney scale, Slaney area norm, float32) -- see SURVEY.md Appendix B; librosa itself is not available
in this image, so the formula is restated here (product code; the oracle has its own copy).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

SR = 24000
N_FFT = 1200
HOP = 600
N_MELS = 64


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))


def slaney_mel_matrix(sr=SR, n_fft=N_FFT, n_mels=N_MELS):
    """(n_mels, n_fft//2+1) float32 triangular filters, area-normalised."""
    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(0.0), _slaney_hz_to_mel(sr / 2.0), n_mels + 2))
    width = np.diff(edges)
    up = (freqs[None, :] - edges[:-2, None]) / width[:-1, None]
    down = (edges[2:, None] - freqs[None, :]) / width[1:, None]
    tri = np.maximum(0.0, np.minimum(up, down))
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return tri.astype(np.float32)


class FeatureExtractor:
    """``FeatureExtractor(scaler, device)(audio)``: audio float32 (B, n_samples, 4) on the GPU, already
    ``int16 / 32768 + 1e-8`` (datasets.py:147) -> features.

    scaler: dict like the reference's ``scaler_wts.pkl`` ({'MEL'|'IV': {'mean','std'}} with shapes
    (1,64,4)/(1,64,3)) or None for mean 0 / std 1.
    """

    def __init__(self, scaler=None, device="cuda:0"):
        self.device = torch.device(device)
        n = np.arange(N_FFT, dtype=np.float64)
        tw = np.stack([np.cos(2.0 * np.pi * n / N_FFT), -np.sin(2.0 * np.pi * n / N_FFT)], axis=1)
        # the same table entries once more in the order passes 1 and 2 of the transform read them (adyolo_hip.h, K1):
        # [k-1][st] = tw[(st k) mod 1200] and [k-1][n3] = tw[10 n3 k] -- neighbouring lanes read neighbouring entries
        k = np.arange(1, 10)[:, None]
        tw = np.concatenate([tw, tw[(k * np.arange(120)[None, :]) % N_FFT].reshape(-1, 2),
                             tw[(10 * k * np.arange(12)[None, :])].reshape(-1, 2)], axis=0)
        mel = slaney_mel_matrix()
        # non-zero weights filter after filter, each (contiguous, triangular) filter cut into pieces of <= 8 bins so
        # that the kernel's work items are balanced (filters span 2 .. ~70 bins)
        ck_mel, ck_start, ck_len, ck_off, weights = [], [], [], [], []
        for m in range(N_MELS):
            nz = np.nonzero(mel[m])[0]
            s, e = int(nz[0]), int(nz[-1]) + 1
            for c0 in range(s, e, 8):
                ck_mel.append(m)
                ck_start.append(c0)
                ck_len.append(min(8, e - c0))
                ck_off.append(len(weights) + (c0 - s))
            weights.extend(mel[m, s:e].tolist())
        mean = np.zeros((7, N_MELS), dtype=np.float64)
        std = np.ones((7, N_MELS), dtype=np.float64)
        if scaler is not None:
            mean[:4] = np.asarray(scaler["MEL"]["mean"], dtype=np.float64).reshape(N_MELS, 4).T
            std[:4] = np.asarray(scaler["MEL"]["std"], dtype=np.float64).reshape(N_MELS, 4).T
            mean[4:] = np.asarray(scaler["IV"]["mean"], dtype=np.float64).reshape(N_MELS, 3).T
            std[4:] = np.asarray(scaler["IV"]["std"], dtype=np.float64).reshape(N_MELS, 3).T
        dev = self.device
        f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev).contiguous()   # noqa: E731
        i32 = lambda a: torch.tensor(np.asarray(a, dtype=np.int32), device=dev).contiguous()     # noqa: E731
        self.twiddle = f32(tw)
        self.ck_mel, self.ck_start, self.ck_len, self.ck_off = i32(ck_mel), i32(ck_start), i32(ck_len), i32(ck_off)
        self.n_chunks = len(ck_mel)
        self.mel_w = f32(weights)
        self.sc_mean, self.sc_rstd = f32(mean), f32(1.0 / std)
        self.mel_nnz = len(weights)

    def __call__(self, audio, channels_last8=True, chunk_offsets=None, chunk_samples=None, validate_offsets=True):
        """-> (B, T, 64, 8) float32 channels-last (8th channel zero) when ``channels_last8`` (what the
        encoder consumes), else (B, 7, T, 64) in the reference's layout (datasets.py:158-160).

        chunk_offsets (int64 tensor on the device, sample offsets into the flattened (B * n_samples) audio) with
        chunk_samples: features of the CHUNKS ``audio.view(-1, 4)[off : off + chunk_samples]`` instead (one output row
        per offset) -- the reference's offline chunking (src/preprocess.py:13-84: 20 s windows at 1 s stride, each chunk
        its own file, hence its own reflect padding and top_db reference) without materialising the chunk audio."""
        if not audio.is_cuda or audio.dtype != torch.float32 or not audio.is_contiguous():
            raise _lib.AdyoloHipError("FeatureExtractor needs contiguous float32 audio (B, n_samples, 4) on the GPU")
        b, n, ch = audio.shape
        if ch != 4 or n % HOP != 0:
            raise _lib.AdyoloHipError("audio must be (B, n_samples, 4) with n_samples %% 600 == 0")
        offs = None
        if chunk_offsets is not None:
            if (chunk_offsets.dtype != torch.int64 or not chunk_offsets.is_cuda or not chunk_offsets.is_contiguous()
                    or chunk_samples is None or chunk_samples % HOP != 0 or chunk_samples < N_FFT):
                raise _lib.AdyoloHipError("chunk_offsets must be a contiguous int64 device tensor and chunk_samples a "
                                          "multiple of 600 (>= 1200)")
            if validate_offsets and chunk_offsets.numel():
                # the kernel reads audio[off : off + chunk_samples] unchecked: an offset outside the buffer would be a silent
                # out-of-bounds read.  One host sync per call (chunking is offline preprocessing, not the train step);
                # validate_offsets=False skips it for offsets the caller has already checked
                lo, hi = int(chunk_offsets.min()), int(chunk_offsets.max())
                if lo < 0 or hi + int(chunk_samples) > b * n:
                    raise _lib.AdyoloHipError("chunk_offsets must satisfy 0 <= off <= %d - chunk_samples (got min %d, max %d)"
                                              % (b * n, lo, hi))
            offs, b, n = chunk_offsets, chunk_offsets.numel(), int(chunk_samples)
        t = n // HOP
        layout = 1 if channels_last8 else 0
        out = torch.empty((b, t, N_MELS, 8) if channels_last8 else (b, 7, t, N_MELS), dtype=torch.float32,
                          device=audio.device)
        chan_max = torch.empty(b * 4, dtype=torch.float32, device=audio.device)
        st = _stream()
        _lib.call("adyolo_feat_stft_mel", _p(audio), _p(offs), _p(self.twiddle), _p(self.ck_mel),
                  _p(self.ck_start), _p(self.ck_len), _p(self.ck_off), _p(self.mel_w), self.n_chunks, self.mel_nnz,
                  _p(self.sc_mean), _p(self.sc_rstd), _p(out), _p(chan_max), b, n, layout, st)
        _lib.call("adyolo_feat_finish", _p(out), _p(chan_max), _p(self.sc_mean), _p(self.sc_rstd), b, t, layout, st)
        return out

    @staticmethod
    def algorithmic_bytes(b, n_samples):
        """HBM bytes one call must move: read 4ch fp32 audio + write 7 x T x 64 fp32."""
        return b * (n_samples * 4 * 4 + 7 * (n_samples // HOP) * N_MELS * 4)


class MicFeatureExtractor:
    """MIC-format feature set of BASELINE config 5 ("DCASE2022 MIC (GCC-PHAT features)"): log-mel of the four microphones
    (K1 on the MIC audio) + six GCC-PHAT channels (``adyolo_feat_gcc_phat``, csrc/features_mic.hip) -> 10 features.
    NOT in the reference, which hard-codes FOA (src/datasets.py:36-37,55; src/main.py:40): parity unpinned; the definition
    is the DCASE2022 SELD baseline's (see include/adyolo_hip.h).

    scaler: {'MEL': {'mean','std'} (1,64,4), 'GCC': {'mean','std'} (1,64,6)} or None.
    ``__call__(audio (B, n, 4))`` -> (B, T, 64, 32) channels-last float32 (features 0-9, zeros above: the 32-channel pixel the
    Winograd stem convolution consumes) or, with ``channels_last=False``, (B, 10, T, 64)."""

    N_FEATURES = 10

    def __init__(self, scaler=None, device="cuda:0"):
        mel_scaler = None
        mean, std = np.zeros((6, N_MELS)), np.ones((6, N_MELS))
        if scaler is not None:
            mel_scaler = {"MEL": scaler["MEL"], "IV": {"mean": np.zeros((1, N_MELS, 3)), "std": np.ones((1, N_MELS, 3))}}
            mean = np.asarray(scaler["GCC"]["mean"], dtype=np.float64).reshape(N_MELS, 6).T
            std = np.asarray(scaler["GCC"]["std"], dtype=np.float64).reshape(N_MELS, 6).T
        self.k1 = FeatureExtractor(mel_scaler, device)
        dev = self.k1.device
        self.gcc_mean = torch.tensor(np.asarray(mean, dtype=np.float32), device=dev).contiguous()
        self.gcc_rstd = torch.tensor(np.asarray(1.0 / std, dtype=np.float32), device=dev).contiguous()

    def __call__(self, audio, channels_last=True):
        if not audio.is_cuda or audio.dtype != torch.float32 or not audio.is_contiguous():
            raise _lib.AdyoloHipError("MicFeatureExtractor needs contiguous float32 audio (B, n_samples, 4) on the GPU")
        b, n, ch = audio.shape
        if ch != 4 or n % HOP != 0:
            raise _lib.AdyoloHipError("audio must be (B, n_samples, 4) with n_samples %% 600 == 0")
        t = n // HOP
        out = torch.zeros((b, t, N_MELS, 32), dtype=torch.float32, device=audio.device)
        mel8 = self.k1(audio, channels_last8=True)                    # (B, T, 64, 8): channels 0-3 = log-mel of the four microphones
        out[..., :4] = mel8[..., :4]                                  # (plumbing copy; the intensity-vector channels are not used)
        _lib.call("adyolo_feat_gcc_phat", _p(audio), _p(None), _p(self.k1.twiddle), _p(self.gcc_mean), _p(self.gcc_rstd),
                  _p(out), b, n, 32, 4, _stream())
        if channels_last:
            return out
        return out[..., :10].permute(0, 3, 1, 2).contiguous()


def load_scaler_npz(path):
    """Scaler statistics stored as .npz (mel_mean/mel_std (1,64,4), iv_mean/iv_std (1,64,3))."""
    z = np.load(path)
    return {"MEL": {"mean": z["mel_mean"], "std": z["mel_std"]}, "IV": {"mean": z["iv_mean"], "std": z["iv_std"]}}


class ScalerFitter:
    """Train-set feature statistics on the GPU (reference src/preprocess.py:86-130: mean / std / max / min over all frames,
    per mel bin and channel, population std).  ``partial_fit(audio)`` runs K1 WITHOUT a scaler on a batch of raw audio
    and accumulates per-column sums in float64; ``finalize()`` returns the reference's ``scaler_wts.pkl`` dictionary
    ({'MEL'|'IV': {'mean','std','max','min'}} with shapes (1, 64, 4) / (1, 64, 3))."""

    def __init__(self, device="cuda:0"):
        self.fx = FeatureExtractor(None, device)
        self.n = 0
        self.acc = None

    def partial_fit(self, audio):
        from . import ops
        feat = self.fx(audio, channels_last8=True)                  # (B, T, 64, 8)
        b, t, f, c = feat.shape
        st = ops.colstats(feat.view(b * t, f * c))                  # float64 [4][512]
        if self.acc is None:
            self.acc = st.clone()
        else:
            self.acc[0] += st[0]
            self.acc[1] += st[1]
            self.acc[2] = torch.maximum(self.acc[2], st[2])
            self.acc[3] = torch.minimum(self.acc[3], st[3])
        self.n += b * t
        return self

    def finalize(self):
        a = self.acc.cpu().numpy().reshape(4, N_MELS, 8)
        mean = a[0] / self.n
        var = np.maximum(a[1] / self.n - mean * mean, 0.0)
        out = {"MEL": {}, "IV": {}}
        for key, sl in (("MEL", slice(0, 4)), ("IV", slice(4, 7))):
            out[key]["mean"] = mean[None, :, sl].copy()
            out[key]["std"] = np.sqrt(var)[None, :, sl].copy()
            out[key]["max"] = a[2][None, :, sl].copy()
            out[key]["min"] = a[3][None, :, sl].copy()
        return out
