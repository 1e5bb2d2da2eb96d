"""Oracle (test infrastructure): FOA feature pipeline, NumPy float64.

Restates ``/root/reference/src/datasets.py:252-292`` (``FeatureLabelProcessor``:
``get_stft_spectrogram``, ``get_logmel_spectrogram``,
``get_melscale_foa_intensity_vectors``, ``get_feature``) and the tensorising step
``datasets.py:147,158-160``.  The three librosa==0.8.1 calls the reference makes
(``librosa.core.stft`` datasets.py:255, ``librosa.filters.mel`` :203,
``librosa.power_to_db`` :265) are NOT available in /root/reference nor in this
image; their published semantics are restated here:

  * stft: center=True reflect padding of n_fft//2, periodic Hann window
    (scipy.signal.get_window('hann', n_fft, fftbins=True)), hop 600, frame k
    starts at 600*k of the padded signal, np.fft.rfft, no normalisation.
  * mel: Slaney scale, fmin=0, fmax=sr/2, Slaney area normalisation, float32.
  * power_to_db: 10*log10(max(S, 1e-10)), then max(., global_max - 80).

PARITY UNPINNED at the librosa boundary (no golden vector exists in the
reference for it); cross-checked in tests against torch.stft and
transformers.audio_utils.mel_filter_bank.
"""
import numpy as np

SR = 24000
N_FFT = 1200
HOP = 600
N_MELS = 64
N_BINS = N_FFT // 2 + 1
EPS = 1e-8          # datasets.py:204


def hann_periodic(n=N_FFT):
    """scipy.signal.get_window('hann', n, fftbins=True) == 0.5 - 0.5 cos(2 pi i / n)."""
    i = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * i / n)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mel)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS):
    """librosa.filters.mel(sr, n_fft, n_mels).T  (datasets.py:203) -> (n_bins, n_mels) float32."""
    fftfreqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_pts = np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2)
    mel_f = _mel_to_hz(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return np.ascontiguousarray(w.astype(np.float32).T)


def stft_all_frames(audio, n_fft=N_FFT, hop=HOP):
    """What ``librosa.core.stft`` hands back at datasets.py:255 (before the ``[:, :nb_feature_frames]`` cut at :257): all
    1 + N // hop centred frames, (T + 1, n_bins, C)."""
    return stft(audio, n_fft, hop, extra_frame=True)


def stft(audio, n_fft=N_FFT, hop=HOP, extra_frame=False):
    """datasets.py:252-258.  audio (N, C) float64 -> (T, n_bins, C) complex128, T=int(N/hop)."""
    audio = np.asarray(audio, dtype=np.float64)
    n, c = audio.shape
    t = int(n / float(hop)) + (1 if extra_frame else 0)
    win = hann_periodic(n_fft)
    out = np.empty((t, n_fft // 2 + 1, c), dtype=np.complex128)
    for ch in range(c):
        y = np.pad(audio[:, ch], n_fft // 2, mode="reflect")
        idx = (np.arange(t) * hop)[:, None] + np.arange(n_fft)[None, :]
        out[:, :, ch] = np.fft.rfft(y[idx] * win[None, :], axis=1)
    return out


def power_to_db(s, amin=1e-10, top_db=80.0):
    ls = 10.0 * np.log10(np.maximum(amin, s))
    return np.maximum(ls, ls.max() - top_db)


def logmel(spec, mel_wts):
    """datasets.py:260-267 -> (T, n_mels, C) float64; top_db clip per channel over the whole clip."""
    t, _, c = spec.shape
    out = np.zeros((t, mel_wts.shape[1], c))
    for ch in range(c):
        mag = np.abs(spec[:, :, ch]) ** 2
        out[:, :, ch] = power_to_db(np.dot(mag, mel_wts))
    return out


def foa_intensity(spec, mel_wts):
    """datasets.py:269-279 -> (T, n_mels, 3) float64."""
    w = spec[:, :, 0]
    inten = np.real(np.conj(w)[:, :, None] * spec[:, :, 1:])
    energy = EPS + (np.abs(w) ** 2 + (np.abs(spec[:, :, 1:]) ** 2).sum(-1) / 3.0)
    inorm = inten / energy[:, :, None]
    return np.einsum("tfc,fm->tmc", inorm, mel_wts.astype(np.float64))


def unit_scaler():
    return {"MEL": {"mean": np.zeros((1, N_MELS, 4)), "std": np.ones((1, N_MELS, 4))},
            "IV": {"mean": np.zeros((1, N_MELS, 3)), "std": np.ones((1, N_MELS, 3))}}


def synthetic_spectrum(seed, t):
    """Seeded complex spectrum (T + 1, 601, 4) for the spectrum-injection fixture (tests/golden/features_ref.npz, made by
    running the reference's own ``get_feature`` on it): Y / Z / X partly coherent with W so the intensity vector is not
    noise; magnitudes span 100 dB and one frame is near-silent, so the 1e-8 of E and power_to_db's floor / clip matter."""
    rng = np.random.default_rng(int(seed))
    shape = (int(t) + 1, N_BINS)
    mag = 10.0 ** rng.uniform(-4.0, 1.0, size=shape)
    w = mag * np.exp(1j * rng.uniform(-np.pi, np.pi, size=shape))
    spec = np.empty(shape + (4,), dtype=np.complex128)
    spec[:, :, 0] = w
    for c in range(1, 4):
        coh = rng.uniform(-1.0, 1.0, size=shape)
        noise = 0.3 * mag * (rng.normal(size=shape) + 1j * rng.normal(size=shape))
        spec[:, :, c] = coh * w + noise
    spec[3, :, :] *= 1e-7
    return spec


def features_from_spectrum(spec, scaler=None, mel_wts=None):
    """datasets.py:287-290 on a given (T, 601, 4) spectrum -> z-scored (MEL (T,64,4), IV (T,64,3)) float64."""
    if mel_wts is None:
        mel_wts = mel_filterbank()
    if scaler is None:
        scaler = unit_scaler()
    mel = (logmel(spec, mel_wts) - scaler["MEL"]["mean"]) / scaler["MEL"]["std"]
    iv = (foa_intensity(spec, mel_wts) - scaler["IV"]["mean"]) / scaler["IV"]["std"]
    return mel, iv


def get_feature(audio, scaler=None, mel_wts=None):
    """datasets.py:281-292 + :158-160.

    audio: (N, 4) float64, already ``int16/32768.0 + 1e-8`` (datasets.py:147).
    Returns float32 (7, T, 64): channels [mel W,Y,Z,X ; IV y,z,x], and nb_label_frames.
    """
    if mel_wts is None:
        mel_wts = mel_filterbank()
    if scaler is None:
        scaler = unit_scaler()
    spec = stft(audio)
    mel = logmel(spec, mel_wts)
    iv = foa_intensity(spec, mel_wts)
    mel = (mel - scaler["MEL"]["mean"]) / scaler["MEL"]["std"]
    iv = (iv - scaler["IV"]["mean"]) / scaler["IV"]["std"]
    feat = np.concatenate([mel.transpose(2, 0, 1), iv.transpose(2, 0, 1)], axis=0)
    nb_label_frames = int(audio.shape[0] / float(int(SR * 0.1)))
    return feat.astype(np.float32), nb_label_frames


def gcc_phat(spec, n_lags=N_MELS):
    """MIC-format GCC-PHAT features, (T, 601, 4) -> (T, n_lags, 6).  NOT part of the reference (FOA is hard-coded there,
    src/datasets.py:36-37,55): PARITY UNPINNED.  Definition of the DCASE2022 SELD baseline the reference's README credits
    (README.md:156; cls_feature_class.py::_get_gcc): per microphone pair m < n, R = conj(X_m) X_n,
    cc = irfft(exp(i angle(R))) over n_fft samples, feature = concat(cc[-n_lags/2:], cc[:n_lags/2])."""
    t, _, c = spec.shape
    out = np.zeros((t, n_lags, c * (c - 1) // 2))
    p = 0
    for m in range(c):
        for n in range(m + 1, c):
            r = np.conj(spec[:, :, m]) * spec[:, :, n]
            cc = np.fft.irfft(np.exp(1j * np.angle(r)), n=N_FFT, axis=1)
            out[:, :, p] = np.concatenate([cc[:, -n_lags // 2:], cc[:, :n_lags // 2]], axis=-1)
            p += 1
    return out


def unit_scaler_mic():
    return {"MEL": {"mean": np.zeros((1, N_MELS, 4)), "std": np.ones((1, N_MELS, 4))},
            "GCC": {"mean": np.zeros((1, N_MELS, 6)), "std": np.ones((1, N_MELS, 6))}}


def get_feature_mic(audio, scaler=None, mel_wts=None):
    """MIC feature set of BASELINE config 5: log-mel of the four microphones (datasets.py:260-267 applied to MIC audio) +
    the six GCC-PHAT channels, z-scored -> float32 (10, T, 64).  Parity unpinned (see ``gcc_phat``)."""
    if mel_wts is None:
        mel_wts = mel_filterbank()
    if scaler is None:
        scaler = unit_scaler_mic()
    spec = stft(audio)
    mel = (logmel(spec, mel_wts) - scaler["MEL"]["mean"]) / scaler["MEL"]["std"]
    gcc = (gcc_phat(spec) - scaler["GCC"]["mean"]) / scaler["GCC"]["std"]
    feat = np.concatenate([mel.transpose(2, 0, 1), gcc.transpose(2, 0, 1)], axis=0)
    return feat.astype(np.float32), int(audio.shape[0] / float(int(SR * 0.1)))


def int16_to_audio(pcm):
    """datasets.py:147."""
    return np.asarray(pcm, dtype=np.float64) / 32768.0 + 1e-8
