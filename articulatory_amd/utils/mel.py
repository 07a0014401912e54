"""Mel filterbank of the reference's MelSpectrogram (articulatory/losses/mel_loss.py:56-62 calls librosa.filters.mel(sr, n_fft, n_mels,
fmin, fmax) with librosa's defaults htk=False, norm='slaney').  librosa is not a dependency here: this restates its published
algorithm (Slaney's Auditory Toolbox mel scale: linear below 1 kHz, logarithmic above; triangular filters scaled to unit area in Hz)."""
import numpy as np


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """(n_mels, 1 + n_fft // 2) float32."""
    n_freq = 1 + n_fft // 2
    freqs = np.linspace(0.0, sr / 2.0, n_freq)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    fb = np.zeros((n_mels, n_freq))
    for i in range(n_mels):
        fb[i] = np.maximum(0.0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    fb *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]
    return fb.astype(np.float32)
