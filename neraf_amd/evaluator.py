"""Evaluation metric chain of NeRAF: what `RAFEvaluator` / `SoundSpacesEvaluator`
(NeRAF_evaluator.py:111-262) and the helper metrics (NeRAF_helper.py:12-161) compute for a predicted room impulse response,
plus the two torchaudio pieces the reference leans on -- the complex `Spectrogram` (NeRAF_evaluator.py:127) and `GriffinLim`
(NeRAF_model.py:139,753-754) -- restated on torch.stft / torch.istft so that they run on the GPU (rocFFT) next to the field.

Same names, arguments and result keys as the reference so that its eval loop can import from here.  The metric kernels are
channel-vectorised numpy (the reference loops over channels) and are pinned to reference outputs by tests/golden/g5_helper.npz
(EDT, C50, envelope distance, SNR, magnitude distance, SpectralLoss; tests/test_evaluator.py).

PARITY UNPINNED for three pieces whose third-party implementations are absent here: `measure_rt60` restates
pyroomacoustics.experimental.measure_rt60 (Schroeder integration, time between the -5 dB and -(5 + decay_db) dB crossings,
extrapolated to 60 dB), `highpass_biquad` restates torchaudio.functional.highpass_biquad (RBJ high-pass, Q = 0.707, output
clamped to [-1, 1] as torchaudio.lfilter does), `GriffinLim` restates torchaudio.transforms.GriffinLim (momentum 0.99, 32
iterations, random initial phase).  They are guarded by analytic tests (exponential decays with known RT60, a sine through the
filter, spectrogram round trips)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- spectral loss (NeRAF_helper.py:12-47) -------------------------------------------------------------------
class SpectralLoss(nn.Module):
    """base_loss between (log-)spectrograms: log(epsilon + |STFT|) for 'mag' inputs (10 log10 with dB=True), identity for
    'log mag' inputs."""

    def __init__(self, base_loss=F.mse_loss, reduction: str = "mean", epsilon: float = 1, dB: bool = False,
                 stft_input_type: str = "mag", **kwargs):
        super().__init__()
        if stft_input_type not in ("mag", "log mag"):
            raise ValueError("stft_input_type must be 'mag' or 'log mag'")
        self.base_loss, self.reduction, self.epsilon, self.dB, self.stft_input_type = base_loss, reduction, epsilon, dB, stft_input_type

    def _log_spectrogram(self, s: torch.Tensor) -> torch.Tensor:
        if self.stft_input_type == "log mag":
            return s
        return 10 * torch.log10(self.epsilon + s) if self.dB else torch.log(self.epsilon + s)

    def forward(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return self.base_loss(self._log_spectrogram(a), self._log_spectrogram(b), reduction=self.reduction)


# ---- energy-decay metrics ------------------------------------------------------------------------------------
def _schroeder_db(h: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Backward-integrated energy of each channel of h [C, T] in dB relative to its first sample, and the number of samples
    kept per channel (the all-zero tail is cut one sample before the last positive energy, as the reference's
    `energy[:i_nz]` does).  Entries beyond that length are +inf dB... i.e. never below any threshold."""
    h = np.atleast_2d(np.asarray(h, dtype=np.float64))
    energy = np.cumsum((h ** 2)[:, ::-1], axis=1)[:, ::-1]
    pos = energy > 0
    n_keep = np.where(pos.any(axis=1), pos.shape[1] - 1 - np.argmax(pos[:, ::-1], axis=1), 0)      # = max index with energy > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        db = 10.0 * np.log10(energy)
    db = db - db[:, :1]
    db[np.arange(h.shape[1])[None, :] >= n_keep[:, None]] = np.inf
    return db, n_keep


def _first_below(db: np.ndarray, level: float) -> np.ndarray:
    """Index of the first sample whose decay curve is strictly below `level` dB, per channel (ValueError if none, like np.min of
    an empty np.where in the reference)."""
    hit = (level - db) > 0
    if not hit.any(axis=1).all():
        raise ValueError("decay curve never reaches %g dB" % level)
    return np.argmax(hit, axis=1)


def measure_edt(h, fs: float = 44100, decay_db: float = 10):
    """Early decay time (NeRAF_helper.py:115-139): time to fall `decay_db` below the start of the Schroeder curve, scaled to 60 dB."""
    h = np.asarray(h, dtype=np.float64)
    if not np.any(h ** 2 > 0):
        return np.nan
    db, _ = _schroeder_db(h)
    return float((60.0 / decay_db) * _first_below(db, -decay_db)[0] / float(fs))


def evaluate_edt(pred_ir, gt_ir, fs):
    """(gt, pred) EDT per channel, in the reference's return order (NeRAF_helper.py:141-155)."""
    pred_ir, gt_ir = np.atleast_2d(pred_ir), np.atleast_2d(gt_ir)
    return (np.array([measure_edt(c, fs=fs) for c in gt_ir]), np.array([measure_edt(c, fs=fs) for c in pred_ir]))


def measure_clarity(signal, time: float = 50, fs: float = 44100) -> float:
    """C_time: early-to-late energy ratio in dB, split after int(time/1000*fs + 1) samples (NeRAF_helper.py:94-97)."""
    h2 = np.asarray(signal, dtype=np.float64) ** 2
    t = int((time / 1000) * fs + 1)
    return float(10 * np.log10(np.sum(h2[:t]) / np.sum(h2[t:])))


def evaluate_clarity(pred_ir, gt_ir, fs):
    pred_ir, gt_ir = np.atleast_2d(pred_ir), np.atleast_2d(gt_ir)
    return (np.array([measure_clarity(c, fs=fs) for c in gt_ir]), np.array([measure_clarity(c, fs=fs) for c in pred_ir]))


def measure_rt60(h, fs: float = 1, decay_db: float = 60) -> float:
    """Reverberation time from the Schroeder curve: (60 / decay_db) x the time between the -5 dB and -(5 + decay_db) dB crossings.
    [pyroomacoustics.experimental.measure_rt60, restated; parity unpinned]"""
    db, _ = _schroeder_db(np.asarray(h, dtype=np.float64))
    t5 = _first_below(db, -5.0)[0] / float(fs)
    td = _first_below(db, -5.0 - decay_db)[0] / float(fs)
    return float((60.0 / decay_db) * (td - t5))


def highpass_biquad(x: np.ndarray, sample_rate: float, cutoff_freq: float, Q: float = 0.707) -> np.ndarray:
    """RBJ high-pass biquad, direct form, output clamped to [-1, 1].  [torchaudio.functional.highpass_biquad, restated]"""
    from scipy.signal import lfilter
    w0 = 2.0 * np.pi * cutoff_freq / sample_rate
    alpha = np.sin(w0) / (2.0 * Q)
    b = np.array([(1 + np.cos(w0)) / 2, -(1 + np.cos(w0)), (1 + np.cos(w0)) / 2])
    a = np.array([1 + alpha, -2 * np.cos(w0), 1 - alpha])
    return np.clip(lfilter(b / a[0], a / a[0], np.asarray(x, dtype=np.float64)), -1.0, 1.0)


def measure_rt60_advance(signal, sr, decay_db: float = 10, cutoff_freq: float = 200) -> float:
    """RAF benchmark variant (NeRAF_helper.py:68-77): 200 Hz high-pass, then T60 from a 10 dB decay."""
    return measure_rt60(highpass_biquad(signal, sr, cutoff_freq), sr, decay_db=decay_db)


def compute_t60(true_in, gen_in, fs, advanced: bool = False):
    """(gt, pred) T60 per channel; -1 for a channel whose decay never reaches the thresholds (NeRAF_helper.py:49-66)."""
    gt, pred = [], []
    for t, g in zip(np.atleast_2d(true_in), np.atleast_2d(gen_in)):
        try:
            if advanced:
                a, b = measure_rt60_advance(t, sr=fs), measure_rt60_advance(g, sr=fs)
            else:
                a, b = measure_rt60(t, fs=fs, decay_db=30), measure_rt60(g, fs=fs, decay_db=30)
        except Exception:
            a, b = -1, -1
        gt.append(a); pred.append(b)
    return np.array(gt), np.array(pred)


# ---- waveform / magnitude distances (NeRAF_helper.py:79-92) --------------------------------------------------
def Envelope_distance(predicted, gt) -> float:
    """Sum over channels of the RMS difference of the Hilbert envelopes."""
    from scipy.signal import hilbert
    p, g = np.atleast_2d(predicted), np.atleast_2d(gt)
    return float(np.sqrt(np.mean((np.abs(hilbert(g, axis=1)) - np.abs(hilbert(p, axis=1))) ** 2, axis=1)).sum())


def SNR(predicted, gt) -> float:
    predicted, gt = np.asarray(predicted), np.asarray(gt)
    return float(10.0 * np.log10((np.mean(gt ** 2) + 1e-4) / (np.mean((predicted - gt) ** 2) + 1e-4)))


def Magnitude_distance(predicted_mag, gt_mag) -> float:
    p, g = np.asarray(predicted_mag), np.asarray(gt_mag)
    return float(np.mean((p - g).reshape(p.shape[0], -1) ** 2, axis=1).sum())


# ---- STFT pieces ---------------------------------------------------------------------------------------------
def spectrogram(wave: torch.Tensor, n_fft: int, win_length: int, hop_length: int) -> torch.Tensor:
    """Complex STFT [..., n_fft/2+1, frames]: Hann window of win_length centred in n_fft, reflect padding, no normalisation.
    [torchaudio.transforms.Spectrogram(power=None), restated]"""
    win = torch.hann_window(win_length, periodic=True, dtype=wave.dtype if wave.is_floating_point() else torch.float32,
                            device=wave.device)
    shp = wave.shape
    out = torch.stft(wave.reshape(-1, shp[-1]), n_fft, hop_length=hop_length, win_length=win_length, window=win, center=True,
                     pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return out.reshape(*shp[:-1], *out.shape[-2:])


class GriffinLim(nn.Module):
    """Phase reconstruction from a magnitude (power = 1) spectrogram [..., F, T] -> waveform [..., (T - 1) * hop].
    Fast Griffin-Lim with momentum 0.99, 32 iterations, random initial phase (seedable through `generator`).
    [torchaudio.transforms.GriffinLim, restated; parity unpinned]"""

    def __init__(self, n_fft: int, win_length: Optional[int] = None, hop_length: Optional[int] = None, power: float = 1.0,
                 n_iter: int = 32, momentum: float = 0.99, rand_init: bool = True):
        super().__init__()
        self.n_fft = n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length or self.win_length // 2
        self.power, self.n_iter, self.momentum, self.rand_init = power, n_iter, momentum, rand_init

    def initial_phase(self, shape, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """The uniform [0, 1) draw ``forward`` makes for a spectrogram of ``shape`` (flattened to [-1, F, T]) -- exposed so that a caller
        batching several spectrograms into ONE reconstruction can draw them in the order separate calls would have."""
        shape = (int(np.prod(shape[:-2])) if len(shape) > 2 else 1, shape[-2], shape[-1])
        device = torch.device(device)
        return torch.rand(shape, generator=generator, device=device if generator is None or generator.device == device else "cpu").to(device)

    def forward(self, specgram: torch.Tensor, generator: Optional[torch.Generator] = None, init_phase: Optional[torch.Tensor] = None) -> torch.Tensor:
        shp = specgram.shape
        mag = specgram.reshape(-1, shp[-2], shp[-1]).float()
        if self.power != 1.0:
            mag = mag.pow(1.0 / self.power)
        win = torch.hann_window(self.win_length, periodic=True, dtype=torch.float32, device=mag.device)
        length = (shp[-1] - 1) * self.hop_length
        if self.rand_init:
            ang = init_phase.reshape(mag.shape).to(mag.device) if init_phase is not None else self.initial_phase(mag.shape, mag.device, generator)
            angles = torch.polar(torch.ones_like(mag), 2 * np.pi * ang)
        else:
            angles = torch.ones_like(mag, dtype=torch.complex64)
        m = self.momentum / (1 + self.momentum)
        prev = torch.zeros_like(angles)
        kw = dict(hop_length=self.hop_length, win_length=self.win_length, window=win, center=True)
        for _ in range(self.n_iter):
            wav = torch.istft(mag * angles, self.n_fft, length=length, **kw)
            rebuilt = torch.stft(wav, self.n_fft, pad_mode="reflect", normalized=False, onesided=True, return_complex=True, **kw)
            angles = rebuilt - prev * m if self.momentum else rebuilt
            angles = angles / (angles.abs() + 1e-16)
            prev = rebuilt
        wav = torch.istft(mag * angles, self.n_fft, length=length, **kw)
        return wav.reshape(*shp[:-2], wav.shape[-1])


# ---- evaluators (NeRAF_evaluator.py:111-262) -----------------------------------------------------------------
def _pad_to(w: np.ndarray, n: int) -> np.ndarray:
    return np.pad(w, ((0, 0), (0, n - w.shape[1])), "constant")


def _room_metrics(wav_prd: np.ndarray, wav_gt_ff: np.ndarray, fs: float, advanced: bool) -> Tuple[float, int, float, float]:
    """T60 error in percent (100 % for an instance with an invalid channel), #invalid, mean |EDT error|, mean |C50 error|."""
    t_gt, t_pr = compute_t60(wav_gt_ff, wav_prd, fs=fs, advanced=advanced)
    invalid = bool(np.any(np.concatenate([t_gt, t_pr]) < -0.5))
    with np.errstate(divide="ignore", invalid="ignore"):
        t60_err = 1.0 if invalid else float(np.mean(np.abs(t_pr - t_gt) / np.abs(t_gt)))
    # the reference passes (pred, gt_ff) to helpers that return (gt, pred) of their (pred_ir, gt_ir) arguments: either way the
    # absolute difference is the same
    e_a, e_b = evaluate_edt(wav_prd, wav_gt_ff, fs=fs)
    c_a, c_b = evaluate_clarity(wav_prd, wav_gt_ff, fs=fs)
    return t60_err * 100.0, int(invalid), float(np.mean(np.abs(e_b - e_a))), float(np.mean(np.abs(c_b - c_a)))


class RAFEvaluator:
    def __init__(self, fs: int = 48000):
        self.fs = fs
        self.spectral_loss_mag = SpectralLoss(base_loss=F.l1_loss, reduction="mean", epsilon=1, dB=False, stft_input_type="mag")
        self.spectral_loss_logmag = SpectralLoss(base_loss=F.l1_loss, reduction="mean", epsilon=1, dB=False, stft_input_type="log mag")
        if fs == 48000:
            self.n_fft, self.win_length, self.hop_len = 1024, 512, 256
        elif fs == 16000:
            self.n_fft, self.win_length, self.hop_len = 512, 256, 128
        else:
            raise ValueError("Sample rate not supported")

    def transform_stft_torch(self, wave: torch.Tensor) -> torch.Tensor:
        return spectrogram(wave, self.n_fft, self.win_length, self.hop_len)

    def get_full_metrics(self, mag_prd, mag_gt, wav_gt_ff, wav_pred_istft, wav_gt_istft, log_prd, log_gt) -> Dict[str, float]:
        wav_prd = _pad_to(np.asarray(wav_pred_istft), wav_gt_ff.shape[1])
        # RAF's spectral error: waveform -> STFT again, log(|.| + 1e-3), L1 against the ground-truth log-magnitude
        back = self.transform_stft_torch(torch.as_tensor(wav_prd))
        log_back = torch.log(back.abs() + 1e-3)[..., :log_gt.shape[2]]
        raf_spectral = self.spectral_loss_logmag(log_back.to(torch.as_tensor(log_gt).dtype), torch.as_tensor(log_gt))
        t60, invalid, edt, c50 = _room_metrics(wav_prd, np.asarray(wav_gt_ff), self.fs, advanced=True)
        return {"audio_T60": float(t60), "audio_total_invalids_T60": float(invalid), "audio_stft_error": float(raf_spectral),
                "audio_EDT": float(edt), "audio_C50": float(c50)}

    def get_stft_metrics(self, mag_prd, mag_gt, gl: bool = False):
        return {"audio_mag": torch.mean(torch.pow(mag_prd - mag_gt, 2)) * 2,
                "audio_spectral_loss": self.spectral_loss_mag(mag_prd, mag_gt).item()}


class SoundSpacesEvaluator:
    def __init__(self, fs: int = 22050):
        self.fs = fs

    def get_full_metrics(self, mag_prd, mag_gt, wav_gt_ff, wav_pred_istft, wav_gt_istft, log_prd, log_gt) -> Dict[str, float]:
        wav_prd = _pad_to(np.asarray(wav_pred_istft), wav_gt_ff.shape[1])
        t60, invalid, edt, c50 = _room_metrics(wav_prd, np.asarray(wav_gt_ff), self.fs, advanced=False)
        return {"audio_T60_mean_error": float(t60), "audio_total_invalids_T60": float(invalid), "audio_EDT": float(edt),
                "audio_C50": float(c50)}

    def get_stft_metrics(self, mag_prd, mag_gt):
        return {"audio_mag": torch.mean(torch.pow(mag_prd - mag_gt, 2)) * 2}
