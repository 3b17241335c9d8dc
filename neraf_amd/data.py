"""Device-resident audio data path.

The reference's `RAFDataset.get_data` (NeRAF_dataset.py:89-132) decodes a wav file and runs a full STFT for EVERY time slice it
serves, behind a 16-worker DataLoader; with a few-millisecond GPU step that loader is the bottleneck.  Here every RIR is
tokenised once -- complex STFT, log(|.| + 1e-3), laid out slice-major [N_rir, T, C, F] so that one training sample (one time
slice of one RIR, the reference's "ray") is a contiguous row -- and kept in HBM; a batch is one index computation and four
gathers on the device, no host work and no synchronisation.

Item semantics follow the reference: flat index i -> (rir = i // max_len, t = i % max_len) (NeRAF_dataset.py:85-86), fields
'audio_idx', 'data' [C, F], 'time_query', 'rot', 'mic_pose', 'source_pose' (:127-128).  Slices past the end of a short RIR hold
log(min|STFT| + 1e-3) (the reference's branch for that case, :112-115, calls torch.min on a complex tensor and cannot run; its
intent -- pad with the quietest value -- is what is implemented).  File decoding (librosa / the RAF folder layout) stays with the
caller: `from_waveforms` takes the decoded, resampled signals."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .evaluator import spectrogram


class DeviceRIRBank:
    def __init__(self, log_mag: torch.Tensor, mic_pose: torch.Tensor, source_pose: torch.Tensor, rot: torch.Tensor):
        """log_mag [N, T, C, F] float32; poses [N, 3] (kept in the dtype given: the reference feeds float64 poses)."""
        if log_mag.dim() != 4 or not (mic_pose.shape[0] == source_pose.shape[0] == rot.shape[0] == log_mag.shape[0]):
            raise ValueError("log_mag must be [N, T, C, F] with one pose triple per RIR")
        self.log_mag = log_mag.contiguous()
        self.mic_pose, self.source_pose, self.rot = mic_pose.contiguous(), source_pose.contiguous(), rot.contiguous()
        self.n_rir, self.max_len = int(log_mag.shape[0]), int(log_mag.shape[1])

    @classmethod
    def from_waveforms(cls, waves: torch.Tensor, fs: int, max_len: int, mic_pose, source_pose, rot, max_len_time: Optional[int] = None,
                       device=None) -> "DeviceRIRBank":
        """waves [N, n] (mono, RAF) or [N, C, n].  STFT parameters per sample rate as NeRAF_dataset.py:56-66."""
        if fs == 48000:
            n_fft, win, hop = 1024, 512, 256
        elif fs == 16000:
            n_fft, win, hop = 512, 256, 128
        else:
            raise ValueError("Sample rate not supported")
        w = waves if waves.dim() == 3 else waves.unsqueeze(1)
        if device is not None:
            w = w.to(device)
        if max_len_time is not None:
            w = w[..., :max_len_time]
        mag = spectrogram(w.float(), n_fft, win, hop).abs()                    # [N, C, F, frames]
        frames = mag.shape[-1]
        if frames >= max_len:
            mag = mag[..., :max_len]
        else:
            pad = mag.amin(dim=(1, 2, 3), keepdim=True).expand(-1, mag.shape[1], mag.shape[2], max_len - frames)
            mag = torch.cat([mag, pad], dim=-1)
        log_mag = torch.log(mag + 1e-3).permute(0, 3, 1, 2)                    # slice-major [N, T, C, F]
        dev = log_mag.device
        as_t = lambda p: torch.as_tensor(p).to(dev)
        return cls(log_mag, as_t(mic_pose), as_t(source_pose), as_t(rot))

    def __len__(self) -> int:
        return self.n_rir * self.max_len

    def get_id_tmp(self, idx: int):
        return idx // self.max_len, idx % self.max_len

    def get_data(self, audio_idx: int) -> Dict[str, object]:
        r, t = self.get_id_tmp(int(audio_idx))
        return {"audio_idx": r, "data": self.log_mag[r, t], "time_query": t, "rot": self.rot[r], "mic_pose": self.mic_pose[r],
                "source_pose": self.source_pose[r]}

    def get_data_eval(self, rir: int) -> Dict[str, torch.Tensor]:
        """Whole RIR for the eval branch: 'data' [C, F, T] as get_outputs_for_camera expects."""
        return {"audio_idx": rir, "data": self.log_mag[rir].permute(1, 2, 0), "rot": self.rot[rir], "mic_pose": self.mic_pose[rir],
                "source_pose": self.source_pose[rir]}

    def batch(self, idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Collated items for flat indices idx [B] (on the bank's device): what the DataLoader's default collate produces."""
        r = torch.div(idx, self.max_len, rounding_mode="floor")
        t = idx - r * self.max_len
        return {"audio_idx": r, "data": self.log_mag[r, t], "time_query": t, "rot": self.rot[r], "mic_pose": self.mic_pose[r],
                "source_pose": self.source_pose[r]}

    def next_train(self, batch_size: int, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """One shuffled training batch (uniform over all (RIR, slice) pairs, with replacement across batches like an infinite
        shuffled loader), produced entirely on the device."""
        idx = torch.randint(0, len(self), (batch_size,), device=self.log_mag.device, generator=generator)
        return self.batch(idx)
