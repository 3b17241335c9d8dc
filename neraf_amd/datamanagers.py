"""Data-manager objects with the surface ``NeRAFPipeline`` reads from nerfstudio's / the reference's managers
(NeRAF_pipeline.py:104-149, :175, :187, :243, :276, :310-325, :360): ``train_dataset`` / ``eval_dataset`` (with ``scene_box``,
``metadata``, ``__len__``), ``next_train`` / ``next_eval`` / ``next_eval_image``, ``fixed_indices_eval_dataloader``,
``train_num_rays_per_batch``, ``get_param_groups``, ``to``.

File decoding, COLMAP / transforms.json parsing and the 16-worker loaders of NeRAF_datamanager.py / NeRAF_dataset.py are out of
scope: these managers serve DEVICE-RESIDENT data -- synthetic (tests, bench) or handed in by the caller
-- and never touch the host inside a step."""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional

import numpy as np
import torch

from . import synth
from .cameras import Cameras
from .config import SceneBox
from .vision import RayBundle


class FixedBatchDataManager:
    """``next_train`` returns the same resident (ray_bundle, batch) pair every step."""

    def __init__(self, ray_bundle, batch: Dict[str, torch.Tensor], train_num_rays_per_batch: Optional[int] = None):
        self.ray_bundle, self.batch = ray_bundle, batch
        if train_num_rays_per_batch is None:
            train_num_rays_per_batch = len(ray_bundle) if ray_bundle is not None else 4096
        self.train_num_rays_per_batch = train_num_rays_per_batch

    def next_train(self, step: int):
        return self.ray_bundle, self.batch

    next_eval = next_train

    def get_param_groups(self):
        return {}


class RotatingBatchDataManager:
    """``next_train`` cycles through K distinct resident (ray_bundle, batch) pairs: every step gathers other hash-table rows and
    scatters into other gradient rows, as a training run that draws a fresh batch per step does (NeRAF_pipeline.py:175, :187),
    without any host work inside the step.  ``pin(i)`` freezes the rotation on pair i (the fixed-batch A/B of bench.py)."""

    def __init__(self, ray_bundles, batches, train_num_rays_per_batch: Optional[int] = None):
        if len(ray_bundles) != len(batches) or not batches:
            raise ValueError("need as many ray bundles (or None entries) as batches, at least one")
        self.ray_bundles, self.batches = list(ray_bundles), list(batches)
        if train_num_rays_per_batch is None:
            train_num_rays_per_batch = len(ray_bundles[0]) if ray_bundles[0] is not None else 4096
        self.train_num_rays_per_batch = train_num_rays_per_batch
        self._pinned: Optional[int] = None
        self._calls = 0

    def __len__(self):
        return len(self.batches)

    def pin(self, i: Optional[int]):
        self._pinned = i

    def next_train(self, step: int):
        i = self._pinned if self._pinned is not None else self._calls % len(self.batches)
        self._calls += 1
        return self.ray_bundles[i], self.batches[i]

    next_eval = next_train

    def get_param_groups(self):
        return {}


class RIRBankDataManager:
    """Audio batches sampled on the device from a ``DeviceRIRBank`` (neraf_amd/data.py); the ray bundle slot is None as in
    NeRAFDataManager.next_train (the audio model needs no rays, NeRAF_pipeline.py:187)."""

    def __init__(self, bank, batch_size: int = 2048, generator: Optional[torch.Generator] = None):
        self.bank, self.batch_size, self.generator = bank, batch_size, generator

    def next_train(self, step: int):
        return None, self.bank.next_train(self.batch_size, generator=self.generator)

    next_eval = next_train

    def get_param_groups(self):
        return {}


# ---- synthetic scene: cameras + images ------------------------------------------------------------------------------------------
def _look_at(eye: np.ndarray, target: np.ndarray) -> np.ndarray:
    """camera_to_world [3,4], OpenGL convention (camera looks along -z, +y up)."""
    fwd = target - eye
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right = right / max(np.linalg.norm(right), 1e-9)
    up2 = np.cross(right, fwd)
    return np.concatenate([np.stack([right, up2, -fwd], axis=1), eye[:, None]], axis=1).astype(np.float32)


def synthetic_cameras(n: int, width: int = 684, height: int = 1024, tag: str = "cams", distortion: bool = True) -> Cameras:
    """n cameras with the RAF intrinsics (data/RAF/FurnishedRoom/transforms.json: 684 x 1024, fl 350.41, OPENCV distortion) on a
    ring inside the unit scene, looking at jittered targets near the centre."""
    ang = synth.uniform(tag + ".ang", (n,), 0.0, 2 * math.pi).astype(np.float64)
    rad = synth.uniform(tag + ".rad", (n,), 0.25, 0.6).astype(np.float64)
    hz = synth.uniform(tag + ".h", (n,), -0.2, 0.2).astype(np.float64)
    tgt = synth.uniform(tag + ".tgt", (n, 3), -0.15, 0.15).astype(np.float64)
    c2w = np.stack([_look_at(np.array([rad[i] * math.cos(ang[i]), rad[i] * math.sin(ang[i]), hz[i]]), tgt[i]) for i in range(n)])
    sx, sy = width / 684.0, height / 1024.0
    dist = torch.tensor([-0.03364928460212668, 0.008088314216939308, -0.00032575844569464275, 0.0,
                         0.00013248765042335402, -0.00043094049868545956]) if distortion else None
    return Cameras(torch.from_numpy(c2w), 350.41100113602533 * sx, 350.41100113602533 * sy, 342.5683121123288 * sx,
                   511.3341668407641 * sy, width, height, dist)


def _global_rank(local_rank: int) -> int:
    """The GLOBAL rank seeds per-rank generators (on a multi-node job local ranks repeat and would duplicate batches)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return int(dist.get_rank())
    except Exception:
        pass
    return int(local_rank)


class _VisionDataset:
    def __init__(self, cameras: Cameras, images: torch.Tensor, scene_box: SceneBox):
        self.cameras, self.images, self.scene_box, self.metadata = cameras, images, scene_box, {}

    def __len__(self):
        return self.cameras.size


class SyntheticVisionDataManager:
    """Pixel-sampling vision manager over device-resident images [N,H,W,3]: ``next_train`` draws ``train_num_rays_per_batch``
    random (camera, row, col) triples and returns (RayBundle, {"image": rgb [R,3], "indices": [R,3]}) like nerfstudio's
    ParallelDataManager (NeRAF_config.py:83-91); ``fixed_indices_eval_dataloader`` yields (camera, {"image": [H,W,3]})."""

    def __init__(self, n_train: int = 8, n_eval: int = 2, width: int = 64, height: int = 96, train_num_rays_per_batch: int = 4096,
                 device="cpu", seed: int = 0, world_size: int = 1, local_rank: int = 0):
        box = SceneBox(torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]))

        def images(n, tag):
            return torch.from_numpy(synth.uniform(tag, (n, height, width, 3), 0.0, 1.0))
        self.train_dataset = _VisionDataset(synthetic_cameras(n_train, width, height, "dm.train"), images(n_train, "dm.train.img"), box)
        self.eval_dataset = _VisionDataset(synthetic_cameras(n_eval, width, height, "dm.eval"), images(n_eval, "dm.eval.img"), box)
        self.train_num_rays_per_batch = train_num_rays_per_batch
        self.eval_num_rays_per_batch = train_num_rays_per_batch
        self.device = torch.device(device)
        self.generator = torch.Generator(device="cpu").manual_seed(seed + 7919 * _global_rank(local_rank))
        self._eval_i = 0
        self.to(device)

    def to(self, device):
        self.device = torch.device(device)
        for ds in (self.train_dataset, self.eval_dataset):
            ds.cameras, ds.images = ds.cameras.to(device), ds.images.to(device)
        return self

    def _sample(self, ds: _VisionDataset, n: int):
        N, H, W = ds.images.shape[:3]
        idx = torch.stack([torch.randint(0, N, (n,), generator=self.generator), torch.randint(0, H, (n,), generator=self.generator),
                           torch.randint(0, W, (n,), generator=self.generator)], dim=-1).to(self.device)
        coords = idx[:, 1:].float() + 0.5
        rb = ds.cameras.generate_rays(idx[:, 0], coords)
        return rb, {"image": ds.images[idx[:, 0], idx[:, 1], idx[:, 2]], "indices": idx}

    def next_train(self, step: int):
        return self._sample(self.train_dataset, self.train_num_rays_per_batch)

    def next_eval(self, step: int):
        return self._sample(self.eval_dataset, self.eval_num_rays_per_batch)

    def next_eval_image(self, step: int):
        i = self._eval_i % len(self.eval_dataset)
        self._eval_i += 1
        return self.eval_dataset.cameras[i], {"image": self.eval_dataset.images[i]}

    @property
    def fixed_indices_eval_dataloader(self):
        return [(self.eval_dataset.cameras[i], {"image": self.eval_dataset.images[i]}) for i in range(len(self.eval_dataset))]

    def get_param_groups(self):
        return {}


# ---- synthetic RIR set ------------------------------------------------------------------------------------------------------------
class _AudioDataset:
    """Indexable like the reference's eval datasets: mode 'eval' -> one time slice per index (NeRAF_dataset.py:129-130), mode
    'eval_image' -> one whole RIR per index with 'data' [C,F,T] and 'waveform' [C,n] (:180-181, :352-353)."""

    def __init__(self, bank, waveforms: torch.Tensor, scene_box: SceneBox):
        self.bank, self.waveforms, self.scene_box, self.mode = bank, waveforms, scene_box, "eval"

    def __len__(self):
        return self.bank.n_rir if self.mode in ("eval_image", "inference") else len(self.bank)

    def __getitem__(self, i: int):
        if self.mode in ("eval_image", "inference"):
            d = self.bank.get_data_eval(int(i))
            d["waveform"] = self.waveforms[int(i)]
            return d
        return self.bank.get_data(int(i))


class _BankDataManager:
    """Common surface of the audio managers (``NeRAFDataManager``, NeRAF_datamanager.py:43-143): ``next_train`` / ``next_eval`` serve
    batches of time slices sampled on the device from a DeviceRIRBank, ``next_eval_image`` whole RIRs in turn."""

    train_dataset: _AudioDataset
    eval_dataset: _AudioDataset

    def _finish(self, batch_size: int, device, seed: int = 0, world_size: int = 1, local_rank: int = 0):
        self.batch_size = batch_size
        self.world_size, self.local_rank, self.seed = world_size, local_rank, seed
        self.generator = None
        self._eval_i = 0
        self.to(device)

    def _rank_generator(self, device):
        """Data parallel: every rank draws its OWN slices (a per-rank device generator, seed + 7919 * global rank) -- ranks that seeded
        torch identically for identical initial weights would otherwise all train on the same audio batch while the STFT loss is
        summed over the ranks as if the batches were distinct.  Single process: torch's default generator, as before."""
        if self.world_size <= 1:
            return None
        rank = _global_rank(self.local_rank)
        dev = torch.device(device)
        g = torch.Generator(device=dev if dev.type == "cuda" else "cpu")
        g.manual_seed(int(self.seed) + 7919 * int(rank))
        return g

    def to(self, device):
        for ds in (self.train_dataset, self.eval_dataset):
            b = ds.bank
            b.log_mag, b.mic_pose, b.source_pose, b.rot = b.log_mag.to(device), b.mic_pose.to(device), b.source_pose.to(device), b.rot.to(device)
            ds.waveforms = ds.waveforms.to(device)
        if getattr(self, "world_size", 1) > 1:
            # the per-rank generator is created ONCE per device: 'cuda' and 'cuda:0' name the same device, and re-creating it on
            # every .to() would re-seed it and restart the rank's audio batch sequence
            dev = torch.device(device)
            if dev.type == "cuda" and dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            if self.generator is None or self.generator.device != dev:
                self.generator = self._rank_generator(dev)
        return self

    def next_train(self, step: int):
        return None, self.train_dataset.bank.next_train(self.batch_size, generator=self.generator)

    def next_eval(self, step: int):
        return None, self.eval_dataset.bank.next_train(self.batch_size, generator=self.generator)

    def next_eval_image(self, step: int):
        i = self._eval_i % self.eval_dataset.bank.n_rir
        self._eval_i += 1
        d = self.eval_dataset.bank.get_data_eval(i)
        d["waveform"] = self.eval_dataset.waveforms[i]
        return None, d

    def get_param_groups(self):
        return {}


class SyntheticAudioDataManager(_BankDataManager):
    """Exponentially decaying noise RIRs in a RAF-like room, tokenised once into a DeviceRIRBank (neraf_amd/data.py)."""

    def __init__(self, n_train: int = 6, n_eval: int = 2, dataset: str = "RAF", batch_size: int = 2048, device="cpu", seed: int = 0,
                 world_size: int = 1, local_rank: int = 0):
        from .data import DeviceRIRBank
        fs, max_len, hop = (48000, 60, 256) if dataset == "RAF" else (16000, 60, 128)
        n = hop * (max_len - 1)
        aabb = torch.from_numpy(synth.audio_aabb())
        box = SceneBox(aabb)

        def make(nr, tag):
            t = np.arange(n) / fs
            tau = synth.uniform(tag + ".tau", (nr, 1), 0.03, 0.08).astype(np.float64)
            w = synth.normal(tag + ".wave", (nr, n)).astype(np.float64) * np.exp(-t[None, :] / tau)
            lo, hi = np.array([-3.0, -1.5, -4.0]), np.array([3.0, 1.5, 4.0])
            mic = synth.uniform(tag + ".mic", (nr, 3), 0, 1, np.float64) * (hi - lo) + lo
            src = synth.uniform(tag + ".src", (nr, 3), 0, 1, np.float64) * (hi - lo) + lo
            ang = np.deg2rad(synth.integers(tag + ".rot", (nr,), 0, 360).astype(np.float64))
            rot = (np.stack([np.cos(ang), np.zeros_like(ang), np.sin(ang)], -1) + 1.0) / 2.0
            waves = torch.from_numpy(w.astype(np.float32))
            bank = DeviceRIRBank.from_waveforms(waves, fs, max_len, torch.from_numpy(mic), torch.from_numpy(src), torch.from_numpy(rot))
            return _AudioDataset(bank, waves[:, None, :], box)
        self.train_dataset, self.eval_dataset = make(n_train, "adm.train"), make(n_eval, "adm.eval")
        self._finish(batch_size, device, seed=seed, world_size=world_size, local_rank=local_rank)


class DiskAudioDataManager(_BankDataManager):
    """``RAFDataManager`` / ``SoundSpacesDataManager`` (NeRAF_datamanager.py:169-251, :270-356) over the datasets' on-disk formats,
    without the DataLoader: both splits are parsed and tokenised ONCE into device-resident banks (neraf_amd/dataparsers.py), and a
    training batch is an index computation plus four gathers on the GPU.  ``max_len`` as in the reference's configs: seconds for RAF
    (0.32 -> 60 frames of 256 samples at 48 kHz, NeRAF_datamanager.py:213-214), frames for SoundSpaces (NeRAF_config.py:43).
    The train split's scene box is the audio model's AABB (NeRAF_pipeline.py:135-139).  Ground-truth waveforms for the eval metrics
    are kept for RAF (the decoded, cropped signal, NeRAF_dataset.py:172-173) and for SoundSpaces (``binaural_rirs/<name>.wav``,
    44.1 -> 22.05 kHz, :326-349); the sample-rate conversions use scipy's polyphase resampler in librosa's place
    (``dataparsers.resample``: tolerance-level parity).  ``fs``: RAF 48000 (default) or 16000; SoundSpaces 22050."""

    def __init__(self, data: str, dataset: str = "RAF", max_len: float = None, batch_size: int = 2048, device="cpu",
                 eval_split: str = "test", world_size: int = 1, local_rank: int = 0, fs: int = None, seed: int = 0):
        from .dataparsers import bank_from_raf, bank_from_soundspaces, load_raf_rir, load_soundspaces_waveform
        import os
        if dataset == "RAF":
            fs = 48000 if fs is None else int(fs)
            hop = 256 if fs == 48000 else 128
            seconds = 0.32 if max_len is None else float(max_len)
            frames, n_time = int(seconds * fs) // hop, int(seconds * fs)                 # NeRAF_datamanager.py:213-214

            def make(split):
                bank, out = bank_from_raf(data, split, fs=fs, max_len=frames, max_len_samples=n_time)
                waves = torch.zeros((len(out.audios_filenames), 1, n_time))
                for i, name in enumerate(out.audios_filenames):
                    w = torch.from_numpy(load_raf_rir(os.path.join(data, "data", name, "rir.wav"), fs)[:n_time])
                    waves[i, 0, :w.shape[0]] = w
                return _AudioDataset(bank, waves, out.scene_box), out
        elif dataset == "SoundSpaces":
            fs = 22050 if fs is None else int(fs)
            frames = 76 if max_len is None else int(max_len)
            n_time = frames * 128                                                         # NeRAF_datamanager.py:311 (hop 128)

            def make(split):
                bank, out = bank_from_soundspaces(data, split, max_len=frames)
                waves = torch.zeros((len(out.audios_filenames), 2, n_time))
                missing = []
                for i, name in enumerate(out.audios_filenames):
                    path = os.path.join(data, "binaural_rirs", name + ".wav")
                    if os.path.exists(path):
                        waves[i] = torch.from_numpy(load_soundspaces_waveform(path, fs, n_time))
                    else:
                        missing.append(name)
                if missing:
                    # training splits may ship without the wavs (nothing reads them there); an EVAL split without them would have its
                    # T60 / EDT / C50 computed against silence
                    if split != "train":
                        raise FileNotFoundError(f"SoundSpaces split '{split}': {len(missing)} ground-truth waveform(s) missing under "
                                                f"{os.path.join(data, 'binaural_rirs')} (first: {missing[0]}.wav)")
                    warnings.warn(f"SoundSpaces split 'train': {len(missing)} of {len(out.audios_filenames)} binaural_rirs wavs are "
                                  "missing; their waveforms are zeros (not read in training)")
                return _AudioDataset(bank, waves, out.scene_box), out
        else:
            raise ValueError("dataset must be 'RAF' or 'SoundSpaces'")
        (self.train_dataset, self.train_dataparser_outputs), (self.eval_dataset, self.eval_dataparser_outputs) = make("train"), make(eval_split)
        self.max_len, self.fs = frames, fs
        self._finish(batch_size, device, seed=seed, world_size=world_size, local_rank=local_rank)
