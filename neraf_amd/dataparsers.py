"""On-disk formats of the two audio datasets, parsed ONCE into the device-resident RIR bank of neraf_amd/data.py.

The reference splits this over a dataparser (poses, rotations, scene box; NeRAF_dataparser.py), a Dataset that decodes one file per
served time slice (NeRAF_dataset.py:89-132, :272-296) and a 16-worker DataLoader.  Here a split is read in one pass:

RAF (`NeRAF_dataparser.py:140-150, 199-246`)::

    <root>/metadata/data-split.json          {"train": [[id, ...]], "validation": [[...]], "test": [[...]]}
    <root>/data/<id>/rx_pos.txt              "x,y,z"                      microphone position
    <root>/data/<id>/tx_pos.txt              "qx,qy,qz,qw,x,y,z"          speaker orientation (xyzw quaternion) + position
    <root>/data/<id>/rir.wav                 48 kHz mono impulse response

SoundSpaces (`NeRAF_dataparser.py:304-336, 369-394`, `NeRAF_datamanager.py:313-317`)::

    <root>/metadata/points.txt               "<point>\\t<x>\\t<y>\\t<z>"     -> position (x, z, -y): up is the second axis
    <root>/metadata_AudioNeRF/split.json     {"train": ["<rot>/<r>_<s>", ...], "test": [...]}
    <root>/binaural_magnitudes_sr22050/<rot>/<r>_<s>.npy     float magnitudes [2, 257, T]

Pose / rotation arithmetic is pinned to the reference's own parsers on synthetic trees (tests/golden/g6_dataparsers.npz,
tests/test_dataparsers.py).  Decoding `rir.wav` uses scipy (the reference calls `librosa.load(sr=None)`, i.e. soundfile's float
conversion: integer PCM / 2^(bits-1), channels averaged); resampling to 16 kHz (`librosa.resample`, NeRAF_dataset.py:98-105) is not
restated -- the RAF configuration of NeRAF trains at 48 kHz (NeRAF_config.py) -- and raises."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .config import SceneBox
from .data import DeviceRIRBank


@dataclass
class AudioDataparserOutputs:
    """Fields of the reference's `RAFDataparserOutputs` / `SoundSpacesDataparserOutputs` (NeRAF_dataparser.py:24-37, :82-88, :263-269)."""
    audios_filenames: List[str]
    microphone_poses: torch.Tensor            # [N, 3] float64
    source_poses: torch.Tensor                # [N, 3] float64
    scene_box: SceneBox
    source_rotations: Optional[torch.Tensor] = None        # RAF: speaker direction cosine in [0, 1]^3
    microphone_rotations: Optional[torch.Tensor] = None    # SoundSpaces: listener direction cosine in [0, 1]^3

    @property
    def rotations(self) -> torch.Tensor:
        return self.source_rotations if self.source_rotations is not None else self.microphone_rotations


def _direction_cosine(deg: np.ndarray) -> np.ndarray:
    """Angle around the up axis -> (cos, 0, sin) mapped to [0, 1] (the NAcF's SH input range), NeRAF_dataparser.py:231-234, :381-383."""
    rad = np.deg2rad(deg)
    return (np.stack([np.cos(rad), np.zeros_like(rad), np.sin(rad)], axis=-1) + 1.0) / 2.0


def _scene_box(mic: np.ndarray) -> SceneBox:
    aabb = np.array([mic.min(axis=0) - 1.0, mic.max(axis=0) + 1.0])          # 1 m margin, :159-163 / :342-345
    return SceneBox(aabb=torch.tensor(aabb, dtype=torch.float32))


def _numbers(path: str) -> List[float]:
    with open(path, "r") as f:
        return [float(v) for v in f.readline().replace("\n", "").split(",")]


def parse_raf(root: str, split: str = "train") -> AudioDataparserOutputs:
    """`RAFDataParser._generate_dataparser_outputs` (NeRAF_dataparser.py:118-176) for split 'train' | 'val' | anything else = test."""
    from scipy.spatial.transform import Rotation
    with open(os.path.join(root, "metadata/data-split.json")) as f:
        splits = json.load(f)
    key = {"train": "train", "val": "validation"}.get(split, "test")
    files = list(splits[key][0])
    n = len(files)
    mic, src, quat = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 4))
    for i, name in enumerate(files):
        rx = _numbers(os.path.join(root, "data", name, "rx_pos.txt"))
        tx = _numbers(os.path.join(root, "data", name, "tx_pos.txt"))
        mic[i], quat[i], src[i] = rx[:3], tx[:4], tx[4:7]
    # speaker yaw: first angle of the intrinsic y-x-z Euler decomposition, rounded to whole degrees (:227-230)
    yaw = np.round(Rotation.from_quat(quat).as_euler("yxz", degrees=True)[:, 0], decimals=0) if n else np.empty((0,))
    return AudioDataparserOutputs(files, torch.from_numpy(mic), torch.from_numpy(src), _scene_box(mic),
                                  source_rotations=torch.from_numpy(_direction_cosine(yaw)))


def parse_soundspaces(root: str, split: str = "train") -> AudioDataparserOutputs:
    """`SoundSpacesDataParser._generate_dataparser_outputs` (NeRAF_dataparser.py:293-357); there is no validation split."""
    positions = {}
    with open(os.path.join(root, "metadata/points.txt"), "r") as f:
        for line in f:
            row = line.replace("\n", "").split("\t")
            if len(row) < 4:
                continue
            x, y, z = (float(v) for v in row[1:4])
            positions[row[0]] = (x, z, -y)                                         # :308
    with open(os.path.join(root, "metadata_AudioNeRF/split.json"), "r") as f:
        files = list(json.load(f)["train" if split == "train" else "test"])
    n = len(files)
    mic, src, deg = np.empty((n, 3)), np.empty((n, 3)), np.empty((n,))
    for i, name in enumerate(files):
        rot, r_s = name.split("/")
        r, s = r_s.split("_")[:2]
        mic[i], src[i], deg[i] = positions[r], positions[s], int(rot)
    return AudioDataparserOutputs(files, torch.from_numpy(mic), torch.from_numpy(src), _scene_box(mic),
                                  microphone_rotations=torch.from_numpy(_direction_cosine(deg)))


def parse_raf_inference(poses_file: str) -> AudioDataparserOutputs:
    """Split 'inference' of the RAF parser (NeRAF_dataparser.py:130-138, :248-260): a pickled dict {'mic_poses' [N,3], 'source_poses' [3],
    'rots' [3]} (the file `AVN_RENDER_POSES` points to); the one source pose and rotation are repeated for every microphone pose."""
    data = np.load(poses_file, allow_pickle=True).item()
    mic = np.asarray(data["mic_poses"], dtype=np.float64)
    n = mic.shape[0]
    src = np.repeat(np.expand_dims(np.asarray(data["source_poses"], dtype=np.float64), 0), n, axis=0)
    rot = np.repeat(np.expand_dims(np.asarray(data["rots"], dtype=np.float64), 0), n, axis=0)
    return AudioDataparserOutputs(list(range(n)), torch.from_numpy(mic), torch.from_numpy(src), _scene_box(mic),
                                  source_rotations=torch.from_numpy(rot))


def parse_soundspaces_inference(poses_file: str) -> AudioDataparserOutputs:
    """Split 'inference' of the SoundSpaces parser (NeRAF_dataparser.py:311-322, :396-447): a pickle {'scene_obs': [{'pose', 'quat',
    'source'}, ...]} from the Habitat renderer.  Listener yaw = first angle of the intrinsic y-z-x Euler decomposition of the xyzw
    quaternion, wrapped to [0, 360); the microphone is put at the source's height (training used a fixed height)."""
    import pickle
    from scipy.spatial.transform import Rotation
    with open(poses_file, "rb") as f:
        obs = pickle.load(f)["scene_obs"]
    n = len(obs)
    mic, src, deg = np.empty((n, 3)), np.empty((n, 3)), np.empty((n,))
    for i, v in enumerate(obs):
        yaw = Rotation.from_quat(v["quat"]).as_euler("yzx", degrees=True)[0]
        if yaw < 0:
            yaw = 360 + yaw
        deg[i] = yaw % 360
        src[i] = np.asarray(v["source"], dtype=np.float64)[:3]
        mic[i] = np.asarray(v["pose"], dtype=np.float64)[:3]
        mic[i, 1] = src[i, 1]
    return AudioDataparserOutputs(list(obs), torch.from_numpy(mic), torch.from_numpy(src), _scene_box(mic),
                                  microphone_rotations=torch.from_numpy(_direction_cosine(deg)))


def read_wav_mono(path: str, expect_sr: Optional[int] = 48000) -> np.ndarray:
    """float32 mono samples of a PCM / float wav, as `librosa.load(path, sr=None)` returns them."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if expect_sr is not None and sr != expect_sr:
        raise ValueError(f"Loaded sample rate should be {expect_sr // 1000}kHz, got {sr}")        # NeRAF_dataset.py:95-96
    if data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        data = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
    else:
        data = data.astype(np.float32)
    return data.mean(axis=1).astype(np.float32) if data.ndim == 2 else data


def resample(x: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    """Sample-rate conversion along the last axis: the role of ``librosa.resample`` in NeRAF_dataset.py:98-105, :145-155, :335-342.
    librosa 0.10 defaults to the soxr_hq resampler; neither librosa nor soxr is in this image, so this is scipy's polyphase FIR
    (``scipy.signal.resample_poly``, Kaiser window beta 5.0) -- the same band-limited interpolation up to the anti-aliasing filter's
    shape: TOLERANCE-level parity (pass band within ~1e-2 relative on broadband RIRs), not sample-exact, and documented as such in
    DESIGN.md.  Output length ceil(n * target / orig), as librosa's."""
    from math import gcd
    from scipy.signal import resample_poly
    if orig_sr == target_sr:
        return np.asarray(x, dtype=np.float32)
    g = gcd(int(orig_sr), int(target_sr))
    y = resample_poly(np.asarray(x, dtype=np.float64), int(target_sr) // g, int(orig_sr) // g, axis=-1)
    n_out = int(np.ceil(x.shape[-1] * target_sr / orig_sr))
    return np.ascontiguousarray(y[..., :n_out], dtype=np.float32)


def _fix_length(x: np.ndarray, size: int) -> np.ndarray:
    """librosa.util.fix_length along the last axis: crop or zero-pad to ``size``."""
    n = x.shape[-1]
    if n >= size:
        return x[..., :size]
    pad = [(0, 0)] * (x.ndim - 1) + [(0, size - n)]
    return np.pad(x, pad, "constant")


def load_raf_rir(path: str, fs: int = 48000) -> np.ndarray:
    """One RAF recording as the dataset hands it to the STFT (NeRAF_dataset.py:93-105): decoded at 48 kHz; for ``fs`` = 16000
    zero-extended to at least 0.1 s and resampled (the reference's own branch tests ``data.shape[1]`` on a mono signal and cannot run;
    its intent is what is implemented)."""
    data = read_wav_mono(path)
    if fs != 48000:
        if data.shape[-1] < int(48000 * 0.1):
            data = _fix_length(data, int(48000 * 0.1))
        data = resample(data, 48000, fs)
    return data


def bank_from_raf(root: str, split: str = "train", fs: int = 48000, max_len: int = 60, max_len_seconds: float = None,
                  max_len_samples: int = None, device=None, chunk: int = 256):
    """One pass over a RAF split -> (DeviceRIRBank [N, max_len, 1, F], AudioDataparserOutputs).  The crop length is given EITHER as
    ``max_len_seconds`` (the config's ``max_len``, 0.32) OR as ``max_len_samples`` (what the dataset receives,
    NeRAF_datamanager.py:214: int(max_len * fs)); default 0.32 s.  RIRs are decoded, (for fs = 16000: resampled, see ``resample``),
    cropped, transformed ``chunk`` at a time on ``device``."""
    if fs not in (48000, 16000):
        raise ValueError("Sample rate not supported")                                              # NeRAF_dataset.py:56-66
    if max_len_seconds is not None and max_len_samples is not None:
        raise ValueError("give max_len_seconds or max_len_samples, not both")
    out = parse_raf(root, split)
    n_time = int(max_len_samples) if max_len_samples is not None else int((0.32 if max_len_seconds is None else max_len_seconds) * fs)
    banks = []
    for c0 in range(0, len(out.audios_filenames), chunk):
        names = out.audios_filenames[c0:c0 + chunk]
        waves = np.zeros((len(names), n_time), np.float32)
        for i, name in enumerate(names):
            w = load_raf_rir(os.path.join(root, "data", name, "rir.wav"), fs)[:n_time]
            waves[i, :w.shape[0]] = w          # a shorter file is zero-extended (its STFT frames past the end are silence)
        sl = slice(c0, c0 + len(names))
        banks.append(DeviceRIRBank.from_waveforms(torch.from_numpy(waves), fs, max_len, out.microphone_poses[sl], out.source_poses[sl],
                                                  out.rotations[sl], device=device))
    return _concat(banks), out


def load_soundspaces_waveform(path: str, fs: int = 22050, max_len_time: int = 76 * 128) -> np.ndarray:
    """Ground-truth binaural waveform of a SoundSpaces eval item (NeRAF_dataset.py:326-349): ``binaural_rirs/<name>.wav`` (44.1 kHz,
    [n, 2]) clipped to [-1, 1], an empty file replaced by 0.5 s of silence, resampled to ``fs`` (zero-extended to 0.1 s first when
    shorter), cropped or zero-padded to ``max_len_time`` samples -> float32 [2, max_len_time]."""
    from scipy.io import wavfile
    _, data = wavfile.read(path)
    if np.issubdtype(data.dtype, np.integer):          # the simulator writes float wavs; integer PCM is scaled like librosa would
        data = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
    w = np.clip(np.asarray(data, dtype=np.float32), -1.0, 1.0)
    w = w.T if w.ndim == 2 else np.stack([w, w])
    if w.shape[1] == 0:
        w = np.zeros((2, int(fs * 0.5)), np.float32)
    if fs != 44100:
        if w.shape[1] < int(44100 * 0.1):
            w = _fix_length(w, int(44100 * 0.1))
        w = resample(w, 44100, fs)
    return np.ascontiguousarray(_fix_length(w, int(max_len_time)), dtype=np.float32)


def bank_from_soundspaces(root: str, split: str = "train", max_len: int = 76, device=None):
    """One pass over a SoundSpaces split -> (DeviceRIRBank [N, max_len, 2, 257], AudioDataparserOutputs): log(magnitude + 1e-3), cropped
    to `max_len` frames or padded with the file's smallest magnitude (NeRAF_dataset.py:279-285, :313-319)."""
    out = parse_soundspaces(root, split)
    rows = []
    for name in out.audios_filenames:
        mag = np.load(os.path.join(root, "binaural_magnitudes_sr22050", name + ".npy")).astype(np.float32)        # [2, 257, T]
        if mag.shape[2] >= max_len:
            mag = mag[:, :, :max_len]
        else:
            mag = np.pad(mag, ((0, 0), (0, 0), (0, max_len - mag.shape[2])), "constant", constant_values=mag.min())
        rows.append(np.log(mag + 1e-3).transpose(2, 0, 1))                                                         # [T, C, F]
    log_mag = torch.from_numpy(np.stack(rows)) if rows else torch.empty((0, max_len, 2, 257))
    dev = device if device is not None else "cpu"
    return DeviceRIRBank(log_mag.to(dev), out.microphone_poses.to(dev), out.source_poses.to(dev), out.rotations.to(dev)), out


def _concat(banks: List[DeviceRIRBank]) -> DeviceRIRBank:
    if len(banks) == 1:
        return banks[0]
    cat = lambda name: torch.cat([getattr(b, name) for b in banks], dim=0)
    return DeviceRIRBank(cat("log_mag"), cat("mic_pose"), cat("source_pose"), cat("rot"))
