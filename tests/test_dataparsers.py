"""On-disk dataset formats -> device-resident RIR bank (neraf_amd/dataparsers.py).

Pose / rotation / scene-box arithmetic is PINNED: tests/golden/g6_dataparsers.npz holds the outputs of the reference's own
`RAFDataParser` / `SoundSpacesDataParser._generate_dataparser_outputs` (NeRAF_dataparser.py:118-176, :293-357) on the synthetic scene
trees of `neraf_amd.synth.raf_tree` / `soundspaces_tree` (generator: tests/tools/gen_golden.py g6).  The audio payload is checked
against the reference's formulas restated here (NeRAF_dataset.py:107-115, :279-285, :313-321)."""
import os

import numpy as np
import pytest
import torch

from neraf_amd import synth
from neraf_amd.data import DeviceRIRBank
from neraf_amd.dataparsers import bank_from_raf, bank_from_soundspaces, parse_raf, parse_soundspaces, read_wav_mono

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g6_dataparsers.npz")


@pytest.mark.parametrize("split", ["train", "val", "test"])
def test_raf_poses_rotations_and_scene_box_match_the_reference_parser(tmp_path, split):
    g = np.load(GOLD)
    synth.write_tree(str(tmp_path), synth.raf_tree())
    out = parse_raf(str(tmp_path), split)
    assert list(out.audios_filenames) == [str(n) for n in g[f"raf_{split}_names"]]
    np.testing.assert_array_equal(out.microphone_poses.numpy(), g[f"raf_{split}_mic"])
    np.testing.assert_array_equal(out.source_poses.numpy(), g[f"raf_{split}_src"])
    np.testing.assert_allclose(out.source_rotations.numpy(), g[f"raf_{split}_rot"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(out.scene_box.aabb.numpy(), g[f"raf_{split}_aabb"])
    assert out.microphone_poses.dtype == torch.float64 and out.scene_box.aabb.dtype == torch.float32
    assert out.microphone_rotations is None and out.rotations is out.source_rotations


@pytest.mark.parametrize("split", ["train", "test"])
def test_soundspaces_poses_rotations_and_scene_box_match_the_reference_parser(tmp_path, split):
    g = np.load(GOLD)
    synth.write_tree(str(tmp_path), synth.soundspaces_tree())
    out = parse_soundspaces(str(tmp_path), split)
    assert list(out.audios_filenames) == [str(n) for n in g[f"ss_{split}_names"]]
    np.testing.assert_array_equal(out.microphone_poses.numpy(), g[f"ss_{split}_mic"])
    np.testing.assert_array_equal(out.source_poses.numpy(), g[f"ss_{split}_src"])
    np.testing.assert_allclose(out.microphone_rotations.numpy(), g[f"ss_{split}_rot"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(out.scene_box.aabb.numpy(), g[f"ss_{split}_aabb"])
    # "val" is the test split too (there is no validation split, NeRAF_dataparser.py:331-333)
    if split == "test":
        assert parse_soundspaces(str(tmp_path), "val").audios_filenames == out.audios_filenames


def _write_raf_wavs(root, names, n=15360 + 700, dtype=np.float32):
    from scipy.io import wavfile
    sig = {}
    tt = np.arange(n) / 48000.0
    for i, name in enumerate(names):
        w = (synth.normal(f"wav.{name}", (n,), 0.3, np.float64) * np.exp(-tt / (0.03 + 0.01 * i))).astype(np.float32)
        path = os.path.join(root, "data", name, "rir.wav")
        if dtype == np.int16:
            q = np.clip(np.round(w * 32768.0), -32768, 32767).astype(np.int16)
            wavfile.write(path, 48000, q)
            sig[name] = q.astype(np.float32) / 32768.0
        else:
            wavfile.write(path, 48000, w)
            sig[name] = w
    return sig


@pytest.mark.parametrize("dtype", [np.float32, np.int16])
def test_raf_bank_is_the_tokenised_split(tmp_path, dtype):
    root = str(tmp_path)
    synth.write_tree(root, synth.raf_tree())
    out = parse_raf(root, "train")
    sig = _write_raf_wavs(root, out.audios_filenames, dtype=dtype)
    bank, out2 = bank_from_raf(root, "train", fs=48000, max_len=60, max_len_seconds=0.32, chunk=2)       # several chunks
    assert out2.audios_filenames == out.audios_filenames
    assert tuple(bank.log_mag.shape) == (5, 60, 1, 513) and len(bank) == 5 * 60
    # the reference item (NeRAF_dataset.py:107-115): decode, crop to max_len_time samples, STFT, log(|.| + 1e-3) of slice t
    waves = torch.from_numpy(np.stack([sig[n][:15360] for n in out.audios_filenames]))
    ref = DeviceRIRBank.from_waveforms(waves, 48000, 60, out.microphone_poses, out.source_poses, out.source_rotations)
    torch.testing.assert_close(bank.log_mag, ref.log_mag, rtol=0, atol=0)
    item = bank.get_data(2 * 60 + 17)
    assert item["audio_idx"] == 2 and item["time_query"] == 17 and tuple(item["data"].shape) == (1, 513)
    torch.testing.assert_close(item["mic_pose"], out.microphone_poses[2])
    torch.testing.assert_close(item["rot"], out.source_rotations[2])
    np.testing.assert_array_equal(read_wav_mono(os.path.join(root, "data", out.audios_filenames[0], "rir.wav")), sig[out.audios_filenames[0]])


def test_raf_rejects_other_sample_rates(tmp_path):
    from scipy.io import wavfile
    root = str(tmp_path)
    synth.write_tree(root, synth.raf_tree())
    names = parse_raf(root, "val").audios_filenames
    for name in names:
        wavfile.write(os.path.join(root, "data", name, "rir.wav"), 44100, np.zeros(20000, np.float32))
    with pytest.raises(ValueError, match="48kHz"):                      # NeRAF_dataset.py:95-96
        bank_from_raf(root, "val")
    with pytest.raises(ValueError, match="not supported"):              # NeRAF_dataset.py:56-66
        bank_from_raf(root, "val", fs=22050)


def test_soundspaces_bank_crops_long_and_pads_short_files(tmp_path):
    root = str(tmp_path)
    synth.write_tree(root, synth.soundspaces_tree())
    out = parse_soundspaces(root, "train")
    mags = {}
    for i, name in enumerate(out.audios_filenames):
        T = 50 + 9 * i                                                  # 50 .. 104 frames around max_len = 76
        m = np.abs(synth.normal(f"ss.mag.{name}", (2, 257, T), 0.2, np.float64)).astype(np.float32) + 1e-4
        path = os.path.join(root, "binaural_magnitudes_sr22050", name + ".npy")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.save(path, m)
        mags[name] = m
    bank, _ = bank_from_soundspaces(root, "train", max_len=76)
    assert tuple(bank.log_mag.shape) == (len(out.audios_filenames), 76, 2, 257)
    for i, name in enumerate(out.audios_filenames):
        m = mags[name]
        for t in (0, 49, 60, 75):
            # train item, NeRAF_dataset.py:279-285: slice t, or the file's smallest magnitude when the file is shorter
            want = np.log(m[:, :, t] + 1e-3) if t < m.shape[2] else np.log(np.ones((2, 257), np.float32) * m.min() + 1e-3)
            np.testing.assert_allclose(bank.get_data(i * 76 + t)["data"].numpy(), want, rtol=1e-6, atol=0)
        ev = bank.get_data_eval(i)["data"]                              # eval item [C, F, T], :313-321
        assert tuple(ev.shape) == (2, 257, 76)
    torch.testing.assert_close(bank.rot, out.microphone_rotations)
    # a duplicate name in the split (same receiver / source / rotation) is a separate row, as in the reference's list
    b = bank.batch(torch.tensor([0, 76 * 3 + 5]))
    assert tuple(b["data"].shape) == (2, 2, 257) and b["time_query"].tolist() == [0, 5] and b["audio_idx"].tolist() == [0, 3]


def test_disk_datamanager_serves_reference_shaped_batches(tmp_path):
    """DiskAudioDataManager over a synthetic RAF tree: the train split's scene box, batches of [B, 1, 513] slices with the parser's
    poses, whole-RIR eval items with the decoded waveform."""
    from neraf_amd.datamanagers import DiskAudioDataManager
    root = str(tmp_path)
    synth.write_tree(root, synth.raf_tree())
    names = [n for split in ("train", "val", "test") for n in parse_raf(root, split).audios_filenames]
    sig = _write_raf_wavs(root, names)
    dm = DiskAudioDataManager(root, dataset="RAF", batch_size=32)
    g = np.load(GOLD)
    np.testing.assert_array_equal(dm.train_dataset.scene_box.aabb.numpy(), g["raf_train_aabb"])
    assert dm.max_len == 60 and len(dm.train_dataset) == 5 * 60
    _, b = dm.next_train(0)
    assert tuple(b["data"].shape) == (32, 1, 513) and b["mic_pose"].dtype == torch.float64
    r, t = b["audio_idx"], b["time_query"]
    torch.testing.assert_close(b["data"], dm.train_dataset.bank.log_mag[r, t])
    np.testing.assert_array_equal(b["source_pose"].numpy(), g["raf_train_src"][r.numpy()])
    dm.eval_dataset.mode = "eval_image"
    assert len(dm.eval_dataset) == 3
    _, e = dm.next_eval_image(0)
    assert tuple(e["data"].shape) == (1, 513, 60) and tuple(e["waveform"].shape) == (1, 15360)
    first_test = parse_raf(root, "test").audios_filenames[0]
    np.testing.assert_array_equal(e["waveform"][0].numpy(), sig[first_test][:15360])


def test_inference_pose_files_match_the_reference_parsers(tmp_path):
    """The 'inference' splits (pose files named by AVN_RENDER_POSES, NeRAF_dataparser.py:130-138, :311-322, :248-260, :396-447)."""
    import pickle
    from neraf_amd.dataparsers import parse_raf_inference, parse_soundspaces_inference
    g = np.load(GOLD)
    raf, ss = synth.inference_pose_files()
    p1, p2 = str(tmp_path / "raf.npy"), str(tmp_path / "ss.pkl")
    np.save(p1, raf, allow_pickle=True)
    with open(p2, "wb") as f:
        pickle.dump(ss, f)
    o = parse_raf_inference(p1)
    np.testing.assert_array_equal(o.microphone_poses.numpy(), g["raf_inf_mic"])
    np.testing.assert_array_equal(o.source_poses.numpy(), g["raf_inf_src"])
    np.testing.assert_array_equal(o.source_rotations.numpy(), g["raf_inf_rot"])
    np.testing.assert_array_equal(o.scene_box.aabb.numpy(), g["raf_inf_aabb"])
    o = parse_soundspaces_inference(p2)
    np.testing.assert_array_equal(o.microphone_poses.numpy(), g["ss_inf_mic"])
    np.testing.assert_array_equal(o.source_poses.numpy(), g["ss_inf_src"])
    np.testing.assert_allclose(o.microphone_rotations.numpy(), g["ss_inf_rot"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(o.scene_box.aabb.numpy(), g["ss_inf_aabb"])


def test_resample_is_band_limited_interpolation():
    """dataparsers.resample stands in for librosa.resample (soxr_hq; neither is in the image): a tone well inside the new pass band
    must come out as the same tone at the new rate (amplitude / phase within 1e-2 away from the edges), a tone above the new Nyquist
    must be suppressed, and the length follows librosa's ceil(n * target / orig)."""
    from neraf_amd.dataparsers import resample
    n = 44100
    t = np.arange(n) / 44100.0
    x = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    y = resample(x, 44100, 22050)
    assert y.shape == (22050,) and y.dtype == np.float32
    want = np.sin(2 * np.pi * 1000.0 * np.arange(22050) / 22050.0)
    assert np.abs(y[200:-200] - want[200:-200]).max() < 1e-2
    hi = np.sin(2 * np.pi * 15000.0 * t).astype(np.float32)          # above 11.025 kHz: aliased unless filtered
    assert np.abs(resample(hi, 44100, 22050)[200:-200]).max() < 2e-2
    assert resample(np.zeros((2, 4801), np.float32), 48000, 16000).shape == (2, 1601)     # ceil(4801 / 3)
    np.testing.assert_array_equal(resample(x, 22050, 22050), x)


def test_raf_bank_at_16k_resamples_and_uses_the_16k_stft(tmp_path):
    """RAF at fs = 16000 (NeRAF_dataset.py:56-66, :98-105): decode at 48 kHz, resample, crop to int(0.32 * 16000) samples,
    STFT (512, 256, 128) -> [N, 40, 1, 257]; items equal the same pipeline applied by hand."""
    from neraf_amd.dataparsers import load_raf_rir, resample
    root = str(tmp_path)
    synth.write_tree(root, synth.raf_tree())
    out = parse_raf(root, "train")
    sig = _write_raf_wavs(root, out.audios_filenames)
    bank, _ = bank_from_raf(root, "train", fs=16000, max_len=40, max_len_seconds=0.32)
    assert tuple(bank.log_mag.shape) == (5, 40, 1, 257)
    n_time = int(0.32 * 16000)
    waves = torch.from_numpy(np.stack([resample(sig[n], 48000, 16000)[:n_time] for n in out.audios_filenames]))
    ref = DeviceRIRBank.from_waveforms(waves, 16000, 40, out.microphone_poses, out.source_poses, out.source_rotations)
    torch.testing.assert_close(bank.log_mag, ref.log_mag, rtol=0, atol=0)
    w = load_raf_rir(os.path.join(root, "data", out.audios_filenames[0], "rir.wav"), 16000)
    assert w.shape[0] == int(np.ceil(sig[out.audios_filenames[0]].shape[0] / 3))
    with pytest.raises(ValueError, match="not both"):
        bank_from_raf(root, "train", max_len_seconds=0.32, max_len_samples=15360)
    # explicit units (advisor finding): a very short crop given in SAMPLES is not reinterpreted as seconds
    b2, _ = bank_from_raf(root, "train", fs=48000, max_len=2, max_len_samples=600)
    b3, _ = bank_from_raf(root, "train", fs=48000, max_len=2, max_len_seconds=600 / 48000)
    torch.testing.assert_close(b2.log_mag, b3.log_mag, rtol=0, atol=0)


def test_soundspaces_disk_manager_serves_ground_truth_waveforms(tmp_path):
    """SoundSpaces eval items carry the ground-truth binaural waveform (NeRAF_dataset.py:326-349): binaural_rirs/<name>.wav at
    44.1 kHz -> clip -> 22.05 kHz -> crop / zero-pad to max_len * 128 samples; an empty file becomes silence."""
    from scipy.io import wavfile
    from neraf_amd.datamanagers import DiskAudioDataManager
    from neraf_amd.dataparsers import resample
    root = str(tmp_path)
    synth.write_tree(root, synth.soundspaces_tree())
    names = {n for split in ("train", "test") for n in parse_soundspaces(root, split).audios_filenames}
    test_names = parse_soundspaces(root, "test").audios_filenames
    sig = {}
    for k, name in enumerate(sorted(names)):
        m = np.abs(synth.normal(f"ssw.mag.{name}", (2, 257, 80), 0.2, np.float64)).astype(np.float32) + 1e-4
        pm = os.path.join(root, "binaural_magnitudes_sr22050", name + ".npy")
        os.makedirs(os.path.dirname(pm), exist_ok=True)
        np.save(pm, m)
        n = 0 if name == test_names[1] else 9000 + 3000 * k             # one empty file; short and long ones around 76*128*2
        w = (synth.normal(f"ssw.wav.{name}", (n, 2), 0.7, np.float64) * np.exp(-np.arange(n) / 4000.0)[:, None]).astype(np.float32)
        pw = os.path.join(root, "binaural_rirs", name + ".wav")
        os.makedirs(os.path.dirname(pw), exist_ok=True)
        wavfile.write(pw, 44100, w)
        sig[name] = w
    dm = DiskAudioDataManager(root, dataset="SoundSpaces", batch_size=16)
    dm.eval_dataset.mode = "eval_image"
    n_time = 76 * 128
    for i, name in enumerate(test_names):
        item = dm.eval_dataset[i]
        assert tuple(item["data"].shape) == (2, 257, 76) and tuple(item["waveform"].shape) == (2, n_time)
        if sig[name].shape[0] == 0:
            assert float(item["waveform"].abs().max()) == 0.0
            continue
        want = resample(np.clip(sig[name], -1, 1).T, 44100, 22050)
        want = want[:, :n_time] if want.shape[1] >= n_time else np.pad(want, ((0, 0), (0, n_time - want.shape[1])))
        np.testing.assert_array_equal(item["waveform"].numpy(), want)
