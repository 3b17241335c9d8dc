"""Device-resident RIR bank (neraf_amd/data.py): item semantics of NeRAF_dataset.py:85-128 on CPU."""
import numpy as np
import torch

from neraf_amd import synth
from neraf_amd.data import DeviceRIRBank
from neraf_amd.evaluator import spectrogram


def _bank(n_rir=5, n=15360, max_len=60):
    fs = 48000
    tt = np.arange(n) / fs
    waves = torch.from_numpy(np.stack([synth.normal(f"bank.{i}", (n,), 1.0, np.float64) * np.exp(-tt / (0.03 + 0.01 * i))
                                       for i in range(n_rir)])).float()
    mic = torch.from_numpy(synth.uniform("bank.mic", (n_rir, 3), -1.0, 1.0)).double()
    src = torch.from_numpy(synth.uniform("bank.src", (n_rir, 3), -1.0, 1.0)).double()
    rot = torch.from_numpy(synth.uniform("bank.rot", (n_rir, 3), -3.0, 3.0)).double()
    return waves, mic, src, rot, DeviceRIRBank.from_waveforms(waves, fs, max_len, mic, src, rot, max_len_time=int(0.32 * fs))


def test_items_follow_reference_indexing_and_values():
    waves, mic, src, rot, bank = _bank()
    assert len(bank) == 5 * 60 and bank.log_mag.shape == (5, 60, 1, 513)
    for idx in (0, 59, 60, 137, 299):
        it = bank.get_data(idx)
        r, t = idx // 60, idx % 60
        assert it["audio_idx"] == r and it["time_query"] == t and it["data"].shape == (1, 513)
        ref = torch.log(spectrogram(waves[r][None, :int(0.32 * 48000)], 1024, 512, 256).abs()[:, :, t] + 1e-3)     # NeRAF_dataset.py:109-111
        np.testing.assert_allclose(it["data"].numpy(), ref.numpy(), rtol=0, atol=0)
        assert torch.equal(it["mic_pose"], mic[r]) and it["mic_pose"].dtype == torch.float64
    ev = bank.get_data_eval(3)
    assert ev["data"].shape == (1, 513, 60) and torch.equal(ev["data"][:, :, 7], bank.get_data(3 * 60 + 7)["data"])


def test_short_rirs_are_padded_with_their_quietest_value():
    _, _, _, _, bank = _bank(n_rir=2, n=5000, max_len=60)          # 5000 samples -> 20 frames
    frames = 5000 // 256 + 1
    tail = bank.log_mag[:, frames:]
    assert tail.shape[1] == 60 - frames
    for r in range(2):
        assert torch.all(tail[r] == tail[r].flatten()[0])
        np.testing.assert_allclose(float(tail[r].flatten()[0]), float(bank.log_mag[r, :frames].min()), rtol=1e-6)


def test_batches_are_collated_items():
    _, _, _, _, bank = _bank()
    g = torch.Generator().manual_seed(3)
    b = bank.next_train(64, generator=g)
    assert b["data"].shape == (64, 1, 513) and b["time_query"].shape == (64,) and b["mic_pose"].shape == (64, 3)
    assert int(b["time_query"].max()) < 60 and int(b["audio_idx"].max()) < 5
    for k in (0, 17, 63):
        it = bank.get_data(int(b["audio_idx"][k]) * 60 + int(b["time_query"][k]))
        assert torch.equal(b["data"][k], it["data"]) and torch.equal(b["source_pose"][k], it["source_pose"])
    # uniform over (rir, slice): every RIR shows up in a large batch
    big = bank.next_train(4096, generator=g)
    assert len(torch.unique(big["audio_idx"])) == 5 and len(torch.unique(big["time_query"])) == 60


def test_bank_datamanagers_draw_distinct_batches_per_rank():
    """Advisor finding (round 2): with world_size > 1 every rank must draw its own slices even when all ranks seeded torch
    identically (identical initial weights): per-rank generator = seed + 7919 * rank."""
    import torch
    from neraf_amd.datamanagers import SyntheticAudioDataManager
    idx = []
    for rank in (0, 1):
        torch.manual_seed(0)
        dm = SyntheticAudioDataManager(3, 1, batch_size=64, world_size=2, local_rank=rank)
        assert dm.generator is not None
        _, b = dm.next_train(0)
        idx.append((b["audio_idx"] * 1000 + b["time_query"]).clone())
    assert not torch.equal(idx[0], idx[1])
    torch.manual_seed(0)
    dm = SyntheticAudioDataManager(3, 1, batch_size=64, world_size=2, local_rank=1)       # reproducible per rank
    _, b = dm.next_train(0)
    assert torch.equal(b["audio_idx"] * 1000 + b["time_query"], idx[1])
    assert SyntheticAudioDataManager(3, 1, batch_size=8).generator is None               # single process: default generator
