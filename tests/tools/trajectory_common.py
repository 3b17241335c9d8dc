"""The scenario of the trajectory-level parity test (tests/test_gpu_trajectory.py, fixture g7): a small LEARNABLE joint scene --
a textured box room seen by 12 cameras + position-dependent decaying RIRs in a RAF-like room -- whose batches are pure functions
of the iteration number (neraf_amd/synth.py), so the HIP pipeline on the GPU and the CPU oracle (oracle/trainer.py) train on
identical data from identical initial weights.  Imported by the fixture generator (gen_trajectory.py) and by the GPU test.

Horizon: 100 iterations, the audio branch from iteration 6.  Joint training of a 43-layer train-mode-BatchNorm encoder with Adam is
chaotic in the strict sense: two fp32 oracle runs that differ only by fp16 rounding of their parameters track each other to 1e-2
(predicted log-magnitudes) for ~100 audio iterations and then separate within ~15 iterations (audio loss 0.0033 vs 0.0066 at
iteration 120; measured with this scenario, DESIGN.md "Trajectory-level parity").  A parity statement about an fp16 engine is
therefore only meaningful inside that horizon, and only relative to that band, which the fixture carries (``probe_*`` arrays)."""
import numpy as np
import torch

from neraf_amd import synth

CFG = dict(R=512, B=128, steps=100, start_step_audio=5, grid_step=1 / 64, n_cam=12, n_rir=16, n_rir_eval=2, T=60, F=513, C=1, fs=48000,
           eval_hw=(32, 48), tag="traj", camera_opt=False)

# Two scenarios over the same scene, batches and initial weights:
#   "g7_trajectory"      -- camera optimizer off: the clean diagnostic (nothing but the radiance and acoustic fields trains);
#   "g8_trajectory_pose" -- camera optimizer SO3xR3 ON, the reference's configuration (NeRAF_config.py:97; what bench.py times): the
#                           per-camera pose deltas start at zero on exact poses, so Adam (eps 1e-15) random-walks them on the sign of
#                           tiny photometric gradients -- a second noise source, visible in the band of this scenario.
#   "g9_long"            -- the G7 scene trained for 1000 iterations (audio from iteration 6): far beyond the horizon inside which two
#                           16-bit-perturbed runs of this chaotic system stay tensor-comparable, and far enough for the METRICS of
#                           BASELINE.json ("PSNR & T60 err vs ref") to mean something (T60 error ~10 % instead of ~650 %): compared
#                           metric by metric against the spread of the oracle's own precision probes.
#   "g10_long_pose"      -- the same 1000 iterations with the camera optimizer SO3xR3 ON: the reference's configuration (NeRAF_config.py:97),
#                           what bench.py times; 16 held-out RIRs (round 6: with round 5's four, the oracle's own summation-order
#                           probe moved its T60 error from 20.0 to 9.9 % -- the gate carried no information).
SCENARIOS = {"g7_trajectory": dict(CFG), "g8_trajectory_pose": dict(CFG, camera_opt=True), "g9_long": dict(CFG, steps=1000, n_rir_eval=8),
             "g10_long_pose": dict(CFG, steps=1000, n_rir_eval=16, camera_opt=True)}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rir_bank(n: int, tag: str, cfg=CFG):
    """(log-magnitude [n, T, C, F] as the dataset tokenises RIRs -- NeRAF_dataset.py:107-117: STFT(1024, 512, 256), log(|.| + 1e-3) --
    waveforms [n, 1, samples], poses)."""
    from neraf_amd.evaluator import spectrogram
    r = synth.trajectory_rirs(n, tag, fs=cfg["fs"], n_samples=256 * cfg["T"])
    w = T(r["waveforms"])
    mag = spectrogram(w[:, None, :], 1024, 512, 256).abs()[..., :cfg["T"]]              # [n, 1, F, T]
    log_mag = torch.log(mag + 1e-3).permute(0, 3, 1, 2).contiguous()                   # [n, T, C, F]
    return {"log_mag": log_mag, "waveforms": w[:, None, :], "mic_pose": T(r["mic_pose"]), "source_pose": T(r["source_pose"]),
            "rot": T(r["rot"]), "tau": r["tau"]}


def ray_batch(step: int, cfg=CFG):
    rb = synth.trajectory_ray_batch(step, cfg["R"], cfg["n_cam"], cfg["tag"])
    return {"origins": T(rb["origins"]), "directions": T(rb["directions"]), "camera_indices": T(rb["camera_indices"]),
            "rgb": T(rb["rgb"]), "jitters": [T(j) for j in rb["jitters"]]}


def audio_batch(step: int, bank, cfg=CFG):
    idx = T(synth.trajectory_audio_indices(step, cfg["B"], cfg["n_rir"], cfg["T"], cfg["tag"]))
    r, t = idx // cfg["T"], idx % cfg["T"]
    return {"time_query": t, "mic_pose": bank["mic_pose"][r], "source_pose": bank["source_pose"][r], "rot": bank["rot"][r],
            "data": bank["log_mag"][r, t]}


def initial_weights(grid_totals=None, cfg=CFG):
    """``grid_totals`` = hash-table rows (proposal 0, proposal 1, main field); None: from the oracle's grid specs (the generator's
    side; the HIP side passes its models' table sizes, so that a disagreement about the tcnn level layout fails loudly)."""
    if grid_totals is None:
        from oracle import vision as V
        spec = V.NerfactoSpec()
        grid_totals = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    tot = tuple(int(v) for v in grid_totals)
    P = {k: T(v) for k, v in synth.vision_params(tot, num_train_data=cfg["n_cam"], table_scale=1e-4, prefix="traj.").items()}
    sdn = {k: T(v) for k, v in synth.nacf_state_dict(1187, 512, cfg["C"], cfg["F"]).items()}
    sdr = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    return P, sdn, sdr


def psnr(a, b) -> float:
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float(10.0 * torch.log10(1.0 / torch.mean((a - b) ** 2)))


# ---- the HIP side: train neraf_amd's pipeline on the scenario, evaluate, compare with the fixture -------------------------------------
def _shard(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n items for ``rank`` (the global batch of an iteration is split, SURVEY 8e "Partitioning")."""
    lo = (n * rank) // world
    return lo, (n * (rank + 1)) // world


class _StepRays:
    """Vision data manager of the scenario: iteration ``step`` -> (this rank's shard of) its ray bundle and target colours."""

    def __init__(self, dev, rank: int = 0, world: int = 1):
        self.dev, self.rank, self.world = dev, rank, world
        self.train_num_rays_per_batch = CFG["R"]      # the GLOBAL batch: it sizes the grid-refresh window, which the model shards itself
        self.jit = {}

    def next_train(self, step):
        from neraf_amd.vision import RayBundle
        b = ray_batch(step)
        lo, hi = _shard(CFG["R"], self.rank, self.world)
        self.jit[step] = [j.reshape(-1)[lo:hi].contiguous().to(self.dev) for j in b["jitters"]]
        return (RayBundle(b["origins"][lo:hi].to(self.dev), b["directions"][lo:hi].to(self.dev), b["camera_indices"][lo:hi].to(self.dev)),
                {"image": b["rgb"][lo:hi].to(self.dev)})


class _StepSlices:
    def __init__(self, bank, dev, rank: int = 0, world: int = 1):
        self.bank, self.dev, self.rank, self.world = bank, dev, rank, world

    def next_train(self, step):
        lo, hi = _shard(CFG["B"], self.rank, self.world)
        return None, {k: v[lo:hi].contiguous().to(self.dev) for k, v in audio_batch(step, self.bank).items()}


def run_hip_trajectory(dev, steps=None, cfg=CFG, rank: int = 0, world: int = 1, fixed_scale: bool = True):
    """Train the HIP pipeline on the scenario; returns (loss curves [steps, 5], held-out image [H,W,3], held-out STFTs [n,T,C,F],
    pipeline, eval bank).  ``world`` > 1 (torch.distributed initialised by the caller): this process is rank ``rank`` of a
    data-parallel job -- it trains on its contiguous shard of every iteration's rays and RIR slices; rays / slices / refresh cells
    are sharded, loss sums and gradients reduced by neraf_amd.parallel; the per-rank loss curves are local for the radiance terms
    (mean over the rank's rays) and global for the audio terms."""
    from neraf_amd import config as Cfg
    from neraf_amd import synth
    from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
    from neraf_amd.pipeline import NeRAFPipeline
    from neraf_amd.vision import NeRAFVisionModel, RayBundle
    steps = cfg["steps"] if steps is None else steps
    if cfg.get("camera_opt"):
        vcfg = Cfg.NeRAFVisionModelConfig(camera_optimizer=Cfg.CameraOptimizerConfig(mode="SO3xR3"))
        vm = vcfg.setup(scene_box=Cfg.SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=cfg["n_cam"], metadata={},
                        device=dev, grad_scaler=None, seed_points=None)
    else:
        vm = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg["n_cam"])
    P, sdn, sdr = initial_weights((vm.proposal_networks[0].table.shape[0], vm.proposal_networks[1].table.shape[0],
                                   vm.field.module.table.shape[0]))
    with torch.no_grad():
        for i in range(2):
            vm.proposal_networks[i].table.copy_(P[f"prop{i}.table"])
            vm.proposal_networks[i].w0.copy_(P[f"prop{i}.w0"])
            vm.proposal_networks[i].w1.copy_(P[f"prop{i}.w1"])
        f = vm.field.module
        f.table.copy_(P["field.table"])
        for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding"):
            getattr(f, k).copy_(P["field." + k])
    am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=cfg["grid_step"]), T(synth.audio_aabb()),
                         process_group=True if world > 1 else None)
    am.field.load_state_dict(sdn)
    am.resnet3d.backbone_net.load_state_dict(sdr)
    vm.to(dev).train(); am.to(dev).train()
    bank = rir_bank(cfg["n_rir"], cfg["tag"] + ".train")
    rays = _StepRays(dev, rank, world)
    vm.jitter_fn = lambda step, R, device: rays.jit[step]
    pipe = NeRAFPipeline(vm, am, datamanager=rays, audio_datamanager=_StepSlices(bank, dev, rank, world),
                         start_step_audio=cfg["start_step_audio"], world_size=world, local_rank=rank)
    if world > 1:
        pipe.attach_gradient_reducer()
    import os
    init_scale = float(os.environ.get("NERAF_TRAJ_INIT_SCALE", "65536"))       # (tests/tools/g7_hip_toggles.py varies it: an experiment)
    opts, scaler = pipe.make_optimizers(init_scale=init_scale, optimizers_config=Cfg.default_optimizers(cfg["start_step_audio"]),
                                        with_schedulers=True)
    keys = ["rgb_loss", "interlevel_loss", "distortion_loss", "audio_sc_loss", "audio_mag_loss"]
    rows = []
    for s in range(steps):
        _, ld = pipe.train_iteration(s, opts, scaler)
        rows.append(torch.stack([ld[k].detach().float().reshape(()) if k in ld else torch.full((), float("nan"), device=dev) for k in keys]))
    curves = torch.stack(rows).cpu().numpy().astype(np.float64) if rows else np.zeros((0, len(keys)))
    pipe._trajectory_scaler = scaler            # (tools/long_trajectory.py reads the final scale: did any step overflow?)
    if fixed_scale:      # the parity fixtures: 100 iterations, far below the scaler's growth interval (long runs pass fixed_scale=False)
        assert scaler.get_scale() == init_scale, "a GradScaler skip would shift the trajectory by one iteration"
    ev = synth.trajectory_eval_camera(*cfg["eval_hw"], tag=cfg["tag"])
    img = vm.get_outputs_for_camera_ray_bundle(RayBundle(T(ev["origins"]).to(dev), T(ev["directions"]).to(dev), None))["rgb"]
    img = img.reshape(*cfg["eval_hw"], 3).cpu().numpy()
    evb = rir_bank(cfg["n_rir_eval"], cfg["tag"] + ".eval")
    am.eval()
    stft = []
    for i in range(cfg["n_rir_eval"]):
        out = am.get_outputs_for_camera(None, None, batch_audio={"mic_pose": evb["mic_pose"][i], "source_pose": evb["source_pose"][i],
                                                                 "rot": evb["rot"][i], "data": evb["log_mag"][i].permute(1, 2, 0)})
        stft.append(out["raw_output"].float().cpu().numpy())
    am.train()
    # the same queries with the encoder's BatchNorms on the grid's own statistics (training-mode forward, as the fixture's
    # ``stft_batch_stats``); after the eval-mode pass, since a training-mode forward also updates the running statistics
    stft_bs = []
    with torch.no_grad():
        feat = am.scene_feature()
        tq = torch.arange(cfg["T"], device=dev)
        for i in range(cfg["n_rir_eval"]):
            e = lambda k: evb[k][i].to(dev).reshape(1, 3).expand(cfg["T"], -1)       # noqa: E731
            stft_bs.append(am.field.forward_queries(feat, tq, e("mic_pose"), e("source_pose"), e("rot"), am.aabb, cfg["T"]).float().cpu().numpy())
    return curves, img, {"eval": np.stack(stft), "batch_stats": np.stack(stft_bs)}, pipe, evb


def audio_metrics(am, stft_tcf: np.ndarray, evb, i: int, seed: int = 0):
    """T60 / EDT / C50 (+ the evaluator's other keys) of a predicted log-magnitude STFT [T,C,F] against held-out RIR i."""
    dev = am.aabb.device
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {"raw_output": torch.from_numpy(stft_tcf)}
    batch = {"data": evb["log_mag"][i].permute(1, 2, 0), "waveform": evb["waveforms"][i]}
    return am.get_audio_metrics(out, batch, generator=g)


def metric_table(am, stfts: dict, evb, gt_image=None, images: dict = None):
    """{name: {metric: mean over the held-out RIRs}} for several sets of predicted log-magnitude STFTs [n,T,C,F] of the SAME held-out
    RIRs (HIP run, fp32 oracle, oracle probes) through ONE evaluator (NeRAF_evaluator.py:131-190: T60 / EDT / C50 errors against ground
    truth, seeded Griffin-Lim), plus the log-STFT rel-L2 against ground truth and -- with ``images`` -- the held-out PSNR."""
    gt = evb["log_mag"].numpy()
    out = {}
    for name, st in stfts.items():
        st = np.asarray(st)
        ms = [audio_metrics(am, st[i], evb, i) for i in range(st.shape[0])]
        row = {k: float(np.mean([float(m[k]) for m in ms])) for k in ms[0]}
        row["stft_rel_l2_vs_gt"] = rel_l2(st, gt[:st.shape[0]])
        if images is not None and name in images and gt_image is not None:
            row["psnr_vs_gt_db"] = psnr(images[name], gt_image)
        out[name] = row
    return out


# ---- the metric-level parity rule (G9 / G10 and the default-mode runs) -----------------------------------------------------------
# FROZEN 2026-10-04, round 6, BEFORE any HIP run of this round was looked at; not to be edited after a result (history of the earlier,
# re-derived gates: tests/test_gpu_trajectory.py).  Per metric (every one an error against GROUND TRUTH, or the held-out PSNR):
#   family  = the fixture's oracle runs of the scenario: the fp32 oracle and each of its probes (G9: six runs, G10: three);
#   spread  = max(family) - min(family); for G10 at least G9's spread of the same metric (three runs under-sample the range);
#   PASS    = the HIP figure lies inside [min(family), max(family)], widened by 0.5 x spread on the WORSE side (lower PSNR, higher error)
#             and by 2 x spread on the BETTER side.  The better side is wide on purpose: an error against ground truth lower than any
#             oracle run's is not a parity failure; that bound only catches a broken evaluation (e.g. scoring training data).
GATE_WORSE, GATE_BETTER = 0.5, 2.0
GATE_RULE = "HIP inside oracle-family [min,max] widened 0.5 x spread on the worse side, 2 x on the better side (frozen 2026-10-04)"
GATE_METRICS = {"psnr_vs_gt_db": True, "audio_T60": False, "audio_EDT": False, "audio_C50": False}      # metric -> higher is better


def family_gate(hip: float, family, higher_is_better: bool, floor_spread: float = 0.0) -> dict:
    lo, hi = float(min(family)), float(max(family))
    s = max(hi - lo, float(floor_spread))
    low = lo - (GATE_WORSE if higher_is_better else GATE_BETTER) * s
    high = hi + (GATE_BETTER if higher_is_better else GATE_WORSE) * s
    return {"hip": float(hip), "family_min": lo, "family_max": hi, "spread": s, "low": low, "high": high, "inside": bool(low <= hip <= high)}


def gate_table(m: dict, family_names, hip_name: str = "hip", floor_family: dict = None) -> dict:
    """{metric: family_gate(...)} from a ``metric_table`` result; ``floor_family`` = {metric: [values]} of G9's family (for G10)."""
    out = {}
    for k, hib in GATE_METRICS.items():
        fam = [m[n][k] for n in family_names]
        floor = (max(floor_family[k]) - min(floor_family[k])) if floor_family else 0.0
        out[k] = family_gate(m[hip_name][k], fam, hib, floor)
    return out


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))


def parity_summary(g, curves, img, stft, am, evb):
    """All the numbers of the comparison: HIP (this run) vs oracle (fixture ``g``), each against ground truth, the fixture's
    fp16-rounding band, audio metrics of the eval-branch predictions, loss-curve tails."""
    res = {"steps": int(g["steps"]),
           "psnr_hip_vs_oracle_db": psnr(img, g["image"]),
           "psnr_hip_vs_gt_db": psnr(img, g["gt_image"]), "psnr_oracle_vs_gt_db": psnr(g["image"], g["gt_image"]),
           "stft_rel_l2_hip_vs_oracle": rel_l2(stft["eval"], g["stft"]),
           "stft_rel_l2_hip_vs_gt": rel_l2(stft["eval"], g["gt_stft"]), "stft_rel_l2_oracle_vs_gt": rel_l2(g["stft"], g["gt_stft"]),
           "stft_bs_rel_l2_hip_vs_oracle": rel_l2(stft["batch_stats"], g["stft_batch_stats"]),
           "stft_bs_rel_l2_hip_vs_gt": rel_l2(stft["batch_stats"], g["gt_stft"]),
           "stft_bs_rel_l2_oracle_vs_gt": rel_l2(g["stft_batch_stats"], g["gt_stft"])}
    if "probe_image" in g:
        res["psnr_fp16param_oracle_vs_oracle_db"] = psnr(g["probe_image"], g["image"])
        res["stft_rel_l2_fp16param_oracle_vs_oracle"] = rel_l2(g["probe_stft"], g["stft"])
        res["stft_bs_rel_l2_fp16param_oracle_vs_oracle"] = rel_l2(g["probe_stft_batch_stats"], g["stft_batch_stats"])
    n = stft["eval"].shape[0]
    for tag, hip, ora, prb in (("", stft["eval"], g["stft"], g["probe_stft"] if "probe_stft" in g else None),
                               ("_bs", stft["batch_stats"], g["stft_batch_stats"],
                                g["probe_stft_batch_stats"] if "probe_stft_batch_stats" in g else None)):
        mh = [audio_metrics(am, hip[i], evb, i) for i in range(n)]
        mo = [audio_metrics(am, np.asarray(ora[i]), evb, i) for i in range(n)]
        mp = [audio_metrics(am, np.asarray(prb[i]), evb, i) for i in range(n)] if prb is not None else None
        for k in mh[0]:
            res[f"{k}{tag}_hip"] = float(np.mean([m[k] for m in mh]))
            res[f"{k}{tag}_oracle"] = float(np.mean([m[k] for m in mo]))
            if mp is not None:
                res[f"{k}{tag}_fp16param_oracle"] = float(np.mean([m[k] for m in mp]))
    keys = [str(k) for k in g["keys"]][:5]
    tail = slice(int(g["steps"]) - 20, int(g["steps"]))
    for j, k in enumerate(keys):
        res[f"{k}_tail_hip"] = float(np.nanmean(curves[tail, j]))
        res[f"{k}_tail_oracle"] = float(np.nanmean(np.asarray(g["curves"])[tail, j]))
    return res
