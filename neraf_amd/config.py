"""The NeRAF method specification as code: every value of ``NeRAF/NeRAF_config.py:33-139`` that is part of the drop-in
contract -- model configs with ``_target`` / ``setup()``, the four optimizer groups WITH their schedulers,
``start_step_audio``, loss factors -- in the shape nerfstudio's Trainer consumes (``MethodSpecification(TrainerConfig(...))``,
``config.pipeline.setup(device=...)``, ``Optimizers(config.optimizers, pipeline.get_param_groups())``).

nerfstudio itself is not importable here, so the small config / optimizer-wrapper classes it would provide are restated
[NS-recall] with the same field names; when nerfstudio IS installed a maintainer registers this module exactly like the
reference's.  The numbers are the reference's; the classes they instantiate are the HIP-backed ones."""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Type

import torch


# ---- nerfstudio's config plumbing [NS-recall: configs/base_config.py] ------------------------------------------------------
@dataclass
class InstantiateConfig:
    """``setup(**kwargs)`` instantiates ``_target(self, **kwargs)``."""
    _target: Type = None

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


@dataclass
class SceneBox:
    """nerfstudio's SceneBox: ``aabb`` [2,3] = (min xyz, max xyz)."""
    aabb: torch.Tensor

    def get_normalized_positions(self, positions: torch.Tensor) -> torch.Tensor:      # NeRAF_model.py:541-542
        aabb = self.aabb.to(positions.device)
        return (positions - aabb[0]) / (aabb[1] - aabb[0])


# ---- optimizers / schedulers [NS-recall: engine/optimizers.py, engine/schedulers.py] ---------------------------------------
@dataclass
class AdamOptimizerConfig:
    lr: float = 5e-4
    eps: float = 1e-8
    max_norm: Optional[float] = None
    weight_decay: float = 0.0


@dataclass
class ExponentialDecaySchedulerConfig:
    """Exponential decay with optional warm-up [NS-recall]: during ``warmup_steps`` the rate ramps from ``lr_pre_warmup`` to
    the initial rate (cosine ramp by default), afterwards it interpolates log-linearly to ``lr_final`` at ``max_steps``."""
    lr_pre_warmup: float = 1e-8
    lr_final: Optional[float] = None
    warmup_steps: int = 0
    max_steps: int = 100000
    ramp: str = "cosine"

    def lr_at(self, step: int, lr_init: float) -> float:
        lr_final = self.lr_final if self.lr_final is not None else lr_init
        if step < self.warmup_steps:
            if self.ramp == "cosine":
                return self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * math.sin(0.5 * math.pi * min(max(step / self.warmup_steps, 0.0), 1.0))
            return self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * step / self.warmup_steps
        t = min(max((step - self.warmup_steps) / (self.max_steps - self.warmup_steps), 0.0), 1.0)
        return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)

    def setup(self, optimizer: torch.optim.Optimizer, lr_init: float) -> torch.optim.lr_scheduler.LambdaLR:
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda step: self.lr_at(step, lr_init) / lr_init)


def default_optimizers(start_step_audio: int = 2000) -> Dict[str, Dict[str, Any]]:
    """NeRAF_config.py:115-132, value for value."""
    return {
        "proposal_networks": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                              "scheduler": ExponentialDecaySchedulerConfig(lr_final=0.0001, max_steps=200000)},
        "fields": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                   "scheduler": ExponentialDecaySchedulerConfig(lr_final=0.0001, max_steps=200000)},
        "audio_fields": {"optimizer": AdamOptimizerConfig(lr=1e-4, eps=1e-15),
                         "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-8, max_steps=1000000 + start_step_audio,
                                                                      warmup_steps=start_step_audio)},
        "camera_opt": {"optimizer": AdamOptimizerConfig(lr=1e-3, eps=1e-15),
                       "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=5000)},
    }


class Optimizers:
    """nerfstudio's ``Optimizers`` [NS-recall]: one optimizer + scheduler per parameter-group NAME, built from the method's
    ``optimizers`` dict and ``pipeline.get_param_groups()``; same methods as the Trainer calls (``zero_grad_all``,
    ``optimizer_scaler_step_all``, ``scheduler_step_all``).

    MI355X-side difference: groups whose optimizer settings allow it are stepped by ONE ``FusedAdam`` launch -- ``proposal_networks``
    and ``fields`` share betas / eps, so they become two groups (each with its own learning rate; the bias-correction counter is
    per parameter tensor, csrc/optim.hip, as torch.optim.Adam's) of one fused optimizer; ``audio_fields`` (which also contains the field parameters,
    NeRAF_pipeline.py:487) is a second one, stepped after it exactly as the reference's dict order does.  Parameters that are not
    contiguous fp32 device tensors (``camera_opt``'s 6-vectors live wherever the caller put them) use torch.optim.Adam."""

    def __init__(self, config: Dict[str, Dict[str, Any]], param_groups: Dict[str, List[torch.nn.Parameter]], fused: bool = True):
        from .optim import FusedAdam
        self.config = config
        self.optimizers: Dict[str, torch.optim.Optimizer] = {}
        self.schedulers: Dict[str, Any] = {}
        self.parameters: Dict[str, List[torch.nn.Parameter]] = {}
        self._steppers: List[torch.optim.Optimizer] = []         # unique optimizer objects, in step order
        self._group_of: Dict[str, tuple] = {}                     # name -> (optimizer, index of its param_group)
        names = [n for n in config if n in param_groups and len(param_groups[n]) > 0]
        fusable = lambda n: fused and all(p.is_cuda and p.dtype == torch.float32 for p in param_groups[n])   # noqa: E731
        done = set()
        for n in names:
            if n in done:
                continue
            oc: AdamOptimizerConfig = config[n]["optimizer"]
            if fusable(n):
                mates = [n] + [m for m in names if m not in done and m != n and fusable(m) and config[m]["optimizer"].eps == oc.eps
                               and not (set(map(id, param_groups[m])) & set(map(id, param_groups[n])))]
                # a parameter may be in ONE group of an optimizer: groups sharing tensors (audio_fields holds the field parameters
                # too) stay separate optimizers and step one after the other, as nerfstudio's per-group optimizers do
                chosen, seen = [], set()
                for m in mates:
                    ids = set(map(id, param_groups[m]))
                    if not (ids & seen):
                        chosen.append(m)
                        seen |= ids
                opt = FusedAdam([{"params": param_groups[m], "lr": config[m]["optimizer"].lr, "name": m} for m in chosen], eps=oc.eps)
                for gi, m in enumerate(chosen):
                    self._group_of[m] = (opt, gi)
                    done.add(m)
            else:
                opt = torch.optim.Adam(param_groups[n], lr=oc.lr, eps=oc.eps, weight_decay=oc.weight_decay)
                self._group_of[n] = (opt, 0)
                done.add(n)
            self._steppers.append(opt)
        for n in names:
            opt, gi = self._group_of[n]
            self.optimizers[n] = opt
            self.parameters[n] = param_groups[n]
        # tensors held by two consecutive fused optimizers (the radiance field: "fields", then "audio_fields") get both updates in
        # the later optimizer's launch (one pass over p and g instead of two; bit-identical, neraf_amd/optim.py)
        # (pairs only: an optimizer that is already the second of a pair is not linked onwards)
        for a, b in zip(self._steppers[:-1], self._steppers[1:]):
            if isinstance(a, FusedAdam) and isinstance(b, FusedAdam) and a._dual_from is None and a._defer_to is None:
                a.fuse_shared_updates_into(b)
        # one LambdaLR per optimizer object, one lambda per param group (LambdaLR accepts a list)
        for opt in self._steppers:
            lambdas = []
            members = sorted((gi, n) for n, (o, gi) in self._group_of.items() if o is opt)
            for gi, n in members:
                sc = config[n].get("scheduler")
                lr0 = config[n]["optimizer"].lr
                lambdas.append((lambda step, sc=sc, lr0=lr0: sc.lr_at(step, lr0) / lr0) if sc is not None else (lambda step: 1.0))
            sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambdas)
            for _, n in members:
                self.schedulers[n] = sched
        self._unique_schedulers = list({id(s): s for s in self.schedulers.values()}.values())

    def zero_grad_all(self) -> None:
        for o in self._steppers:
            o.zero_grad(set_to_none=True)

    def optimizer_scaler_step_all(self, grad_scaler) -> None:
        for o in self._steppers:
            grad_scaler.step(o)

    def optimizer_step_all(self) -> None:
        for o in self._steppers:
            o.step()

    def scheduler_step_all(self, step: int) -> None:
        for s in self._unique_schedulers:
            s.step()

    def get_lr(self, name: str) -> float:
        opt, gi = self._group_of[name]
        return float(opt.param_groups[gi]["lr"])

    @property
    def steppers(self) -> List[torch.optim.Optimizer]:
        return list(self._steppers)


# ---- model / pipeline configs -------------------------------------------------------------------------------------------------
@dataclass
class CameraOptimizerConfig(InstantiateConfig):
    mode: str = "off"
    trans_l2_penalty: float = 1e-2
    rot_l2_penalty: float = 1e-3

    def setup(self, num_cameras: int, device=None):
        from .cameras import CameraOptimizer
        m = CameraOptimizer(num_cameras, self.mode, self.trans_l2_penalty, self.rot_l2_penalty)
        return m.to(device) if device is not None else m


@dataclass
class NeRAFVisionModelConfig(InstantiateConfig):
    """NeRAFVisionModelConfig(NerfactoModelConfig) (NeRAF_model.py:48-52) with the nerfacto defaults the hot path reads
    [NS-recall] and NeRAF's overrides (NeRAF_config.py:94-98)."""
    near_plane: float = 0.05
    far_plane: float = 1000.0
    num_proposal_samples_per_ray: tuple = (256, 96)
    num_nerf_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    eval_num_rays_per_chunk: int = 1 << 15                         # NeRAF_config.py:95
    average_init_density: float = 0.01                             # :96
    camera_optimizer: CameraOptimizerConfig = field(default_factory=lambda: CameraOptimizerConfig(mode="SO3xR3"))   # :97

    def __post_init__(self):
        if self._target is None:
            from .vision import NeRAFVisionModel
            self._target = NeRAFVisionModel


def audio_model_config(**kw):
    """NeRAFAudioModelConfig with ``_target`` (the dataclass itself lives next to the model, neraf_amd/model.py)."""
    from .model import NeRAFAudioModelConfig
    return NeRAFAudioModelConfig(**kw)


@dataclass
class NeRAFPipelineConfig(InstantiateConfig):
    """NeRAF_pipeline.py:45-64.  ``datamanager`` / ``audio_datamanager`` are configs with ``setup(device=..., test_mode=...,
    world_size=..., local_rank=...)`` (or ready data-manager objects)."""
    datamanager: Any = None
    audio_datamanager: Any = None
    vision_model: Any = field(default_factory=NeRAFVisionModelConfig)
    audio_model: Any = None
    start_step_audio: int = 2000
    save_eval_audio_path: Optional[str] = None

    def __post_init__(self):
        if self._target is None:
            from .pipeline import NeRAFPipeline
            self._target = NeRAFPipeline.from_config

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


@dataclass
class TrainerConfig:
    """The TrainerConfig fields NeRAF sets (NeRAF_config.py:70-81, :133-135)."""
    method_name: str = "NeRAF"
    experiment_name: str = "FurnishedRoom_NeRAF"
    steps_per_eval_batch: int = 10000
    steps_per_eval_image: int = 10000
    steps_per_eval_all_images: int = 10000
    steps_per_save: int = 20000
    save_only_latest_checkpoint: bool = False
    max_num_iterations: int = 400001
    mixed_precision: bool = True
    data: Optional[Path] = None
    output_dir: Path = Path("./outputs")
    pipeline: NeRAFPipelineConfig = None
    optimizers: Dict[str, Dict[str, Any]] = None
    viewer_num_rays_per_chunk: int = 1 << 15
    vis: str = "tensorboard"


@dataclass
class MethodSpecification:
    config: TrainerConfig
    description: str


MAX_LEN_SOUNDSPACES = {"office_4": 78, "room_2": 84, "frl_apartment_2": 107, "frl_apartment_4": 103, "apartment_2": 86,
                       "apartment_1": 101}                                                        # NeRAF_config.py:43


def make_method(dataset: Optional[str] = None, scene: Optional[str] = None, datamanager=None, audio_datamanager=None) -> MethodSpecification:
    """``NeRAF_method`` (NeRAF_config.py:69-139) for a dataset / scene; like the reference the defaults come from the environment
    variables NeRAF_dataset / NeRAF_scene (:33-39).  Data-manager configs are the caller's (file I/O is out of scope)."""
    dataset = dataset or os.environ.get("NeRAF_dataset", "RAF")
    scene = scene or os.environ.get("NeRAF_scene", "FurnishedRoom")
    start_step_audio = 2000                                                                          # :66
    if dataset == "SoundSpaces":
        fs, max_len, base_dir = 22050, MAX_LEN_SOUNDSPACES[scene], "../data/SoundSpaces"             # :41-45
    else:
        fs, max_len, base_dir = 48000, 0.32, "../data/RAF"                                           # :53-56
    audio = audio_model_config(dataset=dataset, use_grid=True, grid_step=1 / 128, N_features=1024,
                               use_multiple_viewing_directions=True, loss_factor=1e-3, W_field=512, N_freq_stft=257, fs=fs,
                               criterion="SC+SLMSE", max_len=max_len)                                # :99-111
    pipe = NeRAFPipelineConfig(datamanager=datamanager, audio_datamanager=audio_datamanager,
                               vision_model=NeRAFVisionModelConfig(eval_num_rays_per_chunk=1 << 15, average_init_density=0.01,
                                                                   camera_optimizer=CameraOptimizerConfig(mode="SO3xR3")),
                               audio_model=audio, start_step_audio=start_step_audio, save_eval_audio_path=None)
    cfg = TrainerConfig(method_name="NeRAF", experiment_name=scene + "_NeRAF", data=Path(os.path.join(base_dir, scene)),
                        pipeline=pipe, optimizers=default_optimizers(start_step_audio))
    return MethodSpecification(config=cfg, description="NeRAF method.")


TRAIN_NUM_RAYS_PER_BATCH = 4096          # vision datamanager, NeRAF_config.py:87-88
AUDIO_NUM_RAYS_PER_BATCH = 2048          # audio datamanager (STFT slices), :47-48 / :57-58
