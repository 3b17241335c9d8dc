"""The hot path as the reference names it: ``NeRAFPipeline.get_train_loss_dict`` (NeRAF_pipeline.py:166-222) -- vision forward +
losses, grid refresh, and (after ``start_step_audio``) the audio forward + losses -- plus the pieces of nerfstudio's Trainer /
Optimizers that turn it into a training iteration (parameter groups NeRAF_pipeline.py:476-489, optimizers NeRAF_config.py:115-127,
GradScaler, checkpoint state :438-464 / :492-497).  Same method names, arguments and return values as the reference class, so its
trainer can drive this one; every tensor operation underneath is the HIP engine (include/neraf_hip.h).

Data managers are anything with ``next_train(step) -> (ray_bundle, batch)``; two small ones are provided: a fixed resident batch
(benchmarks, tests) and the device-resident RIR bank of neraf_amd/data.py."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .checkpoint import load_pipeline as _load_pipeline
from .checkpoint import pipeline_state_dict


class FixedBatchDataManager:
    """``next_train`` returns the same resident (ray_bundle, batch) pair every step."""

    def __init__(self, ray_bundle, batch: Dict[str, torch.Tensor], train_num_rays_per_batch: Optional[int] = None):
        self.ray_bundle, self.batch = ray_bundle, batch
        if train_num_rays_per_batch is None:
            train_num_rays_per_batch = len(ray_bundle) if ray_bundle is not None else 4096
        self.train_num_rays_per_batch = train_num_rays_per_batch

    def next_train(self, step: int):
        return self.ray_bundle, self.batch

    def get_param_groups(self):
        return {}


class RIRBankDataManager:
    """Audio batches sampled on the device from a ``DeviceRIRBank`` (neraf_amd/data.py); the ray bundle slot is None as in
    NeRAFDataManager.next_train (the audio model needs no rays, NeRAF_pipeline.py:187)."""

    def __init__(self, bank, batch_size: int = 2048, generator: Optional[torch.Generator] = None):
        self.bank, self.batch_size, self.generator = bank, batch_size, generator

    def next_train(self, step: int):
        return None, self.bank.next_train(self.batch_size, generator=self.generator)

    def get_param_groups(self):
        return {}


class _ScaledLossSum(torch.autograd.Function):
    """(scale * sum_i loss_i, sum_i loss_i) of 0-d device losses in three small launches (stack, sum, multiply) with a one-launch
    backward, instead of the chain of scalar adds and multiplies the Trainer's two lines build."""

    @staticmethod
    def forward(ctx, scaler, *losses):
        stacked = torch.stack([l.reshape(()).float() for l in losses])
        total = stacked.sum()
        if scaler is not None and scaler.is_enabled():
            if scaler._scale is None:
                scaler._lazy_init_scale_growth_tracker(total.device)
            scale = scaler._scale.reshape(())
            scaled = total * scale
        else:
            scale, scaled = None, total.clone()
        ctx.scale = scale
        ctx.n = len(losses)
        ctx.mark_non_differentiable(total)
        return scaled, total

    @staticmethod
    def backward(ctx, g, _g_total):
        gi = g if ctx.scale is None else g * ctx.scale
        return (None,) + (gi,) * ctx.n


class NeRAFPipeline(nn.Module):
    """Joint radiance + acoustic pipeline (NeRAF_pipeline.py:64-222)."""

    def __init__(self, model: nn.Module, audio_model: nn.Module, datamanager=None, audio_datamanager=None,
                 start_step_audio: int = 2000, world_size: int = 1):
        super().__init__()
        self._model = model                                   # nerfstudio names the (possibly DDP-wrapped) vision model `_model`
        self.audio_model = audio_model
        self.datamanager, self.audio_datamanager = datamanager, audio_datamanager
        self.start_step_audio = start_step_audio               # NeRAF_config.py:66
        self.world_size = world_size
        self.audio_model.spatial_distortion = self.model.field.module.spatial_distortion        # :143
        self._reducer = None

    @property
    def model(self):
        return self._model

    @property
    def device(self):
        return self.model.device

    # ---- the hot path ------------------------------------------------------------------------------------------------
    def get_train_loss_dict(self, step: int):
        ray_bundle, batch = self.datamanager.next_train(step)
        model_outputs = self._model(ray_bundle)                                                  # :176
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)                # :178
        if self.audio_model.use_grid:                                                            # :181-184
            self.audio_model.query_grid_one_batch(step, self.model.field, renderer_rgb=self.model.renderer_rgb,
                                                  batch_size=getattr(self.datamanager, "train_num_rays_per_batch", 4096))
        if step > self.start_step_audio:                                                         # :186-199
            _, batch_audio = self.audio_datamanager.next_train(step)
            model_audio_outputs = self.audio_model.get_outputs(batch_audio)
            loss_dict.update(self.audio_model.get_loss_dict(model_audio_outputs, batch_audio, {}))
        return model_outputs, loss_dict, metrics_dict

    def forward(self):
        raise NotImplementedError("call get_train_loss_dict / get_eval_loss_dict (the reference's forward is blank too, :224)")

    @torch.no_grad()
    def get_eval_loss_dict(self, step: int):
        """Eval-mode losses on the next training batch of each manager (the reference evaluates its eval split the same way)."""
        was = self.training
        self.eval()
        try:
            ray_bundle, batch = self.datamanager.next_train(step)
            out = self.model.get_outputs(ray_bundle)
            res = {"rgb_mse": torch.mean((out["rgb"] - batch["image"].to(out["rgb"].device)) ** 2)}
            if step > self.start_step_audio:
                _, ba = self.audio_datamanager.next_train(step)
                res.update(self.audio_model.get_loss_dict(self.audio_model.get_outputs(ba), ba))
            return res
        finally:
            self.train(was)

    # ---- optimisation ------------------------------------------------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """nerfacto's groups plus the audio group, with the field parameters appended to the audio group as well
        ("Backprop on vision too", NeRAF_pipeline.py:487): they are stepped by both optimizers."""
        fields = list(self.model.field.parameters())
        groups = {"proposal_networks": [p for pn in self.model.proposal_networks for p in pn.parameters()], "fields": fields}
        audio = self.audio_model.get_param_groups()
        audio["audio_fields"] = list(audio["audio_fields"]) + fields
        return {**groups, **audio}

    def make_optimizers(self, init_scale: float = 65536.0):
        """FusedAdam equivalents of NeRAF_config.py:115-127 (Adam eps 1e-15; lr 1e-2 / 1e-2 / 1e-4) and the GradScaler.  The
        first two optimizers share hyper-parameters and run as two groups of one launch."""
        from .optim import FusedAdam, GradScaler
        g = self.get_param_groups()
        opt = FusedAdam([{"params": g["proposal_networks"], "lr": 1e-2}, {"params": g["fields"], "lr": 1e-2}], eps=1e-15)
        opt_audio = FusedAdam([{"params": g["audio_fields"], "lr": 1e-4}], eps=1e-15)
        return [opt, opt_audio], GradScaler("cuda", init_scale=init_scale)

    def attach_gradient_reducer(self, group=None):
        """Data parallel: average gradients over the ranks, overlapped with the backward pass (neraf_amd/parallel.py)."""
        from .parallel import GradientReducer
        groups = [list(self.audio_model.field.parameters()), list(self.audio_model.resnet3d.parameters()) if self.audio_model.use_grid else [],
                  list(self.model.field.parameters()), [p for pn in self.model.proposal_networks for p in pn.parameters()]]
        groups = [g for g in groups if g]
        self._reducer = GradientReducer(groups, group=group)
        if self.audio_model.use_grid:
            # the ResNet3D backward assigns its parameters' gradients itself (no per-parameter autograd hooks fire): it tells the
            # reducer when they are final
            net = self.audio_model.resnet3d.backbone_net
            first = next(net.parameters())
            gi = next(i for i, g in enumerate(groups) if any(p is first for p in g))
            red = self._reducer
            net.grads_ready_hook = lambda: red.notify_group(gi)
        return self._reducer

    def train_iteration(self, step: int, optimizers, scaler) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """One Trainer.train_iteration: zero_grad, forward, summed loss, scaled backward, (gradient averaging), optimizer steps."""
        self.model.update_to_step(step)
        for o in optimizers:
            o.zero_grad(set_to_none=True)
        _, loss_dict, _ = self.get_train_loss_dict(step)
        # Trainer.train_iteration: loss = reduce(add, loss_dict.values()); grad_scaler.scale(loss).backward() -- as one node
        scaled, loss = _ScaledLossSum.apply(scaler, *loss_dict.values())
        scaled.backward()
        if self._reducer is not None:
            self._reducer.finish()
        for o in optimizers:
            scaler.step(o)
        scaler.update()
        return loss, loss_dict

    # ---- checkpoints -------------------------------------------------------------------------------------------------
    def state_dict(self, *args, **kwargs) -> Dict[str, Any]:                                    # :492-497
        return pipeline_state_dict(self.model, self.audio_model)

    def load_pipeline(self, loaded_state: Dict[str, Any], step: int):                           # :438-464
        return _load_pipeline(loaded_state, self.model, self.audio_model, step=step)
