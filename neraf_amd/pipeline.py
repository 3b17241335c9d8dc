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


from .datamanagers import FixedBatchDataManager, RIRBankDataManager  # noqa: F401  (re-exported)


_UNITS: Dict[Any, torch.Tensor] = {}


def _unit_scalar(device) -> torch.Tensor:
    """A persistent 0-d one per device: the seed of ``backward`` (torch fills a fresh ones tensor per call otherwise)."""
    t = _UNITS.get(device)
    if t is None:
        t = torch.ones((), dtype=torch.float32, device=device)
        _UNITS[device] = t
    return t


class _ScaledLossSum(torch.autograd.Function):
    """(scale * sum_i loss_i, sum_i loss_i) of 0-d device losses -- Trainer.train_iteration's ``functools.reduce(add, loss_dict.values())``
    and ``grad_scaler.scale(loss)`` [NS-recall] -- as ONE launch (csrc/glue.hip: terms added left to right, as ``reduce`` does) with a
    one-multiply backward, instead of the chain of scalar adds and multiplies those two lines build."""

    @staticmethod
    def forward(ctx, scaler, *losses):
        scale = None
        if scaler is not None and scaler.is_enabled():
            if scaler._scale is None:
                scaler._lazy_init_scale_growth_tracker(losses[0].device)
            scale = scaler._scale.reshape(())
        ctx.scale = scale
        ctx.n = len(losses)
        ctx.set_materialize_grads(False)
        if losses[0].is_cuda and len(losses) <= 12:
            # {scale * sum, sum} in one launch (csrc/glue.hip), terms added left to right as reduce(add, ...) does
            from . import _lib
            from .field import _dev_index, _stream_ptr
            terms = [l.reshape(()).float() for l in losses]
            out = torch.empty(2, dtype=torch.float32, device=losses[0].device)
            dev = _dev_index(out)
            _lib.check(_lib.load().neraf_loss_sum_scale(_lib.ctx(dev), _lib.ptr_array(terms), len(terms),
                                                        scale.data_ptr() if scale is not None else None, out.data_ptr(), _stream_ptr()), dev)
            scaled, total = out[0], out[1]
        else:
            total = torch.stack([l.reshape(()).float() for l in losses]).sum()
            scaled = total * scale if scale is not None else total.clone()
        ctx.mark_non_differentiable(total)
        return scaled, total

    @staticmethod
    def backward(ctx, g, _g_total):
        if g is None:
            return (None,) * (ctx.n + 1)
        if ctx.scale is None:
            gi = g
        elif g.data_ptr() == _unit_scalar(g.device).data_ptr():
            gi = ctx.scale                       # seeded with the persistent one (train_iteration): 1 * scale, no launch.  A view of
                                                 # the scaler's own tensor: shared storage, so autograd never accumulates into it in place
        else:
            gi = g * ctx.scale
        return (None,) + (gi,) * ctx.n


def _write_png(path: str, rgb_u8) -> None:
    """8-bit RGB [H,W,3] -> PNG (stdlib zlib; cv2 / PIL are not in the image): what the reference's cv2.imwrite leaves on disk."""
    import struct
    import zlib
    import numpy as np
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class NeRAFPipeline(nn.Module):
    """Joint radiance + acoustic pipeline (NeRAF_pipeline.py:64-222)."""

    def __init__(self, model: nn.Module, audio_model: nn.Module, datamanager=None, audio_datamanager=None,
                 start_step_audio: int = 2000, world_size: int = 1, local_rank: int = 0, config=None):
        super().__init__()
        self.config = config
        self._model = model                                   # nerfstudio names the (possibly DDP-wrapped) vision model `_model`
        self.audio_model = audio_model
        self.datamanager, self.audio_datamanager = datamanager, audio_datamanager
        self.start_step_audio = start_step_audio               # NeRAF_config.py:66
        self.world_size, self.local_rank = world_size, local_rank
        self.save_eval_audio_path = getattr(config, "save_eval_audio_path", None)
        self.audio_model.spatial_distortion = self.model.field.module.spatial_distortion        # :143
        # :152 (viewer hand-off) -- a plain attribute, as in checkpoint.load_pipeline: registering the audio model as a sub-module of the
        # vision model would duplicate every audio tensor (incl. the 128^3 grid) in the vision state dict and tie their train/eval modes
        object.__setattr__(self.model, "audio_model", self.audio_model)
        self._reducer = None

    @classmethod
    def from_config(cls, config, device="cuda", test_mode: str = "val", world_size: int = 1, local_rank: int = 0, grad_scaler=None):
        """``NeRAFPipeline.__init__(config, device, test_mode, world_size, local_rank, grad_scaler)`` (NeRAF_pipeline.py:86-159), i.e.
        what ``NeRAFPipelineConfig.setup(...)`` runs: set up both data managers, instantiate both models through their configs'
        ``setup`` with the reference's keyword arguments, link spatial distortion / eval data / audio model.  Where the reference
        raises for ``world_size > 1`` (:153-157) this pipeline shards (neraf_amd/parallel.py)."""
        def build(dm):
            return dm.setup(device=device, test_mode=test_mode, world_size=world_size, local_rank=local_rank) if hasattr(dm, "setup") else dm
        datamanager, audio_datamanager = build(config.datamanager), build(config.audio_datamanager)             # :104-109
        seed_pts = None
        tdo = getattr(datamanager, "train_dataparser_outputs", None)
        if tdo is not None and "points3D_xyz" in tdo.metadata:                                                   # :111-118
            seed_pts = (tdo.metadata["points3D_xyz"], tdo.metadata["points3D_rgb"])
        datamanager.to(device)
        audio_datamanager.to(device)
        assert datamanager.train_dataset is not None, "Missing input dataset"                                    # :123
        model = config.vision_model.setup(scene_box=datamanager.train_dataset.scene_box, num_train_data=len(datamanager.train_dataset),
                                          metadata=datamanager.train_dataset.metadata, device=device, grad_scaler=grad_scaler,
                                          seed_points=seed_pts)                                                  # :125-132
        model.to(device)
        audio_kw = dict(scene_box=audio_datamanager.train_dataset.scene_box, num_train_data=len(audio_datamanager.train_dataset),
                        device=device)
        if world_size > 1:
            audio_kw["process_group"] = True
        audio_model = config.audio_model.setup(**audio_kw)                                                       # :135-139
        audio_model.to(device)
        pipe = cls(model, audio_model, datamanager, audio_datamanager, start_step_audio=config.start_step_audio, world_size=world_size,
                   local_rank=local_rank, config=config)
        ev = audio_datamanager.eval_dataset
        if ev is not None and len(ev) > 0:                                                                        # :147
            e0 = ev[0]
            audio_model.set_eval_data(e0["source_pose"], e0["mic_pose"], e0["rot"], e0["data"])
        if world_size > 1:
            pipe.attach_gradient_reducer()
        return pipe

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
        """NeRAF_pipeline.py:231-259: eval-mode outputs, losses and metrics on the next eval batch of each manager."""
        was = self.training
        self.eval()
        try:
            nxt = getattr(self.datamanager, "next_eval", None) or self.datamanager.next_train
            ray_bundle, batch = nxt(step)
            model_outputs = self.model(ray_bundle)
            metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
            gt = (batch["image"] if "image" in batch else batch["rgb"]).to(model_outputs["rgb"].device)
            # NerfactoModel.get_loss_dict outside training: the rgb loss only (interlevel / distortion need the sample lists)
            loss_dict = {"rgb_loss": torch.mean((model_outputs["rgb"] - gt) ** 2)}
            if step > self.start_step_audio:
                nxt_a = getattr(self.audio_datamanager, "next_eval", None) or self.audio_datamanager.next_train
                _, ba = nxt_a(step)
                out_a = self.audio_model.get_outputs(ba)
                metrics_dict.update(self.audio_model.get_metrics_dict(out_a, ba))
                loss_dict.update(self.audio_model.get_loss_dict(out_a, ba, metrics_dict))
            return model_outputs, loss_dict, metrics_dict
        finally:
            self.train(was)

    @torch.no_grad()
    def get_eval_image_metrics_and_images(self, step: int):
        """NeRAF_pipeline.py:261-289: one eval frame (and, once the audio branch runs, one eval RIR)."""
        was = self.training
        self.eval()
        try:
            camera, batch = self.datamanager.next_eval_image(step)
            outputs = self.model.get_outputs_for_camera(camera, None, eval=True)
            metrics_dict, images_dict = self.model.get_image_metrics_and_images(outputs, batch)
            assert "num_rays" not in metrics_dict
            metrics_dict["num_rays"] = camera.height * camera.width * camera.size
            if step > self.start_step_audio:
                _, batch_audio = self.audio_datamanager.next_eval_image(step)
                outputs_audio = self.audio_model.get_outputs_for_camera(None, None, batch_audio=batch_audio)
                m_a, im_a = self.audio_model.get_image_metrics_and_images(outputs_audio, batch_audio)
                metrics_dict.update(m_a)
                images_dict.update(im_a)
            return metrics_dict, images_dict
        finally:
            self.train(was)

    @torch.no_grad()
    def get_average_eval_image_metrics(self, step: Optional[int] = None, output_path=None, get_std: bool = False):
        """NeRAF_pipeline.py:291-436: every eval frame through ``get_outputs_for_camera`` + image metrics, then (when
        ``step > start_step_audio``, or always when an output path is given, :353-354) every eval RIR through the audio model's eval
        branch + audio metrics; means (and stds) over items, throughput keys as the reference names them.

        Data parallel (SURVEY 8e "Eval"): frames and RIRs are dealt round-robin over the ranks (item i -> rank i % world), every
        rank evaluates its share without any collective inside the loop, and the per-item metric rows are all-gathered once at the
        end so that every rank returns the same averages as a single process would."""
        import os
        from time import time
        import numpy as np
        was = self.training
        self.eval()
        try:
            rank, world = (self.local_rank, self.world_size) if self.world_size > 1 else (0, 1)
            if world > 1:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():                       # the GLOBAL rank deals the items
                    rank, world = dist.get_rank(), dist.get_world_size()
            rows_v: List[Dict[str, float]] = []
            for i, (camera, batch) in enumerate(self.datamanager.fixed_indices_eval_dataloader):           # :323
                if i % world != rank:
                    continue
                t0 = time()
                outputs = self.model.get_outputs_for_camera(camera=camera, obb_box=None, eval=True)         # :325
                num_rays = camera.height * camera.width
                metrics_dict, im = self.model.get_image_metrics_and_images(outputs, batch)                  # :328
                if output_path is not None:                                                                # :329-338 (cv2.imwrite)
                    # the reference writes eval_XXXXX.png here and eval_XXXXX.npy for the audio below (:374-380): distinct names
                    arr = (im["img"].detach().cpu().numpy() * 255).astype(np.uint8)
                    _write_png(os.path.join(output_path, f"eval_{str(i).zfill(5)}.png"), arr)
                torch.cuda.synchronize() if torch.cuda.is_available() else None
                metrics_dict["num_rays_per_sec"] = num_rays / (time() - t0)                                 # :341
                metrics_dict["fps"] = metrics_dict["num_rays_per_sec"] / num_rays                           # :343-344
                rows_v.append({k: float(v) for k, v in metrics_dict.items()})
            if output_path is not None:                                                                    # :353-354
                step = self.start_step_audio + 1
            rows_a: List[Dict[str, float]] = []
            if step is not None and step > self.start_step_audio:
                ev = self.audio_datamanager.eval_dataset
                old_mode = getattr(ev, "mode", None)
                if old_mode != "inference":
                    ev.mode = "eval_image"                                                                 # :310-311
                try:
                    # the reference evaluates one RIR (T rows of h) per field call (:355-362); here the rank's RIRs go through the
                    # field in blocks of `eval_rirs_per_call` (N * T rows per call: NeRAFAudioModel.get_outputs_for_rirs), and each
                    # item's output dict, metrics and files are then built exactly as before
                    mine = [i for i in range(len(ev)) if i % world == rank]                                # :355
                    blk = max(int(getattr(self, "eval_rirs_per_call", 32)), 1)
                    dev = self.audio_model.aabb.device
                    for b0 in range(0, len(mine), blk):
                        ids = mine[b0:b0 + blk]
                        t_blk = time()
                        items = [ev[i] for i in ids]                                                       # :360
                        raws = self.audio_model.get_outputs_for_rirs(*(torch.stack([it[k].to(dev).reshape(3) for it in items])
                                                                       for k in ("mic_pose", "source_pose", "rot")))
                        torch.cuda.synchronize() if torch.cuda.is_available() else None
                        # the block's 2 x len(ids) Griffin-Lim reconstructions + metric chains (:364) as one batched reconstruction
                        block_metrics = self.audio_model.get_audio_metrics_block(raws, items)
                        torch.cuda.synchronize() if torch.cuda.is_available() else None
                        share = (time() - t_blk) / len(ids)            # this item's share of the block's field call + metric chain
                        for k, (i, batch) in enumerate(zip(ids, items)):
                            t0 = time()
                            outputs = self.audio_model.eval_outputs_from_raw(raws[k], batch)               # :362
                            metrics_dict = dict(block_metrics[k])
                            if self.save_eval_audio_path is not None:                                          # :366-372
                                d = os.path.join(self.save_eval_audio_path, str(step))
                                os.makedirs(d, exist_ok=True)
                                np.save(os.path.join(d, f"eval_{i}.npy"), {"pred": outputs["raw_output"].detach().cpu().numpy(),
                                                                           **{kk: (v.cpu().numpy() if torch.is_tensor(v) else v) for kk, v in batch.items()}})
                            if output_path is not None:                                                        # :374-380
                                np.save(os.path.join(output_path, f"eval_{str(i).zfill(5)}.npy"),
                                        outputs["raw_output"].permute(1, 2, 0).detach().cpu().numpy())
                            num_rays = batch["data"].shape[-1]                                                 # :382
                            metrics_dict["num_rays_per_sec_audio"] = num_rays / (time() - t0 + share)          # :384
                            metrics_dict["fps_audio"] = metrics_dict["num_rays_per_sec_audio"] / num_rays       # :386-387
                            rows_a.append({kk: float(v) for kk, v in metrics_dict.items()})
                finally:
                    if old_mode != "inference":
                        ev.mode = "eval"                                                                   # :398-399
            if world > 1:
                import torch.distributed as dist
                gathered: List = [None] * world
                dist.all_gather_object(gathered, (rows_v, rows_a))
                rows_v = [r for part in gathered for r in part[0]]
                rows_a = [r for part in gathered for r in part[1]]
            out: Dict[str, float] = {}
            for rows in (rows_v, rows_a):                                                                  # :402-432
                if not rows:
                    continue
                for key in rows[0].keys():
                    vals = torch.tensor([r[key] for r in rows], dtype=torch.float64)
                    if get_std:
                        std, mean = torch.std_mean(vals) if len(rows) > 1 else (torch.zeros(()), vals.mean())
                        out[key], out[f"{key}_std"] = float(mean), float(std)
                    else:
                        out[key] = float(vals.mean())
            return out
        finally:
            self.train(was)

    # ---- optimisation ------------------------------------------------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """NeRAF_pipeline.py:477-490: data-manager groups, the vision model's groups (nerfacto's "proposal_networks", "fields",
        "camera_opt") and the audio group with the field parameters appended to it as well ("Backprop on vision too", :487): they
        are stepped by both optimizers."""
        groups: Dict[str, List[nn.Parameter]] = {}
        for dm in (self.datamanager, self.audio_datamanager):
            if dm is not None and hasattr(dm, "get_param_groups"):
                groups.update(dm.get_param_groups())
        groups.update(self.model.get_param_groups())
        audio = self.audio_model.get_param_groups()
        audio["audio_fields"] = list(audio["audio_fields"]) + list(groups["fields"])
        return {**groups, **audio}

    def make_optimizers(self, init_scale: float = 65536.0, optimizers_config=None, with_schedulers: bool = False):
        """nerfstudio's ``Optimizers`` for NeRAF_config.py:115-132 (Adam eps 1e-15; lr 1e-2 / 1e-2 / 1e-4 / 1e-3, exponential-decay
        schedulers) over ``get_param_groups()`` and the GradScaler (mixed_precision=True, :79).  Returns ([optimizers in step
        order], scaler); ``with_schedulers`` returns the ``neraf_amd.config.Optimizers`` wrapper instead of the bare list (its
        ``scheduler_step_all`` is what the Trainer calls after every iteration)."""
        from .config import Optimizers, default_optimizers
        from .optim import GradScaler
        cfg = optimizers_config if optimizers_config is not None else default_optimizers(self.start_step_audio)
        opts = Optimizers(cfg, self.get_param_groups())
        scaler = GradScaler("cuda", init_scale=init_scale)
        return (opts if with_schedulers else opts.steppers), scaler

    def attach_gradient_reducer(self, group=None):
        """Data parallel: average gradients over the ranks, overlapped with the backward pass (neraf_amd/parallel.py).  The reducer's
        hooks are ARMED by ``train_iteration`` for its own backward pass only; a caller that drives ``get_train_loss_dict`` + backward
        itself sets ``reducer.armed = True`` before the backward and calls ``reducer.finish()`` after it."""
        from .parallel import GradientReducer
        import os
        # The scene encoder's gradients.  Every rank holds the same grid and the same weights and the encoder's backward is linear in
        # d feat, so averaging those 1024 floats (4 KiB) over the ranks BEFORE the backward gives every rank the already-averaged
        # weight gradients and takes the encoder's 17 M parameters (68 MB) off the wire.  That is exact -- and keeps the
        # replicas bit-identical -- iff the encoder's forward and backward are bit-identical on every rank: true under
        # NERAF_DETERMINISTIC=1 (every order-dependent sum of the step is formed in a fixed order, csrc/common.h), where it is the
        # default; with fp32-atomic BatchNorm sums (the default mode) the replicas would drift, and the encoder's gradients are
        # all-reduced like everything else.  NERAF_DP_DFEAT=0 / 1 forces either.
        mode = os.environ.get("NERAF_DP_DFEAT", "auto")
        dfeat = self.audio_model.use_grid and (mode == "1" or (mode == "auto" and os.environ.get("NERAF_DETERMINISTIC") == "1"))
        self.dp_dfeat_allreduce = bool(dfeat)
        groups = [list(self.audio_model.field.parameters()),
                  list(self.audio_model.resnet3d.parameters()) if (self.audio_model.use_grid and not dfeat) else [],
                  list(self.model.field.parameters()), [p for pn in self.model.proposal_networks for p in pn.parameters()],
                  list(self.model.camera_optimizer.parameters()) if hasattr(self.model, "camera_optimizer") else []]
        groups = [g for g in groups if g]
        cmb = os.environ.get("NERAF_DP_COMPRESS_MB")      # e.g. 8: tensors >= 8 MB (the hash-table gradients) are all-reduced in bfloat16
        self._reducer = GradientReducer(groups, group=group, overlap=os.environ.get("NERAF_DP_OVERLAP", "1") != "0",
                                        compress_bytes=int(float(cmb) * (1 << 20)) if cmb else None)
        if dfeat:
            self.audio_model.resnet3d.backbone_net.dp_group = group if group is not None else True
        elif self.audio_model.use_grid:
            # the ResNet3D backward assigns its parameters' gradients itself (no per-parameter autograd hooks fire): it tells the
            # reducer when they are final
            net = self.audio_model.resnet3d.backbone_net
            first = next(net.parameters())
            gi = next(i for i, g in enumerate(groups) if any(p is first for p in g))
            red = self._reducer
            net.grads_ready_hook = lambda: red.notify_group(gi)
        self._reducer.armed = False             # armed by train_iteration for its own backward pass
        return self._reducer

    def train_iteration(self, step: int, optimizers, scaler) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """One Trainer.train_iteration: zero_grad, forward, summed loss, scaled backward, (gradient averaging), optimizer steps."""
        self.model.update_to_step(step)
        wrapper = optimizers if hasattr(optimizers, "scheduler_step_all") else None
        if wrapper is not None:
            optimizers = wrapper.steppers
        for o in optimizers:
            o.zero_grad(set_to_none=True)
        _, loss_dict, _ = self.get_train_loss_dict(step)
        # Trainer.train_iteration: loss = reduce(add, loss_dict.values()); grad_scaler.scale(loss).backward() -- as one node
        scaled, loss = _ScaledLossSum.apply(scaler, *loss_dict.values())
        if self._reducer is not None:
            self._reducer.armed = True          # its hooks act during THIS backward only (see GradientReducer.armed)
        scaled.backward(gradient=_unit_scalar(scaled.device) if scaled.is_cuda else None)
        if self._reducer is not None:
            self._reducer.finish()
            self._reducer.armed = False
        for o in optimizers:
            scaler.step(o)
        scaler.update()
        if wrapper is not None:
            wrapper.scheduler_step_all(step)            # Trainer.train_iteration [NS-recall]: schedulers after the optimizer steps
        return loss, loss_dict

    # ---- checkpoints -------------------------------------------------------------------------------------------------
    def state_dict(self, *args, **kwargs) -> Dict[str, Any]:                                    # :492-497
        return pipeline_state_dict(self.model, self.audio_model)

    def load_pipeline(self, loaded_state: Dict[str, Any], step: int, convert_tcnn: bool = False):    # :438-464
        return _load_pipeline(loaded_state, self.model, self.audio_model, step=step, convert_tcnn=convert_tcnn)
