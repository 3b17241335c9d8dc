"""Host-side mirror of ``NeRAFAudioModel`` (NeRAF_model.py:104-805) on libneraf_hip.

Keeps the reference's config fields (``NeRAFAudioModelConfig`` :83-101), method names and state-dict key
prefixes (``field.soundfield.*``, ``field.STFT_linear.*``, ``resnet3d.backbone_net.*``, ``grid``), so that
``NeRAFPipeline`` (NeRAF_pipeline.py:135-222) can drive it unchanged:

    query_grid_one_batch(step, vision_field, renderer_rgb, batch_size)   :294-407
    get_outputs(batch_audio) -> Tensor [B,C,F]                           :531-566
    get_loss_dict(outputs, batch)                                        :584-600
    get_outputs_for_camera(None, None, batch_audio) -> dict              :610-728  (eval branch, camera=None)
    get_param_groups() -> {"audio_fields": [...]}                        :730-737

Everything numeric runs in HIP kernels through the C ABI; there is no fallback without the GPU library.
Differences from the reference that are deliberate (and cheaper):
  * the voxel grid and the refresh frustums live on the device (the reference builds them on the CPU and
    copies 73,728 frustums per step, :311-337);
  * the ResNet3D feature is cached between calls while the grid and the encoder weights are unchanged
    (the reference recomputes it for every eval RIR, :680-684, although the grid is static).
The audio loss trains the NAcF MLP, the ResNet3D and -- through the grid cells refreshed in the same step
(NeRAF_model.py:395-400, "Backprop on vision too" NeRAF_pipeline.py:487) -- the radiance field: the refresh is an
autograd node (``_RefreshFn``) whose values the ResNet3D node receives as ``window_vals``.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .config import InstantiateConfig
from .evaluator import GriffinLim, RAFEvaluator, SoundSpacesEvaluator
from .field import NeRAFAudioSoundField, _dev_index, _stream_ptr
from .losses import STFTLoss
from .resnet3d import ResNet3D_helper
from .vision import FieldHeadNames, Frustums, RaySamples


@dataclass
class NeRAFAudioModelConfig(InstantiateConfig):
    """Fields and defaults of the reference config (NeRAF_model.py:88-101); ``setup(scene_box=..., num_train_data=..., device=...)``
    instantiates the model as NeRAF_pipeline.py:135-139 does."""
    dataset: str = "SoundSpaces"
    use_grid: bool = True
    grid_step: float = 1 / 128
    N_features: int = 1024
    use_multiple_viewing_directions: bool = True
    loss_factor: float = 1e-3
    max_len: float = 76
    W_field: int = 512
    fs: int = 22050
    criterion: str = "SC+SLMSE"
    N_freq_stft: int = 257
    hop_len: int = 128
    win_len: int = 512

    def __post_init__(self):
        if self._target is None:
            self._target = NeRAFAudioModel


class _few_host_threads:
    """The metric chain's host side is dozens of torch CPU operations on 30 k-element tensors (exp, log, a 61-frame STFT, L1 means):
    on a 128-thread host every one of them wakes the whole intra-op pool -- measured 94 ms per RIR with 128 threads, 17.6 ms with 8,
    16.4 ms with 1 (tools/eval_metrics_threads_probe.py), i.e. 80 % of ``get_average_eval_image_metrics``' time per RIR was thread
    wake-ups.  Caps the pool at 8 threads for the duration of the block and restores it.  Element-wise results are unchanged; the
    room-acoustic metrics (T60 / EDT / C50) are numpy on the Griffin-Lim waveform and do not depend on it at all."""

    def __init__(self, cap: int = 8):
        self.cap = cap

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.cap:
            torch.set_num_threads(self.cap)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


class _RefreshFn(torch.autograd.Function):
    """vals [4, n] = (mean_dirs rgb, alpha) of the refreshed cells as a function of the radiance-field parameters
    (NeRAF_model.py:339-357, :386).  Forward = fused field query in AABB mode; backward = fused field backward."""

    @staticmethod
    def forward(ctx, field, coords, aabb, dirs, nd: int, delta: float, consts, slab, *params: torch.Tensor):
        """coords [n,3] cell centres in (0,1)^3 (a window of coordinates_to_render), aabb the radiance field's box, dirs [nd,3];
        consts = (dd [n*nd,3], z [n*nd,2], cam [n*nd]) are the per-shape constants of the query; slab = (grid tensor, first cell) or None:
        the launch that forms the values also writes them into the grid window (the detached write of NeRAF_model.py:395-400).  The nd*n queries are laid out
        CELL-major (the nd directions of a cell adjacent) -- the result does not depend on the order, and the backward's
        hash-gradient pre-reduction then merges the nd identical positions into one update per table entry."""
        lib = _lib.load()
        n = coords.shape[0]
        dev = _dev_index(coords)
        dd, z, cam = consts
        oris = torch.empty((n * nd, 3), dtype=torch.float32, device=coords.device)
        _lib.check(lib.neraf_refresh_origins(_lib.ctx(dev), coords.data_ptr(), n, nd, _lib.host_f32(aabb), oris.data_ptr(),
                                             _stream_ptr()), dev)                          # :315, :327-333
        packed = field.packed(with_average=False)       # the refresh queries with camera index 0's embedding (:334)
        rgb, den, saved = field.query(oris, dd, z, cam, use_average_embedding=False, packed=packed, save=1)    # [n*nd,1,3], [n*nd,1]
        vals = torch.empty((4, n), dtype=torch.float32, device=coords.device)
        grid, start = slab if slab is not None else (None, 0)
        nvox = grid.shape[1] * grid.shape[2] * grid.shape[3] if grid is not None else 0
        _lib.check(lib.neraf_grid_refresh_vals(_lib.ctx(dev), rgb.data_ptr(), den.data_ptr(), n, nd, 1, delta, vals.data_ptr(),
                                               grid.data_ptr() if grid is not None else None, nvox, int(start), _stream_ptr()), dev)   # :352-357, :386
        ctx.field, ctx.nd, ctx.delta, ctx.packed, ctx.dev, ctx.saved = field, nd, delta, packed, dev, saved
        # the caller switches the scene contraction off for the duration of THIS call only (NeRAF_model.py:302, :407): the backward
        # runs after it has been switched back on and must map positions as the forward did
        ctx.contract = field.spatial_distortion is not None
        ctx.save_for_backward(oris, dd, z, cam, den)
        return vals

    @staticmethod
    def backward(ctx, dvals: torch.Tensor):
        oris, dd, z, cam, den = ctx.saved_tensors
        nd, delta = ctx.nd, ctx.delta
        n = oris.shape[0] // nd
        lib = _lib.load()
        dvals = dvals.float().contiguous()
        d_rgb = torch.empty((nd * n, 1, 3), dtype=torch.float32, device=oris.device)
        d_den = torch.empty((nd * n, 1), dtype=torch.float32, device=oris.device)
        _lib.check(lib.neraf_grid_refresh_vals_bwd(_lib.ctx(ctx.dev), dvals.data_ptr(), den.data_ptr(), n, nd, 1, delta,
                                                   d_rgb.data_ptr(), d_den.data_ptr(), _stream_ptr()), ctx.dev)
        grads = ctx.field.backward_query(ctx.packed, oris, dd, z, cam, den, d_rgb, d_den, pos_run=nd, saved=ctx.saved,
                                         in_autograd=all(ctx.needs_input_grad[8:15]),          # cell-major: nd rays per position
                                         contract=ctx.contract)
        return (None, None, None, None, None, None, None, None, *grads)


class NeRAFAudioModel(nn.Module):
    def __init__(self, config: NeRAFAudioModelConfig, aabb: torch.Tensor = None, process_group=None, scene_box=None,
                 num_train_data: int = 0, device=None, **kwargs):
        """``aabb`` [2,3] (or ``scene_box.aabb``, the nerfstudio ``Model.__init__(config, scene_box, num_train_data, **kwargs)``
        form used by ``config.setup``, NeRAF_pipeline.py:135-139) is the audio scene box: mic bounding box +- 1 m
        (NeRAF_dataparser.py:155-161)."""
        super().__init__()
        self.config = config
        if aabb is None:
            if scene_box is None:
                raise ValueError("NeRAFAudioModel needs the audio scene box (aabb=... or scene_box=...)")
            aabb = scene_box.aabb
        self.scene_box, self.num_train_data = scene_box, num_train_data
        self.register_buffer("aabb", torch.as_tensor(aabb).float())
        self.dataset = config.dataset
        if self.dataset == "RAF":                                   # default_RAF_config, :109-119, :126-129
            config.fs, config.max_len = 48000, 0.32
            config.N_freq_stft, config.hop_len, config.win_len = 513, 256, 512
            self.max_len = int(config.max_len * config.fs) // config.hop_len
            self.mic_ch = 1
            self.evaluator = RAFEvaluator(fs=config.fs)              # :130
        else:                                                       # :131-134
            self.max_len = int(config.max_len)
            self.mic_ch = 2
            self.evaluator = SoundSpacesEvaluator(fs=config.fs)
        self.istft_transform = GriffinLim(n_fft=(config.N_freq_stft - 1) * 2, win_length=config.win_len, hop_length=config.hop_len,
                                          power=1)                  # :139
        self.use_grid = config.use_grid
        self.loss_factor = config.loss_factor
        self.process_group = process_group
        self.criterion_name = config.criterion
        if self.criterion_name == "MSE":                            # :141-149
            self.criterion = None
        else:
            self.criterion = STFTLoss(loss_type="mse" if "MSE" in self.criterion_name else "l1", process_group=process_group)
        self.spatial_distortion = None                              # :155, set by the pipeline (NeRAF_pipeline.py:143)
        n_query = 21 + 63 + 63 + 16                                 # :169-171
        if self.use_grid:
            self.grid_size = np.array([0, 1, 0, 1, 0, 1])
            self.grid_step = config.grid_step
            self.N_features = config.N_features
            self.resnet3d = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=self.grid_step,
                                            N_features=self.N_features)                                    # :185
            self.field = NeRAFAudioSoundField(self.N_features + n_query, config.W_field, sound_rez=self.mic_ch,
                                              N_frequencies=config.N_freq_stft)                            # :189
            self._delta = 1e-2                                                                             # :191
            # a buffer, so that it follows .to(device): a CPU tensor copied per step is a blocking, stream-ordered H2D copy
            self.register_buffer("view_dirs", self._generate_fixed_viewing_directions()
                                 if config.use_multiple_viewing_directions else None, persistent=False)
            S = self.resnet3d.backbone_net.grid_size
            ax = torch.arange(0 + self.grid_step / 2, 1, self.grid_step)                                   # :200
            coords = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).view(-1, 3)
            self.register_buffer("coordinates_to_render", coords, persistent=False)                       # :202
            self.grid_batch_i = 0                                                                          # :203
            self.register_buffer("grid", self._fresh_grid(S, ax), persistent=True)   # the pipeline checkpoints it (NeRAF_pipeline.py:492-497)
        else:
            self.field = NeRAFAudioSoundField(n_query, config.W_field, sound_rez=self.mic_ch, N_frequencies=config.N_freq_stft)  # :207
        self._feat_cache = None
        self._feat_key = None
        self._window = None
        self._grid_gen, self._grid_dirty = 0, None          # write generation of the grid; (gen, gen before, first cell, n cells) of the last write
        self.eval_source_pose = self.eval_mic_pose = self.eval_rot = self.eval_gt = None

    # ---- A2: grid --------------------------------------------------------------------------------
    @staticmethod
    def _fresh_grid(S: int, ax: torch.Tensor) -> torch.Tensor:
        g = torch.zeros((7, S, S, S), dtype=torch.float32)
        g[4:] = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=0)          # :275-277
        return g

    def reset_grid(self, device=None):                                                  # :269-277
        S = self.grid.shape[1]
        ax = torch.arange(0 + self.grid_step / 2, 1, self.grid_step)
        self.grid.copy_(self._fresh_grid(S, ax).to(self.grid.device))
        self._feat_key = None
        self.mark_grid_written()

    def mark_grid_written(self, cell_start: int = None, n_cells: int = None, version_before: int = None):
        """Every write to ``self.grid`` goes through here: the ResNet3D host layer keeps the grid's fp16 channels-last image between
        steps and re-converts only the window a refresh wrote -- which it may do only while it has seen every generation.  A write
        of unknown extent (reset, checkpoint load, anything from outside) passes no window and forces the next full conversion.
        ``version_before`` / the tensor's current version bracket the write, so that a torch in-place operation on the grid by anybody
        else (which bumps the version counter) is noticed too; writes through a raw pointer are the caller's to report."""
        before = self._grid_gen
        self._grid_gen = before + 1
        self._grid_dirty = ((self._grid_gen, before, int(cell_start), int(n_cells), int(version_before), int(self.grid._version))
                            if cell_start is not None and version_before is not None else None)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        if getattr(self, "use_grid", False):
            self.mark_grid_written()
            self._feat_key = None

    def _generate_fixed_viewing_directions(self) -> torch.Tensor:                       # :279-292, reproduced literally
        phis = [math.pi / 3, 0, -math.pi]
        thetas = [k * math.pi / 3 for k in range(0, 6)]
        v = [torch.Tensor([math.cos(p) * math.sin(t), math.cos(p) * math.sin(t), math.sin(t)]) for p in phis for t in thetas]
        return torch.stack(v, dim=0)

    # ---- A3: refresh -----------------------------------------------------------------------------
    def query_grid_one_batch(self, step, vision_field, renderer_rgb=None, batch_size=4096):
        if not self.use_grid:
            return
        lib = _lib.load()
        self._window = None
        module = vision_field.module
        saved = module.spatial_distortion
        module.spatial_distortion = None                                                # :302
        try:
            aabb = module.aabb
            n_cells = self.coordinates_to_render.shape[0]
            i = self.grid_batch_i
            if i + batch_size > n_cells:                                                # :308-309
                batch_size = n_cells - i
            coords = self.coordinates_to_render[i:i + batch_size]
            if self.view_dirs is not None:
                dirs = self.view_dirs.to(coords.device)
            else:
                if getattr(self, "_one_dir", None) is None or self._one_dir.device != coords.device:
                    self._one_dir = torch.tensor([[1.0, 0.0, 0.0]], device=coords.device)
                dirs = self._one_dir
            nd = dirs.shape[0]
            nvox = self.grid.shape[1] * self.grid.shape[2] * self.grid.shape[3]
            differentiable = (self.training and torch.is_grad_enabled() and renderer_rgb is not None
                              and any(p.requires_grad for p in module.grad_params()))
            dp = self._dp_world()
            if differentiable:
                # the grid is detached and the fresh values keep their graph (:395-400): vals is an autograd node over the
                # radiance-field parameters, consumed by the ResNet3D node in scene_feature()
                vb = self.grid._version
                if dp is None:
                    # the kernel that forms the values writes the grid window too (a raw-pointer write: the version counter stays)
                    vals = _RefreshFn.apply(module, coords, aabb, dirs, nd, self._delta, self._refresh_consts(dirs, batch_size),
                                            (self.grid, i), *module.grad_params())
                else:
                    # data parallel (SURVEY 8e "Partitioning"): every rank holds the same field, so rank r queries only its
                    # share of the window's cells (and back-propagates only through those); the shares are assembled into the
                    # same [4, n] values on every rank -- the grid stays bit-identical across replicas (the field query has no
                    # atomics) at 1/world of the refresh work per rank
                    from .parallel import gather_shards, shard_range
                    group, rank, world = dp
                    lo, hi = shard_range(batch_size, rank, world)
                    local = _RefreshFn.apply(module, coords[lo:hi], aabb, dirs, nd, self._delta, self._refresh_consts(dirs, hi - lo),
                                             None, *module.grad_params()) if hi > lo else torch.zeros((4, 0), device=coords.device)
                    vals = gather_shards(local, lo, hi, batch_size, group)
                    with torch.no_grad():
                        self.grid.view(7, nvox)[0:4, i:i + batch_size] = vals
                self.mark_grid_written(i, batch_size, vb)
                self._window = (i, batch_size, vals)
            else:
                with torch.no_grad():
                    ori = coords * (aabb[1] - aabb[0]) + aabb[0]                            # :315
                    oris = ori.repeat(nd, 1)                                                # direction-major concat, :327-333
                    dd = dirs.repeat_interleave(batch_size, dim=0)
                    z = torch.zeros((oris.shape[0], 1), device=ori.device)
                    rs = RaySamples(Frustums(oris, dd, z, z), torch.zeros((oris.shape[0], 1), dtype=torch.int32, device=ori.device))
                    was = module.training
                    module.train(True)           # the reference queries with camera index 0's appearance embedding (:334)
                    out = vision_field.forward(rs)                                      # :339
                    module.train(was)
                    rgb = out[FieldHeadNames.RGB].contiguous()
                    den = out[FieldHeadNames.DENSITY].reshape(-1).contiguous()
                    # renderer_rgb with weights == 1 on a single sample returns rgb unchanged (:344-350); without a renderer
                    # the reference applies a sigmoid to the (already sigmoid) colour (:390)
                    if renderer_rgb is None:
                        rgb = torch.sigmoid(rgb)
                    dev = _dev_index(rgb)
                    vb = self.grid._version
                    _lib.check(lib.neraf_grid_refresh_write(_lib.ctx(dev), rgb.data_ptr(), den.data_ptr(), batch_size, nd, self._delta,
                                                            self.grid.data_ptr(), nvox, i, _stream_ptr()), dev)
                    self.mark_grid_written(i, batch_size, vb)
            self.grid_batch_i += batch_size                                             # :402-404
            if self.grid_batch_i >= n_cells:
                self.grid_batch_i = 0
            self._feat_key = None
        finally:
            module.spatial_distortion = saved                                           # :407

    def _dp_world(self):
        """(group, rank, world) when the model was built with a process group of more than one rank, else None."""
        pg = getattr(self, "process_group", None)
        if pg is None:
            return None
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized():
            return None
        group = None if pg is True else pg
        world = dist.get_world_size(group)
        # ``dp_single_rank_collectives`` (tests): keep the data-parallel code path -- sharded refresh, assembled through the collectives --
        # also in a group of ONE rank, which exercises it on the real backend of a one-GPU box
        return (group, dist.get_rank(group), world) if (world > 1 or getattr(self, "dp_single_rank_collectives", False)) else None

    def _refresh_consts(self, dirs, n):
        """(directions [n*nd,3] cell-major, zero frustum extents [n*nd,2], camera index 0 [n*nd]) of a refresh window: constant
        per (n, directions), built once."""
        key = (n, dirs.data_ptr(), dirs._version, str(dirs.device))
        c = getattr(self, "_refresh_const_cache", None)
        if c is None or c[0] != key:
            nd = dirs.shape[0]
            c = (key, (dirs.repeat(n, 1).contiguous(), torch.zeros((n * nd, 2), device=dirs.device),
                       torch.zeros(n * nd, dtype=torch.int32, device=dirs.device)))
            self._refresh_const_cache = c
        return c[1]

    # ---- A4: scene feature -----------------------------------------------------------------------
    def scene_feature(self) -> torch.Tensor:
        """ResNet3D(grid) -> [1024].  Cached in eval mode while the grid is untouched."""
        if not self.training and self._feat_key is not None and self._feat_cache is not None:
            return self._feat_cache
        win = getattr(self, "_window", None)
        d = self._grid_dirty
        gs = d if (d is not None and d[0] == self._grid_gen) else (self._grid_gen, -1, 0, 0, -1, -1)
        if self.training and win is not None:
            out = self.resnet3d(self.grid.unsqueeze(0), window=(win[0], win[1], 4), window_vals=win[2], grid_state=gs)
            self._window = None
        else:
            out = self.resnet3d(self.grid.unsqueeze(0), grid_state=gs)                  # :554-557
        feat = out.flatten()
        gb = getattr(out, "_neraf_grad_buffer", None)
        if gb is not None:                   # where the encoder's backward wants d loss / d feature written (ResNet3D.dfeat_buffer)
            feat._neraf_grad_buffer = gb
        if not self.training:
            self._feat_cache, self._feat_key = feat, True
        return feat

    # ---- A1 + A5 ---------------------------------------------------------------------------------
    def get_outputs(self, batch_audio: Dict[str, torch.Tensor]) -> torch.Tensor:        # :531-566
        dev = self.aabb.device
        feat = self.scene_feature() if self.use_grid else torch.zeros(0, device=dev)
        return self.field.forward_queries(feat, batch_audio["time_query"].to(dev), batch_audio["mic_pose"].to(dev),
                                          batch_audio["source_pose"].to(dev), batch_audio["rot"].to(dev), self.aabb, self.max_len)

    def get_metrics_dict(self, outputs, batch):
        return {}

    def get_loss_dict(self, outputs, batch, metrics_dict=None):                         # :584-600
        gt = batch["data"].to(outputs.device).float()
        pred = outputs.float()
        if self.criterion_name == "MSE":
            return {"audio_mse": torch.nn.functional.mse_loss(pred, gt) * self.loss_factor}
        # sc * 1e-1 * loss_factor, mag * 1.0 * loss_factor (:592-599) -- the products of the constants are formed in fp32 like the
        # reference's two successive multiplications only up to 1 ulp; applied inside the loss node (one kernel each way)
        w = getattr(self, "_loss_w", None)
        if w is None or w.device != pred.device:
            w = torch.tensor([1e-1 * self.loss_factor, 1.0 * self.loss_factor], dtype=torch.float32, device=pred.device)
            self._loss_w = w
        return self.criterion(pred, gt, w)

    def set_eval_data(self, eval_source_pose, eval_mic_pose, eval_rot, eval_gt):        # :602-607
        # plain attributes, set once per evaluated RIR: nn.Module.__setattr__'s parameter / buffer / module bookkeeping is 10 us for the four
        self.__dict__.update(eval_source_pose=eval_source_pose, eval_mic_pose=eval_mic_pose, eval_rot=eval_rot, eval_gt=eval_gt)

    # ---- A7: eval branch -------------------------------------------------------------------------
    @torch.no_grad()
    def get_outputs_for_camera(self, camera, obb_box, batch_audio=None):                # :610-728 (camera=None branch)
        if camera is not None:
            raise NotImplementedError("viewer camera branch (:611-646) is UI code, out of scope (SURVEY.md row 15)")
        dev = self.aabb.device
        T = self.max_len
        tq = getattr(self, "_eval_tq", None)                                            # :649 (the same T indices for every RIR)
        if tq is None or tq.numel() != T or tq.device != dev:
            tq = self._eval_tq = torch.arange(0, T, 1, device=dev)
        # one (microphone, source, orientation) for all T time queries (:650-652, :676-678 expand them to [T,3]; the prologue kernel
        # reads the single row for every query instead)
        mic = batch_audio["mic_pose"].to(dev).reshape(1, 3)
        src = batch_audio["source_pose"].to(dev).reshape(1, 3)
        rot = batch_audio["rot"].to(dev).reshape(1, 3)
        feat = self.scene_feature() if self.use_grid else torch.zeros(0, device=dev)
        out = self.field.forward_queries(feat, tq, mic, src, rot, self.aabb, T)         # [T,C,F]
        return self.eval_outputs_from_raw(out, batch_audio)

    @torch.no_grad()
    def eval_outputs_from_raw(self, out: torch.Tensor, batch_audio) -> Dict[str, torch.Tensor]:
        """The output dict of the eval branch (:695-726) from the raw log-magnitudes ``out`` [T,C,F] of one RIR -- the per-channel panels,
        the ground-truth panels, the grid views.  ``get_outputs_for_camera`` ends here; a caller that evaluated many RIRs in one
        field call (``get_outputs_for_rirs``) builds each item's dict with it."""
        self.set_eval_data(batch_audio["mic_pose"], batch_audio["source_pose"], batch_audio["rot"], batch_audio["data"])    # :653
        dev = self.aabb.device
        stft: Dict[str, torch.Tensor] = {}
        host = out.cpu()                           # ONE device-to-host copy of [T,C,F]; the panels below are host-side views of it
        for ch in range(out.shape[1]):                                                  # :695-700
            stft["stft_ch_" + str(ch)] = torch.flip(host[:, ch, :].transpose(0, 1).unsqueeze(-1), [0])
        gt = self.eval_gt.cpu()
        for ch in range(gt.shape[0]):                                                   # :703-714
            stft["gt_ch_" + str(ch)] = torch.flip(gt[ch, :, :].unsqueeze(-1), [0])
        for ch in range(gt.shape[0]):
            stft["comparison_ch_" + str(ch)] = torch.cat([stft["stft_ch_" + str(ch)], stft["gt_ch_" + str(ch)]], dim=1)
        if self.use_grid:                                                               # :716-723
            # two full reductions over the 128^3 grid per call in the reference; the grid is static in eval, so the panels are
            # cached per write generation of the grid (mark_grid_written) and tensor version
            key = (self._grid_gen, self.grid._version, self.grid.data_ptr())
            pc = getattr(self, "_grid_panels", None)
            if pc is None or pc[0] != key:
                pc = (key, self.grid[0:3].mean(dim=3).permute(1, 2, 0), self.grid[3].mean(dim=2).unsqueeze(-1))
                self._grid_panels = pc
            stft["grid"], stft["grid_density"] = pc[1], pc[2]
        stft["raw_output"] = out                                                        # :725-726
        return stft

    @torch.no_grad()
    def get_outputs_for_rirs(self, mic_pose: torch.Tensor, source_pose: torch.Tensor, rot: torch.Tensor) -> torch.Tensor:
        """The eval branch (:648-694) for N RIRs in ONE field call: every RIR is the same T time queries at its own (microphone,
        source, orientation), so N RIRs are N*T independent rows of ``h`` (:560) -- the batch the MFMA GEMMs want (the reference
        evaluates one RIR = T rows per call, NeRAF_pipeline.py:355-362).  mic / source / rot [N,3] -> log-magnitude STFTs
        [N,T,C,F]; row n equals ``get_outputs_for_camera(None, None, batch_n)['raw_output']``."""
        dev = self.aabb.device
        T = self.max_len
        N = int(mic_pose.shape[0])
        tq = torch.arange(0, T, 1, device=dev).repeat(N)

        def rows(p):
            return p.to(dev).reshape(N, 1, 3).expand(N, T, 3).reshape(N * T, 3)
        feat = self.scene_feature() if self.use_grid else torch.zeros(0, device=dev)
        out = self.field.forward_queries(feat, tq, rows(mic_pose), rows(source_pose), rows(rot), self.aabb, T)
        return out.reshape(N, T, out.shape[1], out.shape[2])

    # ---- metrics --------------------------------------------------------------
    def get_metrics_dict(self, outputs: torch.Tensor, batch: Dict[str, torch.Tensor]):   # :568-581
        with torch.no_grad():
            mag_prd = torch.clip(torch.exp(outputs.detach().cpu()) - 1e-3, 0.0, 10000.0)
            mag_gt = torch.clip(torch.exp(batch["data"].detach().cpu()) - 1e-3, 0.0, 10000.0)
            return self.evaluator.get_stft_metrics(mag_prd, mag_gt)

    def get_audio_metrics(self, outputs: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], generator=None) -> Dict[str, float]:
        """The metric half of get_image_metrics_and_images (:738-761): magnitude STFTs -> Griffin-Lim waveforms (on the model's
        device) -> T60 / EDT / C50 (and RAF's spectral error) against the ground-truth waveform.  batch: 'data' [C,F,T] log-magnitude,
        'waveform' [C, n]; the image half is in ``get_image_metrics_and_images``."""
        with torch.no_grad(), _few_host_threads():
            stft = outputs["raw_output"].permute(1, 2, 0).detach().cpu()               # [C, F, T]
            data = batch["data"].detach().cpu()
            mag_prd = torch.clip(torch.exp(stft) - 1e-3, 0.0, 10000.0)
            mag_gt = torch.clip(torch.exp(data) - 1e-3, 0.0, 10000.0)
            dev = self.aabb.device
            wav_gt = batch["waveform"].detach().cpu().numpy()
            wav_istft_gt = self.istft_transform(mag_gt.to(dev), generator=generator).cpu().numpy()
            wav_istft_prd = self.istft_transform(mag_prd.to(dev), generator=generator).cpu().numpy()
            return self.evaluator.get_full_metrics(mag_prd.numpy(), mag_gt.numpy(), wav_gt, wav_istft_prd, wav_istft_gt,
                                                   stft.numpy(), data.numpy())

    def get_audio_metrics_block(self, raws: torch.Tensor, batches, generator=None):
        """``get_audio_metrics`` for N RIRs at once (the eval loop's blocks, NeRAFPipeline.get_average_eval_image_metrics): raws
        [N,T,C,F] log-magnitudes, batches = the N eval items.  The 2 N Griffin-Lim reconstructions (ground truth and prediction of
        every item: 32 iterations of an inverse + forward STFT each) run as ONE batched reconstruction instead of 2 N launches
        sequences of 61-frame transforms; the random initial phases are drawn item by item in the order the separate calls draw them
        (gt_0, prd_0, gt_1, ...), so a seeded generator gives the per-item results.  Returns the N metric dicts."""
        with torch.no_grad(), _few_host_threads():
            n = int(raws.shape[0])
            stft = raws.permute(0, 2, 3, 1).detach().cpu()                                # [N, C, F, T]
            data = torch.stack([b["data"].detach().cpu() for b in batches])            # [N, C, F, T]
            mag_prd = torch.clip(torch.exp(stft) - 1e-3, 0.0, 10000.0)
            mag_gt = torch.clip(torch.exp(data) - 1e-3, 0.0, 10000.0)
            dev = self.aabb.device
            gl = self.istft_transform
            phases = []
            for k in range(n):
                phases.append(gl.initial_phase(mag_gt[k].shape, dev, generator))
                phases.append(gl.initial_phase(mag_prd[k].shape, dev, generator))
            mags = torch.stack([mag_gt, mag_prd], dim=1).reshape(2 * n, *mag_gt.shape[1:]).to(dev)      # gt_0, prd_0, gt_1, ...
            wav = gl(mags, init_phase=torch.cat(phases, 0)).cpu().numpy()               # [2 N, C, L]
            out = []
            for k, b in enumerate(batches):
                out.append(self.evaluator.get_full_metrics(mag_prd[k].numpy(), mag_gt[k].numpy(), b["waveform"].detach().cpu().numpy(),
                                                           wav[2 * k + 1], wav[2 * k], stft[k].numpy(), data[k].numpy()))
            return out

    def get_image_metrics_and_images(self, outputs: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], generator=None):
        """NeRAF_model.py:738-803 as its callers use it (NeRAF_pipeline.py:278, :364): (metrics_dict, images_dict).  The metric
        half is ``get_audio_metrics``.  The image half (:763-803) is the reference's: every channel's predicted and ground-truth
        log-STFT panels normalised by the min / max over the GROUND-TRUTH panels, colour-mapped with matplotlib's viridis and set
        side by side as ``comparison_ch_{c}`` [F, 2T, 3]; with the grid in use, ``grid`` (the mean colour view) and the colour-mapped
        min-max-normalised ``grid_density``.  Host-side presentation code (numpy + matplotlib, imported here)."""
        from matplotlib import cm
        metrics = self.get_audio_metrics(outputs, batch, generator=generator)
        images: Dict[str, torch.Tensor] = {}
        ids = [k.replace("gt_ch_", "") for k in outputs if "gt" in k]                  # :765-773
        if ids:
            min_gt = min(float(outputs["gt_ch_" + i].min()) for i in ids)
            max_gt = max(float(outputs["gt_ch_" + i].max()) for i in ids)

            def panel(v: torch.Tensor) -> np.ndarray:
                v = (v.detach().cpu().double().numpy().squeeze() - min_gt) / (max_gt - min_gt)
                return cm.viridis(v)[..., :3]
            for i in ids:                                                               # :776-793
                images["comparison_ch_" + i] = torch.from_numpy(np.concatenate([panel(outputs["stft_ch_" + i]), panel(outputs["gt_ch_" + i])], axis=1))
        if self.use_grid and "grid" in outputs:                                         # :795-801
            images["grid"] = outputs["grid"]
            gd = outputs["grid_density"].detach().cpu().double().numpy().squeeze()
            images["grid_density"] = torch.from_numpy(cm.viridis((gd - gd.min()) / (gd.max() - gd.min()))[..., :3])
        return metrics, images

    def get_param_groups(self):                                                         # :730-737
        params = list(self.field.parameters())
        if self.use_grid:
            params += list(self.resnet3d.parameters())
        return {"audio_fields": params}
