"""Host-side mirror of ``NeRAF/NeRAF_resnet3d.py`` (``ResNet3D_helper`` :266-285, ``ResNet3D`` :116-201,
``Bottleneck`` :76-113) on libneraf_hip.

The module tree only holds parameters: it reproduces the reference's sub-module names so that state-dict
keys are identical (``backbone_net.conv1.weight``, ``backbone_net.layer2.0.downsample.1.running_var`` ...)
and reference checkpoints load with ``load_state_dict``.  ``forward`` never calls a torch conv: it hands
the 43 Conv3d weights and 43 BatchNorm3d parameter sets to ``neraf_resnet3d_fwd`` (implicit-GEMM MFMA
convolutions with fused BN statistics, include/neraf_hip.h).  In training mode the feature carries an autograd
node (``_ResNet3DFn``) whose backward runs ``neraf_resnet3d_bwd`` (dgrad / wgrad GEMMs, BatchNorm backward) and
returns gradients for every Conv3d weight and BatchNorm3d affine parameter; the gradient w.r.t. the cells of the
grid refreshed this step is handed to ``grid_grad_sink`` (the edge into the radiance field, NeRAF_model.py:395-400).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
import torch.nn as nn

from . import _lib
from .field import _dev_index, _stream_ptr


import os as _os

_GRID_WINDOW = _os.environ.get("NERAF_GRID_WINDOW", "1") != "0"      # 0: always convert the whole grid (measurement / debugging)


class _ResNet3DFn(torch.autograd.Function):
    """feat[1024] = ResNet3D(grid).  The 129 parameters (43 conv weights, 43 x (bn.weight, bn.bias)) do NOT travel through
    autograd: the backward kernels write all their gradients into one persistent flat buffer and this node assigns
    ``p.grad`` = (cached) views of it directly, as FSDP-style flat-parameter engines do.  Routing 129 tensors through autograd cost
    1.5-2 ms of host time per step (129 fresh view objects, 129 AccumulateGrad nodes, a 68 MB copy) for nothing.  Two flat
    buffers alternate, so the gradients of step N stay valid while step N+1 computes; accumulation without zero_grad adds in
    place.  ``anchor`` is a 1-element leaf that only keeps this node in the graph."""

    @staticmethod
    def forward(ctx, net: "ResNet3D", feat: torch.Tensor, window, sink, window_vals, anchor):
        # the forward kernels already ran (ResNet3D.forward); this node only attaches the backward.  ``window_vals``
        # ([n_ch, n_cells], may be None) are the grad-carrying values that were written into the grid window this step
        # (NeRAF_model.py:395-400): their gradient is the grid gradient at those cells.
        # ``feat`` is one of the module's two alternating output buffers (no copy): it stays valid until the training forward
        # after the next one, i.e. through this iteration's backward and beyond.
        ctx.net, ctx.window, ctx.sink = net, window, sink
        ctx.has_vals = window_vals is not None
        return feat.detach()

    @staticmethod
    def backward(ctx, dfeat: torch.Tensor):
        lib = _lib.load()
        net: ResNet3D = ctx.net
        dev = _dev_index(dfeat)
        h, st = _lib.ctx(dev), _stream_ptr()
        device = dfeat.device
        tb = net._host_tables()
        params = tb["params"]
        # every buffer the kernels see is persistent (module-owned): the ~270-launch sequence is replayed as ONE hipGraph keyed by
        # its argument pointers (include/neraf_hip.h, neraf_graph_stats), so the pointers must not change from step to step
        if net._packed_t is None or net._packed_t.device != device:
            net._packed_t = torch.empty(lib.neraf_resnet3d_bwd_packed_bytes(C.byref(net._desc)), dtype=torch.uint8, device=device)
        packed_t = net._packed_t
        _lib.check(lib.neraf_resnet3d_pack_weights_bwd(h, C.byref(net._desc), tb["conv_w_ptrs"], packed_t.data_ptr(), st), dev)
        if net._bws is None or net._bws.device != device:
            net._bws = torch.empty(lib.neraf_resnet3d_bwd_workspace_bytes(C.byref(net._desc)), dtype=torch.uint8, device=device)
            # the fp16 gradient chain's per-tensor scales live in this buffer: a new buffer starts with calibration passes
            _lib.check(lib.neraf_resnet3d_bwd_reset(h, net._bws.data_ptr()), dev)
        if net._grad_bufs is None or net._grad_bufs[0]["flat"].device != device:
            sizes = [p.numel() for p in params]
            nconv = len(tb["pairs"])
            net._grad_bufs = []
            for _ in range(2):
                flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
                views = [v.view(p.shape) for v, p in zip(torch.split(flat, sizes), params)]
                net._grad_bufs.append(dict(flat=flat, views=views, lo=flat.data_ptr(), hi=flat.data_ptr() + flat.numel() * 4,
                                           ptrs=(_lib.ptr_array(views[:nconv]), _lib.ptr_array(views[nconv:]))))
            net._grad_turn = 0
        # pick the buffer that no live gradient aliases (normally they simply alternate)
        g0 = params[0].grad
        k = net._grad_turn
        if g0 is not None and net._grad_bufs[k]["lo"] <= g0.data_ptr() < net._grad_bufs[k]["hi"]:
            k ^= 1
        buf = net._grad_bufs[k]
        net._grad_turn = k ^ 1
        start, n_cells, n_ch = ctx.window if ctx.window is not None else (0, 0, 0)
        if n_cells > 0 and (net._dgrid_buf is None or tuple(net._dgrid_buf.shape) != (n_ch, n_cells) or net._dgrid_buf.device != device):
            net._dgrid_buf = torch.empty((n_ch, n_cells), dtype=torch.float32, device=device)
        net.dfeat_buffer(device)
        if dfeat.data_ptr() != net._dfeat_buf.data_ptr():      # the consumer wrote its gradient straight into the buffer: see dfeat_buffer
            net._dfeat_buf.copy_(dfeat.reshape(-1))
        if net.dp_group is not None:
            # data parallel: every rank holds the same grid and weights and the backward is linear in d feat, so
            # averaging these 1024 floats (4 KiB) replaces the all-reduce of 17 M ResNet3D gradients (68 MB).  Exact only if the
            # forward is bit-identical on every rank: NeRAFPipeline.attach_gradient_reducer switches this on under
            # NERAF_DETERMINISTIC=1 (fixed-order sums) and leaves it off in the default mode, where the BatchNorm statistics are
            # accumulated with fp32 atomics (last-bit order dependence, amplified by the chaotic encoder) and the gradients are
            # all-reduced instead (GradientReducer); NERAF_DP_DFEAT=0 / 1 forces either
            import torch.distributed as dist
            grp = None if net.dp_group is True else net.dp_group
            if dist.get_backend(grp) == "nccl":
                dist.all_reduce(net._dfeat_buf, op=dist.ReduceOp.AVG, group=grp)
            else:
                dist.all_reduce(net._dfeat_buf, group=grp)
                net._dfeat_buf.div_(dist.get_world_size(grp))
        _lib.check(lib.neraf_resnet3d_bwd(h, C.byref(net._desc), packed_t.data_ptr(), tb["conv_w_ptrs"], tb["bn_ptrs"],
                                          net._ws.data_ptr(), net._bws.data_ptr(), net._dfeat_buf.data_ptr(), buf["ptrs"][0],
                                          buf["ptrs"][1], start, n_cells, n_ch,
                                          net._dgrid_buf.data_ptr() if n_cells > 0 else None, st), dev)
        if g0 is None:
            for p, v in zip(params, buf["views"]):
                if p.requires_grad:
                    p.grad = v
        else:                                      # gradient accumulation (no zero_grad since the last backward): add in place
            live = [(p, v) for p, v in zip(params, buf["views"]) if p.requires_grad]
            for p, v in live:
                if p.grad is None:
                    p.grad = v.clone()
            torch._foreach_add_([p.grad for p, _ in live], [v for _, v in live])
        if net.grads_ready_hook is not None:       # e.g. the data-parallel reducer: this group's gradients are final
            net.grads_ready_hook()
        # the module's persistent buffer itself (no copy): valid until this module's next backward; its consumer (the grid refresh's
        # backward node) reads it inside this pass
        dgrid = net._dgrid_buf.detach() if n_cells > 0 else None
        if ctx.sink is not None and dgrid is not None:
            ctx.sink(dgrid)
        return (None, None, None, None, dgrid if ctx.has_vals else None, None)


class _Bottleneck(nn.Module):
    """Parameter container with the attribute names of the reference Bottleneck (NeRAF_resnet3d.py:79-90)."""
    expansion = 4

    def __init__(self, in_planes: int, planes: int, stride: int, with_downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv3d(in_planes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(nn.Conv3d(in_planes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                            nn.BatchNorm3d(planes * 4))

    def conv_bn_pairs(self):
        pairs = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        if self.downsample is not None:
            pairs.append((self.downsample[0], self.downsample[1]))
        return pairs


class ResNet3D(nn.Module):
    """'resnet50' layout truncated after layer3 (N_features = 1024) or with layer4 (N_features = 2048), on a 64^3, 128^3 or 256^3
    grid (grid_step 1/64 | 1/128 | 1/256) -- every combination the reference's constructor accepts (NeRAF_resnet3d.py:116-165: the
    average pool spans the last layer's whole output in each of them)."""

    def __init__(self, in_channels: int = 7, layers=(3, 4, 6), grid_step: float = 1 / 128, N_features: int = 1024):
        super().__init__()
        if N_features not in (1024, 2048):
            raise ValueError("N_features should be 1024 or 2048")                      # NeRAF_resnet3d.py:128
        layers = tuple(layers)
        if N_features == 2048 and layers == (3, 4, 6):
            layers = (3, 4, 6, 3)                                                      # resnet50's layer4, :131 / :237
        if in_channels != 7 or layers != ((3, 4, 6) if N_features == 1024 else (3, 4, 6, 3)):
            raise NotImplementedError("the HIP scene encoder implements the configurations NeRAF uses: "
                                      "in_channels=7, resnet50 [3,4,6(,3)], N_features 1024 | 2048 (NeRAF_model.py:185)")
        if grid_step >= 1 / 64 - 1 / 512:
            self.grid_size = 64
        elif grid_step >= 1 / 128 - 1 / 512:
            self.grid_size = 128
        else:
            self.grid_size = 256                                                       # :150-156
        self.n_layers = len(layers)
        self.conv1 = nn.Conv3d(in_channels, 64, kernel_size=5, stride=2, padding=2, bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        in_planes = 64
        for li, (planes, nblocks, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2)), start=1):
            blocks = []
            for b in range(nblocks):
                s = stride if b == 0 else 1
                blocks.append(_Bottleneck(in_planes, planes, s, b == 0 and (s != 1 or in_planes != planes * 4)))
                in_planes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.N_features = N_features
        for m in self.modules():                                   # NeRAF_resnet3d.py:160-165
            if isinstance(m, nn.Conv3d):
                nn.init.xavier_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self._desc = _lib.ResnetDesc(self.grid_size, in_channels, N_features)
        self._ws = None
        self._bws = None
        self._packed, self._packed_key = None, None
        self._tables = None
        self._packed_t = self._grad_bufs = self._dfeat_buf = self._dgrid_buf = self._feat_buf = None
        self._grad_turn = 0
        self._anchor = None          # 1-element leaf keeping _ResNet3DFn in the autograd graph (the parameters bypass autograd)
        self.grads_ready_hook = None # callable() invoked when the backward has assigned every parameter gradient
        self.dp_group = None         # torch.distributed group (or True = default group): average d feat over ranks in the backward
        self.grid_window = None      # (cell_start, n_cells, n_channels): grid cells whose gradient the backward should produce
        self.grid_grad_sink = None   # callable(dgrid_cells fp32 [n_ch, n_cells]) invoked inside the backward

    def dfeat_buffer(self, device) -> torch.Tensor:
        """The persistent fp32 [1024] buffer the backward sequence reads d loss / d feature from (its pointer is part of the captured
        hipGraph).  A consumer of the feature that is handed this buffer (``feature._neraf_grad_buffer``, set by ``forward``) may write
        its gradient w.r.t. the feature directly into it and return it as that gradient: the backward then skips its 4 KB copy."""
        if self._dfeat_buf is None or self._dfeat_buf.device != torch.device(device):
            self._dfeat_buf = torch.empty(self.N_features, dtype=torch.float32, device=device)
        return self._dfeat_buf

    def conv_bn_pairs(self):
        pairs = [(self.conv1, self.bn1)]
        for li in range(1, self.n_layers + 1):
            for blk in getattr(self, f"layer{li}"):
                pairs += blk.conv_bn_pairs()
        return pairs

    def _host_tables(self):
        """(pairs, conv weights, BN tensors, ctypes pointer arrays) of the 43 conv/BN pairs.  Building 215 detached views and three
        pointer arrays costs ~0.5 ms of host time per call; they are cached and re-validated by comparing the 215 data pointers
        (tens of microseconds), which catches .to(device), load_state_dict into new storage and parameter replacement alike."""
        c = self._tables
        if c is not None:
            if [t.data_ptr() for t in c["live"]] == c["ptrs"]:
                return c
        pairs = self.conv_bn_pairs()
        conv_w = [cv.weight.detach().contiguous() for cv, _ in pairs]
        bn: List[torch.Tensor] = []
        live: List[torch.Tensor] = [cv.weight for cv, _ in pairs]
        for _, b in pairs:
            bn += [b.weight.detach(), b.bias.detach(), b.running_mean, b.running_var]
            live += [b.weight, b.bias, b.running_mean, b.running_var, b.num_batches_tracked]
        self._tables = c = dict(pairs=pairs, conv_w=conv_w, bn=bn, live=live, ptrs=[t.data_ptr() for t in live],
                                conv_w_ptrs=_lib.ptr_array(conv_w), bn_ptrs=_lib.ptr_array(bn),
                                nbt=[b.num_batches_tracked for _, b in pairs],
                                nbt_ptrs=_lib.ptr_array([b.num_batches_tracked for _, b in pairs]),
                                params=[cv.weight for cv, _ in pairs] + [t for _, b in pairs for t in (b.weight, b.bias)])
        return c

    def forward(self, x: torch.Tensor, window=None, window_vals: torch.Tensor = None, grid_state=None) -> torch.Tensor:
        """x fp32 [1,7,S,S,S] -> [1,N_features,1,1,1] (NeRAF_resnet3d.py:184-198).  ``window`` = (cell_start, n_cells, n_ch) and
        ``window_vals`` [n_ch, n_cells] (requires_grad) describe the grid cells refreshed this step, whose gradient is
        returned to ``window_vals`` by the backward.

        ``grid_state`` = (generation_now, generation_before, cell_start, n_cells, version_before, version_after): the owner of the grid
        vouches that between its generations `before` and `now` only the cells [cell_start, cell_start + n_cells) were written, by a
        write that took the tensor's version counter from `version_before` to `version_after`.  When the image this module converted
        last (same storage, same workspace, same mode) was of generation `before` at version `version_before`, and the tensor still
        is at `version_after`, only that window is re-converted (the fp32 -> fp16 channels-last conversion of all 2 M cells is 25 us
        of a step in which 4096 changed); in every other case -- no state given, another tensor, a generation or an in-place
        operation in between missed -- the whole grid is converted."""
        lib = _lib.load()
        S = self.grid_size
        if tuple(x.shape) != (1, 7, S, S, S):
            raise ValueError(f"expected a [1,7,{S},{S},{S}] grid, got {tuple(x.shape)}")
        dev = _dev_index(x)
        h, st = _lib.ctx(dev), _stream_ptr()
        grid = x.detach().float().contiguous()
        tb = self._host_tables()
        pairs, conv_w = tb["pairs"], tb["conv_w"]
        assert len(pairs) == lib.neraf_resnet3d_num_convs(C.byref(self._desc))
        # fp16 implicit-GEMM weight blob: re-packed every training forward (optimizers update the weights in place without
        # bumping tensor versions), in eval only when a weight's (storage, version) changed
        repack = self.training or self._packed is None or self._packed.device != x.device
        if not repack:
            from . import optim
            key = (optim.UPDATE_EPOCH,) + tuple((w.data_ptr(), w._version) for w in conv_w)
            repack = key != self._packed_key
        if repack:
            if self._packed is None or self._packed.device != x.device:      # persistent: its pointer is part of the graph key
                self._packed = torch.empty(lib.neraf_resnet3d_packed_bytes(C.byref(self._desc)), dtype=torch.uint8, device=x.device)
            _lib.check(lib.neraf_resnet3d_pack_weights(h, C.byref(self._desc), tb["conv_w_ptrs"], self._packed.data_ptr(), st), dev)
            if self.training:
                self._packed_key = None
            else:
                from . import optim
                self._packed_key = (optim.UPDATE_EPOCH,) + tuple((w.data_ptr(), w._version) for w in conv_w)
        packed = self._packed
        if self._ws is None or self._ws.device != x.device:
            self._ws = torch.empty(lib.neraf_resnet3d_workspace_bytes(C.byref(self._desc)), dtype=torch.uint8, device=x.device)
        # Eval-mode forwards run in a workspace of their own (allocated at the first one): the training workspace holds the activations,
        # pool arguments and BatchNorm statistics a pending training backward reads, and an evaluation between a training forward and
        # its backward (a viewer frame, an eval hook) must not overwrite them (the third feature buffer alone protected
        # 4 KB of output, not the saved activations).  ~0.4 GB at 128^3, of 288.
        if self.training:
            ws = self._ws
        else:
            if getattr(self, "_ws_eval", None) is None or self._ws_eval.device != x.device:
                self._ws_eval = torch.empty(self._ws.numel(), dtype=torch.uint8, device=x.device)
            ws = self._ws_eval
        if self._feat_buf is None or self._feat_buf[0].device != x.device:
            self._feat_buf = [torch.empty(self.N_features, dtype=torch.float32, device=x.device) for _ in range(3)]
            self._feat_turn = 0
        # two output buffers alternate between TRAINING forwards (two captured forward graphs): the feature handed out is not copied,
        # and stays intact while the next forward writes the other buffer -- contract: at most one training feature is awaiting its
        # backward when the forward after next runs (the training loop's shape).  Eval-mode forwards write a third buffer of their own
        # (and their own workspace, above)
        if self.training:
            self._feat_turn ^= 1
            feat_buf = self._feat_buf[self._feat_turn]
        else:
            feat_buf = self._feat_buf[2]
        # the fp16 channels-last image of the grid lives in the workspace: each workspace remembers which grid generation it holds
        key = (grid.data_ptr(), ws.data_ptr(), bool(self.training))
        if not isinstance(getattr(self, "_x0_states", None), dict):
            self._x0_states = {}
        win = (0, 0)
        if (grid_state is not None and _GRID_WINDOW and self._x0_states.get(bool(self.training)) == (key, grid_state[1], grid_state[4])
                and grid_state[0] == grid_state[1] + 1 and x._version == grid_state[5] and 0 < grid_state[3] <= S ** 3 - grid_state[2]):
            win = (int(grid_state[2]), int(grid_state[3]))
        _lib.check(lib.neraf_resnet3d_fwd(h, C.byref(self._desc), packed.data_ptr(), tb["bn_ptrs"], grid.data_ptr(),
                                          ws.data_ptr(), feat_buf.data_ptr(), int(self.training), win[0], win[1], st), dev)
        self._x0_states[bool(self.training)] = (key, grid_state[0], x._version) if grid_state is not None else None
        feat = feat_buf
        if self.training:
            mom = pairs[0][1].momentum if pairs[0][1].momentum is not None else 0.1
            if mom > 0:
                # running statistics and the 43 num_batches_tracked counters in one launch
                _lib.check(lib.neraf_resnet3d_update_running_stats(h, C.byref(self._desc), self._ws.data_ptr(),
                                                                   tb["bn_ptrs"], float(mom), tb["nbt_ptrs"], st), dev)
        if self.training and torch.is_grad_enabled() and any(c.weight.requires_grad for c, _ in pairs):
            if window is None:
                window = self.grid_window
            if window_vals is not None and not window_vals.requires_grad:
                window_vals = None
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros(1, dtype=torch.float32, device=x.device, requires_grad=True)
            feat = _ResNet3DFn.apply(self, feat, window, self.grid_grad_sink, window_vals, self._anchor)    # the buffer itself, see there
            out = feat.reshape(1, self.N_features, 1, 1, 1)
            out._neraf_grad_buffer = self.dfeat_buffer(x.device)
            return out
        return feat.clone().reshape(1, self.N_features, 1, 1, 1)      # inference: callers cache this (never the persistent buffer)


class ResNet3D_helper(nn.Module):
    """Same constructor and attribute (``backbone_net``) as the reference helper (NeRAF_resnet3d.py:266-285)."""

    def __init__(self, in_channels: int = 3, backbone: str = "resnet50", pretrained: bool = False, grid_step=None,
                 N_features: int = 1024):
        super().__init__()
        if backbone != "resnet50" or pretrained:
            raise NotImplementedError("only backbone='resnet50', pretrained=False (what NeRAF instantiates, NeRAF_model.py:185)")
        self.backbone_net = ResNet3D(in_channels, (3, 4, 6), grid_step if grid_step is not None else 1 / 128, N_features if N_features is not None else 1024)

    def forward(self, x, window=None, window_vals=None, grid_state=None):
        return self.backbone_net(x, window, window_vals, grid_state)
