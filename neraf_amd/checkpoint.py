"""Checkpoint interop (the part that can be validated offline).

`NeRAFPipeline.state_dict` (NeRAF_pipeline.py:492-497) is the nerfstudio pipeline state -- `_model.*` (vision), `audio_model.*`,
`datamanager.*` -- plus the voxel grid under "audio_model.grid"; `load_pipeline` (:438-464) strips DDP's "module." prefix, pops the
grid, loads the rest and re-attaches the grid.  The audio half of a reference checkpoint loads as it is: `NeRAFAudioSoundField`,
`ResNet3D_helper` and the grid use the reference's parameter names and shapes (tests/test_gpu_model.py checks the key sets).

The radiance half of a REFERENCE checkpoint is tiny-cuda-nn's flat parameter blobs (`_model.field.module.mlp_base.params`, ...).
`tcnn_blobs_to_native` converts them under the layout tiny-cuda-nn documents [TCNN-recall, UNVERIFIED here: tcnn is not installed and
no released checkpoint can be fetched offline]: a blob is the network's weight matrices in layer order, each `[out, in]` row-major
with in / out padded to multiples of 16, followed (NetworkWithInputEncoding) by the encoding's levels in order, each `[entries,
features]` row-major.  The element counts of every blob must match that hypothesis exactly, otherwise the blob is left in
`skipped_tcnn`; a matching count is necessary, not sufficient -- the first run next to a real tcnn must compare one forward.
Checkpoints written by THIS package round-trip completely (native `_model.*` keys)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch


def pipeline_state_dict(vision_model: torch.nn.Module, audio_model: torch.nn.Module) -> Dict[str, torch.Tensor]:
    """Pipeline-level state in the reference's key space: `_model.*`, `audio_model.*` and "audio_model.grid"."""
    out = {"_model." + k: v for k, v in vision_model.state_dict().items()}
    out.update({"audio_model." + k: v for k, v in audio_model.state_dict().items()})
    if getattr(audio_model, "use_grid", False):
        out["audio_model.grid"] = audio_model.grid        # already a persistent buffer here; the reference adds it by hand (:496)
    return out


def load_pipeline(loaded_state: Dict[str, Any], vision_model: Optional[torch.nn.Module], audio_model: torch.nn.Module,
                  step: Optional[int] = None, convert_tcnn: bool = False) -> Dict[str, List[str]]:
    """Mirror of NeRAFPipeline.load_pipeline.  Returns {'loaded', 'skipped_tcnn', 'ignored', 'missing'} key lists.
    ``convert_tcnn`` (opt-in): also load tiny-cuda-nn flat parameter blobs through ``tcnn_blobs_to_native``; its layout hypothesis is
    UNVERIFIED in this image (module docstring), so blobs are left in ``skipped_tcnn`` by default and a conversion warns."""
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in loaded_state.items()}
    report = {"loaded": [], "skipped_tcnn": [], "ignored": [], "missing": []}
    if step is not None and vision_model is not None and hasattr(vision_model, "update_to_step"):
        vision_model.update_to_step(step)
    grid = state.pop("audio_model.grid", None)

    def load_into(module: torch.nn.Module, prefix: str):
        own = module.state_dict()
        sub = {}
        for k, v in state.items():
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if name.endswith(".params") and name not in own:
                report["skipped_tcnn"].append(k)             # tcnn flat parameter blob
            elif name in own and (tuple(own[name].shape) == tuple(v.shape) or (own[name].dim() == 0 and v.numel() == 1)):
                sub[name] = v.reshape(own[name].shape)
                report["loaded"].append(k)
            else:
                report["ignored"].append(k)                  # e.g. torchaudio's GriffinLim window, loss-module buffers
        report["missing"] += [prefix + n for n in own if n not in sub and n != "grid"]
        module.load_state_dict(sub, strict=False)

    load_into(audio_model, "audio_model.")
    if vision_model is not None:
        load_into(vision_model, "_model.")
        if convert_tcnn and report["skipped_tcnn"] and hasattr(vision_model, "proposal_networks"):
            conv = tcnn_blobs_to_native(state, vision_model, "_model.")          # blobs whose element counts match the documented layout
            report["converted_tcnn"] = conv["converted"]
            report["skipped_tcnn"] = [k for k in report["skipped_tcnn"] if k not in conv["converted"]]
            if conv["converted"]:
                import warnings
                warnings.warn("loaded %d tiny-cuda-nn parameter blobs under an UNVERIFIED layout hypothesis (element counts matched); "
                              "compare one forward against the reference before trusting the radiance field" % len(conv["converted"]))
    report["ignored"] += [k for k in state if not k.startswith(("audio_model.", "_model."))]       # datamanager.*, camera optimizer
    if grid is not None and getattr(audio_model, "use_grid", False):
        with torch.no_grad():
            audio_model.grid.copy_(grid.to(audio_model.grid.device, audio_model.grid.dtype))        # :456
        audio_model._feat_key = None
        if hasattr(audio_model, "mark_grid_written"):
            audio_model.mark_grid_written()                                                          # a write of unknown extent
    if vision_model is not None and hasattr(vision_model, "field"):
        audio_model.spatial_distortion = vision_model.field.module.spatial_distortion               # :459
        # :465 -- kept a plain attribute: registering the audio model as a sub-module of the vision model would duplicate its
        # parameters in the vision model's state dict and parameter groups
        object.__setattr__(vision_model, "audio_model", audio_model)
    return report


# ---- tiny-cuda-nn flat parameter blobs [TCNN-recall, unverified layout] -----------------------------------------------------------
def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def split_tcnn_mlp(blob: torch.Tensor, n_in: int, width: int, n_hidden: int, n_out: int):
    """FullyFusedMLP blob -> [W0 [width, pad16(n_in)], (n_hidden - 1) x [width, width], W_out [pad16(n_out), width]]; None if the size
    does not match."""
    shapes = [(width, _pad16(n_in))] + [(width, width)] * (n_hidden - 1) + [(_pad16(n_out), width)]
    need = sum(a * b for a, b in shapes)
    if blob.numel() != need:
        return None
    out, off = [], 0
    for a, b in shapes:
        out.append(blob[off:off + a * b].reshape(a, b).float())
        off += a * b
    return out


def join_tcnn_mlp(mats) -> torch.Tensor:
    return torch.cat([m.reshape(-1) for m in mats])


def tcnn_blobs_to_native(state: Dict[str, torch.Tensor], vision_model: torch.nn.Module, prefix: str = "_model.") -> Dict[str, List[str]]:
    """Load the radiance half of a reference (nerfstudio + tcnn) checkpoint into ``vision_model`` from tcnn blobs, accepting both key
    styles nerfstudio has used: ``field[.module].mlp_base.params`` (NetworkWithInputEncoding: MLP then grid) + ``mlp_head.params``,
    and ``mlp_base_grid.tcnn_encoding.params`` / ``mlp_base_mlp.tcnn_encoding.params`` / ``mlp_head.tcnn_encoding.params``;
    proposal networks likewise under ``proposal_networks.{i}``; the appearance embedding under ``embedding_appearance.embedding.weight``.
    Returns {'converted': [...], 'skipped_tcnn': [...]} (size mismatches are skipped, never guessed)."""
    rep = {"converted": [], "skipped_tcnn": []}
    f = vision_model.field.module

    def find(*suffixes):
        for k in state:
            if k.startswith(prefix) and any(k[len(prefix):] == s_ or k[len(prefix):] == s_.replace("field.", "field.module.") for s_ in suffixes):
                return k
        return None

    def load_net_with_grid(key, table, mlp_targets, n_in, width, n_hidden, n_out):
        blob = state[key].reshape(-1)
        n_grid = table.numel()
        if blob.numel() <= n_grid:
            rep["skipped_tcnn"].append(key)
            return
        mats = split_tcnn_mlp(blob[:blob.numel() - n_grid], n_in, width, n_hidden, n_out)
        if mats is None:
            rep["skipped_tcnn"].append(key)
            return
        with torch.no_grad():
            table.copy_(blob[blob.numel() - n_grid:].reshape(table.shape).float())
            for t, m in zip(mlp_targets, mats):
                t.copy_(m[:t.shape[0], :t.shape[1]])
        rep["converted"].append(key)

    def load_mlp(key, targets, n_in, width, n_hidden, n_out):
        mats = split_tcnn_mlp(state[key].reshape(-1), n_in, width, n_hidden, n_out)
        if mats is None:
            rep["skipped_tcnn"].append(key)
            return
        with torch.no_grad():
            for t, m in zip(targets, mats):
                t.copy_(m[:t.shape[0], :t.shape[1]])
        rep["converted"].append(key)

    def load_grid(key, table):
        blob = state[key].reshape(-1)
        if blob.numel() != table.numel():
            rep["skipped_tcnn"].append(key)
            return
        with torch.no_grad():
            table.copy_(blob.reshape(table.shape).float())
        rep["converted"].append(key)

    k = find("field.mlp_base.params")
    if k:
        load_net_with_grid(k, f.table, [f.base_w0, f.base_w1], 32, 64, 1, 16)
    else:
        kg, km = find("field.mlp_base_grid.tcnn_encoding.params"), find("field.mlp_base_mlp.tcnn_encoding.params")
        if kg:
            load_grid(kg, f.table)
        if km:
            load_mlp(km, [f.base_w0, f.base_w1], 32, 64, 1, 16)
    k = find("field.mlp_head.params", "field.mlp_head.tcnn_encoding.params")
    if k:
        load_mlp(k, [f.head_w0, f.head_w1, f.head_w2], 63, 64, 2, 3)
    k = find("field.embedding_appearance.embedding.weight")
    if k and tuple(state[k].shape) == tuple(f.embedding.shape):
        with torch.no_grad():
            f.embedding.copy_(state[k].float())
        rep["converted"].append(k)
    for i, pn in enumerate(vision_model.proposal_networks):
        k = find(f"proposal_networks.{i}.mlp_base.params")
        if k:
            load_net_with_grid(k, pn.table, [pn.w0, pn.w1], 10, 16, 1, 1)
        else:
            kg, km = find(f"proposal_networks.{i}.mlp_base_grid.tcnn_encoding.params"), find(f"proposal_networks.{i}.mlp_base_mlp.tcnn_encoding.params")
            if kg:
                load_grid(kg, pn.table)
            if km:
                load_mlp(km, [pn.w0, pn.w1], 10, 16, 1, 1)
    return rep
