"""Checkpoint interop (the part that can be validated offline).

`NeRAFPipeline.state_dict` (NeRAF_pipeline.py:492-497) is the nerfstudio pipeline state -- `_model.*` (vision), `audio_model.*`,
`datamanager.*` -- plus the voxel grid under "audio_model.grid"; `load_pipeline` (:438-464) strips DDP's "module." prefix, pops the
grid, loads the rest and re-attaches the grid.  The audio half of a reference checkpoint loads as it is: `NeRAFAudioSoundField`,
`ResNet3D_helper` and the grid use the reference's parameter names and shapes (tests/test_gpu_model.py checks the key sets).

The radiance half of a REFERENCE checkpoint is tiny-cuda-nn's flat parameter blobs (`_model.field.mlp_base.params`, ...); their
internal layout cannot be checked here (tcnn is absent), so they are reported in `skipped_tcnn` and left to a converter that has
the real library next to it.  Checkpoints written by THIS package round-trip completely (native `_model.*` keys)."""
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
                  step: Optional[int] = None) -> Dict[str, List[str]]:
    """Mirror of NeRAFPipeline.load_pipeline.  Returns {'loaded', 'skipped_tcnn', 'ignored', 'missing'} key lists."""
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
    report["ignored"] += [k for k in state if not k.startswith(("audio_model.", "_model."))]       # datamanager.*, camera optimizer
    if grid is not None and getattr(audio_model, "use_grid", False):
        with torch.no_grad():
            audio_model.grid.copy_(grid.to(audio_model.grid.device, audio_model.grid.dtype))        # :456
        audio_model._feat_key = None
    if vision_model is not None and hasattr(vision_model, "field"):
        audio_model.spatial_distortion = vision_model.field.module.spatial_distortion               # :459
        # :465 -- kept a plain attribute: registering the audio model as a sub-module of the vision model would duplicate its
        # parameters in the vision model's state dict and parameter groups
        object.__setattr__(vision_model, "audio_model", audio_model)
    return report
