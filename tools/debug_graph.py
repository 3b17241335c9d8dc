import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neraf_amd import synth, _lib
from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
from neraf_amd.vision import NeRAFVisionModel
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")
vm = NeRAFVisionModel(torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]), 210).to(dev)
am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 64), T(synth.audio_aabb()))
am.field.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(1187, 512, 1, 513).items()})
am.resnet3d.backbone_net.load_state_dict({k: T(v) for k, v in synth.resnet3d_state_dict(7).items()})
am.to(dev)
net = am.resnet3d.backbone_net
def stats(tag):
    w = net.layer2[1].conv2.weight.grad
    b = net.bn1.weight.grad
    print(tag, "conv grad rms", float(w.double().pow(2).mean().sqrt()) if w is not None else None, "finite", bool(torch.isfinite(w).all()) if w is not None else None,
          "bn1", float(b.abs().max()) if b is not None else None)
def clear():
    for p in list(vm.parameters()) + list(am.parameters()): p.grad = None
b = {k: T(v).to(dev) for k, v in synth.audio_batch(128, 1, 513, 60, tag="t.joint").items()}
seq = sys.argv[1] if len(sys.argv) > 1 else "AEW"
for ch in seq:
    if ch == "A":      # train, no window
        am.train(); clear()
        y = am.get_outputs(b); ld = am.get_loss_dict(y, b); (ld["audio_sc_loss"] + ld["audio_mag_loss"]).backward(); stats("A no-window")
    if ch == "E":
        am.eval()
        with torch.no_grad():
            f = am.scene_feature(); print("E eval feat", float(f.abs().mean()))
    if ch == "W":
        am.train(); clear(); vm.train()
        am.query_grid_one_batch(0, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=2048)
        y = am.get_outputs(b); ld = am.get_loss_dict(y, b); (ld["audio_sc_loss"] + ld["audio_mag_loss"]).backward(); stats("W window")
torch.cuda.synchronize()
lib = _lib.load(); import ctypes as C
c, l = C.c_int(), C.c_int()
print("graphs enabled", lib.neraf_graph_stats(_lib.ctx(0), C.byref(c), C.byref(l)), "captures", c.value, "launches", l.value)

if "X" in seq:      # exact replica of tests/test_gpu_model.py eval test followed by the joint test
    am.train()
    for m in am.resnet3d.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.momentum = 0.0
    B = 256
    bb = {k: T(v) for k, v in synth.audio_batch(B, 1, 513, 60, tag="t.model").items()}
    y = am.get_outputs({k: v.to(dev) for k, v in bb.items()})
    ld = am.get_loss_dict(y, {k: v.to(dev) for k, v in bb.items()})
    (ld["audio_sc_loss"] + ld["audio_mag_loss"]).backward(); stats("X1 bwd")
    am.eval()
    item = {"mic_pose": bb["mic_pose"][5], "source_pose": bb["source_pose"][5], "rot": bb["rot"][5],
            "data": T(synth.uniform("t.model.gt", (1, 513, 60), -6.0, 1.0))}
    out = am.get_outputs_for_camera(None, None, batch_audio=item)
    am.train()
    vm.train(); clear()
    am.query_grid_one_batch(0, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=2048)
    y = am.get_outputs(b); ld = am.get_loss_dict(y, b); (ld["audio_sc_loss"] + ld["audio_mag_loss"]).backward(); stats("X2 joint")
    torch.cuda.synchronize()
    c, l = C.c_int(), C.c_int()
    print("graphs enabled", lib.neraf_graph_stats(_lib.ctx(0), C.byref(c), C.byref(l)), "captures", c.value, "launches", l.value)
