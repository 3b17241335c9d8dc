"""Stress of tests/test_gpu_pipeline.py's flow over initialisation seeds: gradient magnitudes of the proposal networks (exactly
zero when the interlevel loss has no histogram violation), non-finite gradients, and whether the proposal weights move."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neraf_amd import synth
from neraf_amd.data import DeviceRIRBank
from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
from neraf_amd.pipeline import FixedBatchDataManager, NeRAFPipeline, RIRBankDataManager, _ScaledLossSum
from neraf_amd.vision import NeRAFVisionModel, RayBundle
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")
am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 64), T(synth.audio_aabb()))
am.field.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(1187, 512, 1, 513).items()})
am.resnet3d.backbone_net.load_state_dict({k: T(v) for k, v in synth.resnet3d_state_dict(7).items()})
am.to(dev).train()
rb = synth.ray_batch(512, tag="pipe.rays")
bundle = RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev))
n = 15360
tt = np.arange(n) / 48000.0
waves = torch.from_numpy(np.stack([synth.normal(f"pipe.rir{i}", (n,), 1.0, np.float64) * np.exp(-tt / 0.05) for i in range(6)])).float()
aabb = synth.audio_aabb()
pos = lambda tag: T(synth.uniform(tag, (6, 3), 0.2, 0.8)).double() * T(aabb[1] - aabb[0]).double() + T(aabb[0]).double()
bank = DeviceRIRBank.from_waveforms(waves, 48000, 60, pos("pipe.mic"), pos("pipe.src"), T(synth.uniform("pipe.rot", (6, 3), -3, 3)).double(), device=dev)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    torch.manual_seed(seed)
    vm = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 210).to(dev).train()
    pipe = NeRAFPipeline(vm, am, datamanager=FixedBatchDataManager(bundle, {"image": T(rb["rgb"]).to(dev)}, 512),
                         audio_datamanager=RIRBankDataManager(bank, 256, generator=torch.Generator(device=dev).manual_seed(0)), start_step_audio=5)
    pipe.get_train_loss_dict(3); pipe.get_train_loss_dict(6)
    opts, scaler = pipe.make_optimizers(init_scale=1024.0)
    w0 = vm.proposal_networks[0].w0
    before = w0.detach().clone()
    msgs = []
    for step in (7, 8):
        pipe.model.update_to_step(step)
        for o in opts: o.zero_grad(set_to_none=True)
        _, ld, _ = pipe.get_train_loss_dict(step)
        scaled, loss = _ScaledLossSum.apply(scaler, *ld.values())
        scaled.backward()
        bad = [n_ for n_, p in list(vm.named_parameters()) + list(am.named_parameters()) if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        nog = [n_ for n_, p in vm.named_parameters() if p.grad is None]
        msgs.append(f"step {step}: loss {float(loss):.4g} nonfinite {bad} nograd {nog} scale {scaler.get_scale()}")
        for o in opts: scaler.step(o)
        scaler.update()
    pn = vm.proposal_networks[0]
    st0 = opts[0].state
    info = {n_: (float(p.grad.abs().max()), float(st0[p]["exp_avg"].abs().max()) if p in st0 and "exp_avg" in st0[p] else None)
            for n_, p in (("p0.table", pn.table), ("p0.w0", pn.w0), ("p0.w1", pn.w1), ("p1.w0", vm.proposal_networks[1].w0), ("f.base_w0", vm.field.module.base_w0))}
    print("seed", seed, "w0 changed", not torch.equal(before, w0.detach()), info, "step", float(st0[pn.w0]["step"]) if pn.w0 in st0 else None)
