"""Worker of tests/test_gpu_autograd_audit.py (fresh process, NERAF_DETERMINISTIC=1: every sum of the step in a fixed order, so gradients of
two runs of the same computation are the same BITS).  It runs iteration 1 of the trajectory scenario (after an iteration 0 that is
discarded: the compared forward then takes the windowed grid re-conversion and the second feature buffer) in three ways and stores
every parameter gradient of each:

  pipeline    NeRAFPipeline.get_train_loss_dict(1) + ONE backward over the summed loss dict -- as Trainer.train_iteration drives the
              autograd nodes: the grid refresh's node is the first producer of the field gradients, the render loss node the second
              (it ADDS into the first one's tensors in place), the encoder's node converts the refreshed window only;
  perturbed   the same, with everything a caller may legally do between that forward and its backward done in between:
              update_to_step(another step), an eval-mode RIR through the audio model (encoder forward in eval mode), an eval-mode
              render, an eval-mode pipeline loss dict, spatial_distortion switched off and on again;
  by_hand     the three nodes driven by hand in the pipeline's order: vision forward + loss node; the refresh node with the scene
              contraction held OFF from its forward THROUGH its backward; the encoder called directly on the whole grid (no window
              tracking); two SEPARATE backward passes (audio losses, then radiance losses) so that every node is the first producer of
              its pass and torch sums the two field gradients.

    python tests/tools/autograd_audit_worker.py <out.npz> [--mutate contract|eval_ws]
``--mutate`` re-introduces a bug on purpose (the test asserts that the comparison then FAILS): ``contract`` = round 5's (the field
backward maps positions by the module's CURRENT spatial_distortion instead of the one its forward used), ``eval_ws`` = ADVICE r5's
(eval-mode encoder forwards share the training workspace)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch


def main():
    out_path = sys.argv[1]
    mutate = sys.argv[sys.argv.index("--mutate") + 1] if "--mutate" in sys.argv else None
    assert os.environ.get("NERAF_DETERMINISTIC") == "1"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import trajectory_common as TC
    from neraf_amd.model import _RefreshFn
    from neraf_amd.vision import NerfactoField, RayBundle
    cfg = dict(TC.CFG, start_step_audio=-1)

    if mutate == "contract":
        orig = NerfactoField.backward_query

        def buggy(self, *a, **kw):
            kw["contract"] = None                      # round 5's bug: the module's setting at BACKWARD time
            return orig(self, *a, **kw)
        NerfactoField.backward_query = buggy

    def fresh():
        torch.manual_seed(0)
        _, _, _, pipe, _ = TC.run_hip_trajectory(dev, steps=0, cfg=cfg)
        pipe.model.train(); pipe.audio_model.train()
        return pipe

    def params(pipe):
        return {**{"vision." + k: p for k, p in pipe.model.named_parameters()}, **{"audio." + k: p for k, p in pipe.audio_model.named_parameters()}}

    def clear(pipe):
        for p in params(pipe).values():
            p.grad = None

    def grads(pipe):
        torch.cuda.synchronize()
        return {k: p.grad.detach().cpu().numpy().copy() for k, p in params(pipe).items() if p.grad is not None}

    def iteration0(pipe):
        pipe.model.update_to_step(0)
        clear(pipe)
        _, ld, _ = pipe.get_train_loss_dict(0)
        sum(ld.values()).backward()
        clear(pipe)

    res = {}
    # ---- pipeline
    pipe = fresh()
    iteration0(pipe)
    pipe.model.update_to_step(1)
    _, ld, _ = pipe.get_train_loss_dict(1)
    sum(ld.values()).backward()
    res["pipeline"] = grads(pipe)
    losses = {k: float(v.detach()) for k, v in ld.items()}

    # ---- perturbed
    pipe = fresh()
    vm, am = pipe.model, pipe.audio_model
    net = am.resnet3d.backbone_net
    iteration0(pipe)
    vm.update_to_step(1)
    _, ld, _ = pipe.get_train_loss_dict(1)
    if mutate == "eval_ws":
        net._ws_eval = net._ws                          # ADVICE r5's hazard: eval forwards in the training workspace
    vm.update_to_step(777)                              # annealing / proposal schedule state of another step
    evb = TC.rir_bank(2, cfg["tag"] + ".eval")
    am.eval()
    am.get_outputs_for_camera(None, None, batch_audio={"mic_pose": evb["mic_pose"][0], "source_pose": evb["source_pose"][0],
                                                       "rot": evb["rot"][0], "data": evb["log_mag"][0].permute(1, 2, 0)})
    am.train()
    ev = TC.synth.trajectory_eval_camera(8, 8, tag=cfg["tag"])
    vm.eval()
    vm.get_outputs_for_camera_ray_bundle(RayBundle(TC.T(ev["origins"]).to(dev), TC.T(ev["directions"]).to(dev), None))
    vm.train()
    pipe.get_eval_loss_dict(1)                          # eval-mode forwards of BOTH models (pipeline.eval() ... train(was))
    f = vm.field.module
    sd = f.spatial_distortion
    f.spatial_distortion = None
    f.spatial_distortion = sd
    sum(ld.values()).backward()
    res["perturbed"] = grads(pipe)

    # ---- by hand
    pipe = fresh()
    vm, am = pipe.model, pipe.audio_model
    f, net = vm.field.module, am.resnet3d.backbone_net
    iteration0(pipe)
    vm.update_to_step(1)
    bundle, batch = pipe.datamanager.next_train(1)
    outs = vm(bundle)
    ldv = vm.get_loss_dict(outs, batch, vm.get_metrics_dict(outs, batch))
    n = cfg["R"]
    first = am.grid_batch_i
    dirs = am.view_dirs.to(dev).contiguous()
    coords = am.coordinates_to_render[first:first + n].contiguous()
    old = f.spatial_distortion
    f.spatial_distortion = None                         # held off until the refresh node's backward has run
    vals = _RefreshFn.apply(f, coords, f.aabb, dirs, dirs.shape[0], am._delta, am._refresh_consts(dirs, n), (am.grid, first), *f.grad_params())
    am.mark_grid_written()                              # extent not vouched for: the encoder converts the WHOLE grid
    feat = net(am.grid.unsqueeze(0), window=(first, n, 4), window_vals=vals, grid_state=None).flatten()
    _, ba = pipe.audio_datamanager.next_train(1)
    y = am.field.forward_queries(feat, ba["time_query"].to(dev), ba["mic_pose"].to(dev), ba["source_pose"].to(dev), ba["rot"].to(dev),
                                 am.aabb, am.max_len)
    lda = am.get_loss_dict(y, ba, {})
    sum(lda.values()).backward()                        # pass 1: NAcF -> encoder -> refresh node (contraction still off)
    f.spatial_distortion = old
    sum(ldv.values()).backward()                        # pass 2: render loss node; AccumulateGrad adds the field gradients
    res["by_hand"] = grads(pipe)
    losses_hand = {k: float(v.detach()) for k, v in {**ldv, **lda}.items()}

    flat = {}
    for name, g in res.items():
        for k, v in g.items():
            flat[f"{name}/{k}"] = v
    np.savez(out_path, **flat, loss_keys=np.array(sorted(losses)), losses=np.array([losses[k] for k in sorted(losses)]),
             losses_hand=np.array([losses_hand[k] for k in sorted(losses)]))


if __name__ == "__main__":
    main()
