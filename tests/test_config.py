"""The method specification as code (neraf_amd/config.py) against the values of NeRAF/NeRAF_config.py:33-139, the scheduler
formula, and config.setup() -> pipeline instantiation (NeRAF_pipeline.py:86-159) on CPU (no kernels run: construction only)."""
import math

import numpy as np
import pytest
import torch

from neraf_amd import config as C


def test_method_values_match_the_reference_config():
    m = C.make_method("RAF", "FurnishedRoom")
    t = m.config
    assert (t.method_name, t.max_num_iterations, t.mixed_precision, t.steps_per_save, t.steps_per_eval_all_images) == ("NeRAF", 400001, True, 20000, 10000)
    p = t.pipeline
    assert p.start_step_audio == 2000 and p.vision_model.eval_num_rays_per_chunk == 1 << 15 and p.vision_model.average_init_density == 0.01
    assert p.vision_model.camera_optimizer.mode == "SO3xR3"
    a = p.audio_model
    assert (a.dataset, a.use_grid, a.grid_step, a.N_features, a.loss_factor, a.W_field, a.criterion, a.fs, a.max_len) == \
        ("RAF", True, 1 / 128, 1024, 1e-3, 512, "SC+SLMSE", 48000, 0.32)
    o = t.optimizers
    assert list(o) == ["proposal_networks", "fields", "audio_fields", "camera_opt"]
    assert [(o[k]["optimizer"].lr, o[k]["optimizer"].eps) for k in o] == [(1e-2, 1e-15), (1e-2, 1e-15), (1e-4, 1e-15), (1e-3, 1e-15)]
    s = o["audio_fields"]["scheduler"]
    assert (s.lr_final, s.max_steps, s.warmup_steps) == (1e-8, 1002000, 2000)
    assert (o["camera_opt"]["scheduler"].lr_final, o["camera_opt"]["scheduler"].max_steps) == (1e-4, 5000)
    ss = C.make_method("SoundSpaces", "apartment_1").config.pipeline.audio_model
    assert (ss.fs, ss.max_len) == (22050, 101)


def test_exponential_decay_scheduler_formula():
    s = C.ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=200000)
    np.testing.assert_allclose(s.lr_at(0, 1e-2), 1e-2)
    np.testing.assert_allclose(s.lr_at(100000, 1e-2), 1e-3, rtol=1e-12)          # log-linear midpoint
    np.testing.assert_allclose(s.lr_at(200000, 1e-2), 1e-4)
    np.testing.assert_allclose(s.lr_at(300000, 1e-2), 1e-4)                      # clamped after max_steps
    w = C.ExponentialDecaySchedulerConfig(lr_final=1e-8, max_steps=1002000, warmup_steps=2000)
    np.testing.assert_allclose(w.lr_at(0, 1e-4), 1e-8)
    np.testing.assert_allclose(w.lr_at(1000, 1e-4), 1e-8 + (1e-4 - 1e-8) * math.sin(0.25 * math.pi), rtol=1e-12)
    np.testing.assert_allclose(w.lr_at(2000, 1e-4), 1e-4)
    np.testing.assert_allclose(w.lr_at(502000, 1e-4), 1e-6, rtol=1e-9)
    # as a torch LambdaLR on a real optimizer
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([p], lr=1e-2)
    sched = s.setup(opt, 1e-2)
    for _ in range(10):
        opt.step(); sched.step()
    np.testing.assert_allclose(opt.param_groups[0]["lr"], s.lr_at(10, 1e-2), rtol=1e-12)


def test_pipeline_config_setup_instantiates_like_the_reference():
    from neraf_amd.datamanagers import SyntheticAudioDataManager, SyntheticVisionDataManager
    m = C.make_method("RAF", "FurnishedRoom", datamanager=SyntheticVisionDataManager(5, 2, 32, 48, 128),
                      audio_datamanager=SyntheticAudioDataManager(3, 2, batch_size=32))
    pipe = m.config.pipeline.setup(device="cpu", test_mode="val", world_size=1, local_rank=0, grad_scaler=None)
    assert pipe.start_step_audio == 2000 and pipe.model.audio_model is pipe.audio_model
    assert pipe.model.num_train_data == 5 and pipe.model.field.module.embedding.shape == (5, 32)
    assert pipe.audio_model.spatial_distortion == pipe.model.field.module.spatial_distortion
    assert pipe.audio_model.eval_gt is not None                                  # set_eval_data(eval_dataset[0]...), :147
    g = pipe.get_param_groups()
    assert list(g) == ["proposal_networks", "fields", "camera_opt", "audio_fields"]
    assert g["camera_opt"][0].shape == (5, 6)
    assert all(any(p is q for q in g["audio_fields"]) for p in g["fields"])       # :487
    # state-dict key prefixes of the reference (checkpoint row)
    keys = pipe.state_dict().keys()
    for k in ("_model.field.module.table", "_model.proposal_networks.0.table", "audio_model.field.soundfield.0.weight",
              "audio_model.resnet3d.backbone_net.conv1.weight", "audio_model.grid"):
        assert k in keys, k
    rb, batch = pipe.datamanager.next_train(0)
    assert rb.origins.shape == (128, 3) and batch["image"].shape == (128, 3)
    _, ba = pipe.audio_datamanager.next_train(0)
    assert ba["data"].shape == (32, 1, 513) and ba["time_query"].shape == (32,)


def test_optimizers_wrapper_on_cpu_parameters_uses_torch_adam_and_schedules():
    ps = {"fields": [torch.nn.Parameter(torch.ones(4))], "audio_fields": [torch.nn.Parameter(torch.ones(2))],
          "camera_opt": [torch.nn.Parameter(torch.zeros(3, 6))]}
    ps["audio_fields"].append(ps["fields"][0])
    opts = C.Optimizers(C.default_optimizers(2000), ps)
    assert set(opts.optimizers) == {"fields", "audio_fields", "camera_opt"} and len(opts.steppers) == 3
    for step in range(3):
        opts.zero_grad_all()
        for g in ps.values():
            for p in g:
                p.grad = torch.ones_like(p)
        opts.optimizer_step_all()
        opts.scheduler_step_all(step)
    np.testing.assert_allclose(opts.get_lr("fields"), C.default_optimizers()["fields"]["scheduler"].lr_at(3, 1e-2), rtol=1e-12)
    np.testing.assert_allclose(opts.get_lr("audio_fields"), C.default_optimizers()["audio_fields"]["scheduler"].lr_at(3, 1e-4), rtol=1e-12)
    np.testing.assert_allclose(opts.get_lr("camera_opt"), C.default_optimizers()["camera_opt"]["scheduler"].lr_at(3, 1e-3), rtol=1e-12)
