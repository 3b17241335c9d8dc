#!/usr/bin/env python3
"""Generate golden vectors from the *imported* reference (container-only tool).

Runs ONLY where /root/reference exists (this build container).  It imports the
reference's pure-torch modules (NeRAF_resnet3d, NeRAF_field, NeRAF_evaluator,
NeRAF_helper, and NeRAF_model for the grid index/scatter arithmetic) with the
absent third-party packages (nerfstudio, tinycudann, torchaudio, librosa,
pyroomacoustics, jaxtyping, cv2) stubbed at ``sys.meta_path``, drives them
with inputs/weights regenerated from ``neraf_amd.synth`` (integer PRNG keyed by
name), and writes small ``.npz`` fixtures under tests/golden/.  Nothing of the
reference (source or bytecode) is written anywhere; fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/tools/gen_golden.py

Fixtures: G1 resnet3d, G2 nacf, G3 stft_loss, G4 grid scatter,
G5 helper metrics, G6 dataparsers (poses / rotations / scene boxes of synthetic RAF and SoundSpaces trees).
"""
import enum
import importlib.abc
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from neraf_amd import synth

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
STUB_ROOTS = ("nerfstudio", "torchaudio", "pyroomacoustics", "librosa", "jaxtyping", "cv2", "tyro", "rich",
              "matplotlib", "tinycudann")


class _Dummy:
    def __init__(self, *a, **k):
        self._kw = k                 # e.g. SceneBox(aabb=...): the argument is what the fixture records

    def __enter__(self):             # rich.progress.Progress(...) as a context manager
        return self

    def __exit__(self, *a):
        return False

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Dummy()

    def __class_getitem__(cls, item):
        return cls


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Dummy,), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    import NeRAF.NeRAF_resnet3d as r3
    import NeRAF.NeRAF_field as nf
    import NeRAF.NeRAF_evaluator as ne
    import NeRAF.NeRAF_helper as nh
    import NeRAF.NeRAF_model as nm
    return r3, nf, ne, nh, nm


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def stats(x: torch.Tensor):
    x = x.detach().double()
    return np.array([x.mean().item(), x.abs().mean().item(), x.pow(2).mean().sqrt().item()], np.float64)


# --------------------------------------------------------------------------
def g1_resnet3d(r3):
    """ResNet3D_helper(7,'resnet50',grid_step,1024) -- NeRAF_resnet3d.py:266-285."""
    torch.manual_seed(0)
    sd_np = synth.resnet3d_state_dict(7)
    for S, gs, tag in ((64, 1 / 64, "g1_resnet3d_64"), (128, 1 / 128, "g1_resnet3d_128")):
        net = r3.ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=gs, N_features=1024)
        missing = net.backbone_net.load_state_dict({k: t(v) for k, v in sd_np.items()}, strict=True)
        x = t(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).requires_grad_(True)
        wsum = t(synth.uniform("g1.outw", (1024,), -1.0, 1.0))
        out = {}
        # --- train mode (batch statistics) with stage hooks
        net.train()
        stage = {}
        bb = net.backbone_net
        hooks = [
            bb.relu.register_forward_hook(lambda m, i, o: stage.__setitem__("conv1", o.detach().clone())),
            bb.maxpool.register_forward_hook(lambda m, i, o: stage.__setitem__("maxpool", o.detach().clone())),
            bb.layer1.register_forward_hook(lambda m, i, o: stage.__setitem__("layer1", o.detach().clone())),
            bb.layer2.register_forward_hook(lambda m, i, o: stage.__setitem__("layer2", o.detach().clone())),
            bb.layer3.register_forward_hook(lambda m, i, o: stage.__setitem__("layer3", o.detach().clone())),
        ]
        # momentum must not move running stats between the train and eval passes
        for m in bb.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.momentum = 0.0
        y = net(x)
        for h in hooks:
            h.remove()
        loss = (y.flatten() * wsum).sum()
        loss.backward()
        out["out_train"] = y.detach().flatten().numpy()
        for k, v in stage.items():
            out["stage_" + k] = stats(v)
        # a fixed slab of the first-stage activation so layout bugs show up as more than a checksum drift
        out["conv1_slab"] = stage["conv1"][0, :8, 3, 5, :16].numpy()
        out["layer3_slab"] = stage["layer3"][0, :16, 1, 2, :].numpy()
        gx = x.grad.detach()
        probes = synth.integers("g1.probes", (64, 4), 0, 10 ** 9)
        probes = np.stack([probes[:, 0] % 7, probes[:, 1] % S, probes[:, 2] % S, probes[:, 3] % S], 1)
        out["probe_idx"] = probes
        out["dx_probe"] = gx[0, probes[:, 0], probes[:, 1], probes[:, 2], probes[:, 3]].numpy()
        out["dx_stats"] = stats(gx)
        out["dw_conv1"] = bb.conv1.weight.grad.detach().numpy()
        out["dw_l1_0_conv2_stats"] = stats(bb.layer1[0].conv2.weight.grad)
        out["dw_l1_0_conv2_slab"] = bb.layer1[0].conv2.weight.grad[:4, :4].detach().numpy()
        out["dw_l3_5_conv3_stats"] = stats(bb.layer3[5].conv3.weight.grad)
        out["dw_l3_5_conv3_slab"] = bb.layer3[5].conv3.weight.grad[:8, :8, 0, 0, 0].detach().numpy()
        out["dgamma_bn1"] = bb.bn1.weight.grad.detach().numpy()
        out["dbeta_bn1"] = bb.bn1.bias.grad.detach().numpy()
        out["dgamma_l2_0_ds"] = bb.layer2[0].downsample[1].weight.grad.detach().numpy()
        # --- eval mode (running statistics)
        net.eval()
        with torch.no_grad():
            out["out_eval"] = net(x).flatten().numpy()
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
        print(tag, "out_train", stats(y), "dx", out["dx_stats"])


def g1v_resnet3d_variants(r3, which=None):
    """The other configurations the reference's constructor accepts (NeRAF_resnet3d.py:128-156): N_features = 2048 (layer4) on the
    64^3 and 128^3 grids, and grid_step = 1/256 (7 x 256^3) with N_features = 1024.  Same content as G1, with the last layer's
    stages; fixtures g1_resnet3d_<S>_<N>.npz."""
    torch.manual_seed(0)
    for S, N in ((64, 2048), (128, 2048), (256, 1024)):
        tag = f"g1_resnet3d_{S}_{N}"
        if which and tag not in which:
            continue
        layers = (3, 4, 6, 3) if N == 2048 else (3, 4, 6)
        sd_np = synth.resnet3d_state_dict(7, layers=layers)
        net = r3.ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=1 / S, N_features=N)
        net.backbone_net.load_state_dict({k: t(v) for k, v in sd_np.items()}, strict=True)
        x = t(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).requires_grad_(True)
        wsum = t(synth.uniform(f"g1.outw{N}", (N,), -1.0, 1.0))
        out = {}
        net.train()
        stage = {}
        bb = net.backbone_net
        names = ["layer1", "layer2", "layer3"] + (["layer4"] if N == 2048 else [])
        hooks = [bb.relu.register_forward_hook(lambda m, i, o: stage.__setitem__("conv1", o.detach().clone())),
                 bb.maxpool.register_forward_hook(lambda m, i, o: stage.__setitem__("maxpool", o.detach().clone()))]
        for nm in names:
            hooks.append(getattr(bb, nm).register_forward_hook(lambda m, i, o, nm=nm: stage.__setitem__(nm, o.detach().clone())))
        for m in bb.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.momentum = 0.0
        y = net(x)
        for h in hooks:
            h.remove()
        assert tuple(y.shape) == (1, N, 1, 1, 1), y.shape
        (y.flatten() * wsum).sum().backward()
        out["out_train"] = y.detach().flatten().numpy()
        for k, v in stage.items():
            out["stage_" + k] = stats(v)
        last = names[-1]
        out["conv1_slab"] = stage["conv1"][0, :8, 3, 5, :16].numpy()
        out["last_slab"] = stage[last][0, :16, 1, 1, :].numpy()
        gx = x.grad.detach()
        probes = synth.integers("g1.probes", (64, 4), 0, 10 ** 9)
        probes = np.stack([probes[:, 0] % 7, probes[:, 1] % S, probes[:, 2] % S, probes[:, 3] % S], 1)
        out["probe_idx"] = probes
        out["dx_probe"] = gx[0, probes[:, 0], probes[:, 1], probes[:, 2], probes[:, 3]].numpy()
        out["dx_stats"] = stats(gx)
        out["dw_conv1"] = bb.conv1.weight.grad.detach().numpy()
        out["dw_l1_0_conv2_stats"] = stats(bb.layer1[0].conv2.weight.grad)
        lastl = getattr(bb, last)
        out["dw_last_conv3_stats"] = stats(lastl[-1].conv3.weight.grad)
        out["dw_last_conv3_slab"] = lastl[-1].conv3.weight.grad[:8, :8, 0, 0, 0].detach().numpy()
        out["dw_last_0_conv2_stats"] = stats(lastl[0].conv2.weight.grad)
        out["dgamma_bn1"] = bb.bn1.weight.grad.detach().numpy()
        out["dbeta_bn1"] = bb.bn1.bias.grad.detach().numpy()
        out["dgamma_last_0_ds"] = lastl[0].downsample[1].weight.grad.detach().numpy()
        net.eval()
        with torch.no_grad():
            out["out_eval"] = net(x).flatten().numpy()
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
        print(tag, "out_train", stats(y), "dx", out["dx_stats"], flush=True)


def g2_nacf(nf):
    """NeRAFAudioSoundField(1187,512,sound_rez,N_frequencies) -- NeRAF_field.py:37-65."""
    for C, Fq, tag in ((1, 513, "g2_nacf_raf"), (2, 257, "g2_nacf_ss")):
        sd_np = synth.nacf_state_dict(1187, 512, C, Fq)
        net = nf.NeRAFAudioSoundField(1187, 512, sound_rez=C, N_frequencies=Fq)
        net.load_state_dict({k: t(v) for k, v in sd_np.items()}, strict=True)
        h = t(synth.uniform("g2.h", (8, 1187), -1.0, 1.0)).requires_grad_(True)
        wout = t(synth.uniform("g2.wout", (8, C, Fq), -1.0, 1.0))
        y = net(h)
        (y * wout).sum().backward()
        np.savez_compressed(
            os.path.join(OUT, tag + ".npz"),
            out=y.detach().numpy(), dh=h.grad.numpy(),
            dw0_slab=net.soundfield[0].weight.grad[:4, :8].numpy(),
            dw0_stats=stats(net.soundfield[0].weight.grad),
            db0=net.soundfield[0].bias.grad[:16].numpy(),
            dw4_stats=stats(net.soundfield[4].weight.grad),
            dwh0_slab=net.STFT_linear[0].weight.grad[:4, :8].numpy(),
            dbh_last=net.STFT_linear[C - 1].bias.grad.numpy(),
        )
        print(tag, stats(y))


def g3_stft_loss(ne):
    """STFTLoss('mse'|'l1') -- NeRAF_evaluator.py:76-108."""
    out = {}
    for C, Fq in ((1, 513), (2, 257)):
        x = t(synth.uniform(f"g3.x{C}", (8, C, Fq), -6.0, 2.0)).requires_grad_(True)
        y = t(synth.uniform(f"g3.y{C}", (8, C, Fq), -6.0, 2.0))
        for lt in ("mse", "l1"):
            crit = ne.STFTLoss(loss_type=lt)
            d = crit(x, y)
            sc, mag = d["audio_sc_loss"], d["audio_mag_loss"]
            x.grad = None
            (sc * 0.1 * 1e-3 + mag * 1.0 * 1e-3).backward()
            out[f"sc_{lt}_{C}"] = np.array(sc.item(), np.float64)
            out[f"mag_{lt}_{C}"] = np.array(mag.item(), np.float64)
            out[f"dx_{lt}_{C}"] = x.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g3_stft_loss.npz"), **out)
    print("g3", {k: float(v) for k, v in out.items() if v.ndim == 0})


# --------------------------------------------------------------------------
class _Frustums:
    def __init__(self, origins, directions, starts, ends, pixel_area):
        self.origins, self.directions, self.starts, self.ends = origins, directions, starts, ends


class _RaySamples:
    def __init__(self, frustums, camera_indices=None):
        self.frustums, self.camera_indices = frustums, camera_indices

    def to(self, device):
        return self


class _FieldHeadNames(enum.Enum):
    RGB = "rgb"
    DENSITY = "density"


def _renderer_rgb(rgb, weights):
    # nerfstudio RGBRenderer.combine_rgb, background "last_sample" [NS-recall]
    comp = torch.sum(weights * rgb, dim=-2)
    acc = torch.sum(weights, dim=-2)
    return comp + rgb[..., -1, :] * (1.0 - acc)


def g4_grid(nm):
    """query_grid_one_batch index/scatter arithmetic -- NeRAF_model.py:294-407 driven with stand-ins."""
    nm.Frustums, nm.RaySamples, nm.FieldHeadNames = _Frustums, _RaySamples, _FieldHeadNames
    M = nm.NeRAFAudioModel
    out = {}
    for gs, bs, tag in ((1 / 16, 1500, "s16"), (1 / 128, 4096, "s128")):
        self = types.SimpleNamespace()
        self.use_grid = True
        self.grid_size = np.array([0, 1, 0, 1, 0, 1])
        self.grid_step = gs
        self.device = "cpu"
        self._delta = 1e-2
        self.spatial_distortion = "SENTINEL"
        self.reset_grid = types.MethodType(M.reset_grid, self)
        self.view_dirs = M._generate_fixed_viewing_directions(self)
        # NeRAF_model.py:200-203
        ax = torch.arange(0 + gs / 2, 1, gs)
        gc = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1)
        self.coordinates_to_render = gc.view(-1, 3)
        self.grid_batch_i = 0
        self.grid = None
        aabb = torch.tensor([[-3.5, -2.0, -4.5], [4.0, 2.5, 5.0]])
        queried = []

        def fwd(ray, compute_normals=False):
            pos = ray.frustums.origins + ray.frustums.directions * (ray.frustums.starts + ray.frustums.ends) / 2
            queried.append((ray.frustums.origins.clone(), ray.frustums.directions.clone()))
            rgb, dens = synth.toy_field(pos, ray.frustums.directions)
            return {_FieldHeadNames.RGB: rgb, _FieldHeadNames.DENSITY: dens}

        vf = types.SimpleNamespace(module=types.SimpleNamespace(spatial_distortion="X", aabb=aabb), forward=fwd)
        if tag == "s16":
            out["view_dirs"] = self.view_dirs.numpy()
            steps = 4  # 1500,1500,1096(partial)+wrap, then 1500 again
        else:
            steps = 2
            self.grid_batch_i = 128 ** 3 - 4096 - 1000  # second step is the last partial batch -> wrap
        cursors = []
        for s in range(steps):
            M.query_grid_one_batch(self, s, vf, renderer_rgb=_renderer_rgb, batch_size=bs)
            cursors.append(self.grid_batch_i)
            assert vf.module.spatial_distortion == "SENTINEL"
        out[f"cursors_{tag}"] = np.array(cursors, np.int64)
        o0, d0 = queried[0]
        out[f"ori_first_{tag}"] = o0[:6].numpy()
        out[f"ori_stats_{tag}"] = stats(o0)
        out[f"dir_rows_{tag}"] = d0[:: o0.shape[0] // 18][:18].numpy()
        if tag == "s16":
            out["grid_s16"] = self.grid[:4].numpy().copy()
            out["grid_coords_s16"] = self.grid[4:, :2, :2, :].numpy().copy()
        else:
            g = self.grid
            out["grid_stats_s128"] = np.stack([stats(g[c]) for c in range(7)])
            nz = (g[3] != 0).nonzero()
            out["grid_nnz_s128"] = np.array(nz.shape[0], np.int64)
            out["grid_first_nz_s128"] = nz[:4].numpy()
            out["grid_last_nz_s128"] = nz[-4:].numpy()
            out["grid_slab_s128"] = g[:4, 0, 0, :64].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g4_grid.npz"), **out)
    print("g4 cursors", out["cursors_s16"], out["cursors_s128"], "nnz128", out["grid_nnz_s128"])


def g5_helper(nh):
    """numpy metric kernels -- NeRAF_helper.py:79-161 on a synthetic exponentially decaying IR."""
    fs = 48000
    n = 15360
    tt = np.arange(n) / fs
    out = {}
    irs = []
    for c, tau in enumerate((0.05, 0.08)):
        noise = synth.normal(f"g5.noise{c}", (n,), 1.0, np.float64)
        irs.append(noise * np.exp(-tt / tau))
    gt = np.stack(irs)
    pred = gt * 0.9 + 0.05 * np.stack([synth.normal(f"g5.pert{c}", (n,), 1.0, np.float64) * np.exp(-tt / 0.06)
                                       for c in range(2)])
    out["edt_gt"], out["edt_pred"] = nh.evaluate_edt(pred, gt, fs)
    out["c50_gt"], out["c50_pred"] = nh.evaluate_clarity(pred, gt, fs)
    out["envelope"] = np.array(nh.Envelope_distance(pred, gt))
    out["snr"] = np.array(nh.SNR(pred, gt))
    out["magdist"] = np.array(nh.Magnitude_distance(np.abs(pred), np.abs(gt)))
    sl = nh.SpectralLoss(base_loss=torch.nn.functional.mse_loss, epsilon=1, dB=False, stft_input_type="mag")
    out["spectral"] = np.array(sl(t(np.abs(pred)), t(np.abs(gt))).item())
    np.savez_compressed(os.path.join(OUT, "g5_helper.npz"), **out)
    print("g5", {k: np.asarray(v).tolist() for k, v in out.items()})


def g6_dataparsers():
    """The reference's own dataparsers (NeRAF_dataparser.py: RAFDataParser / SoundSpacesDataParser._generate_dataparser_outputs) on
    synthetic scene trees in the datasets' on-disk formats (neraf_amd.synth.raf_tree / soundspaces_tree): poses, direction cosines
    and the audio scene box of every split."""
    import tempfile
    import types as _types
    import NeRAF.NeRAF_dataparser as ndp
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        synth.write_tree(tmp, synth.raf_tree())
        parser = object.__new__(ndp.RAFDataParser)
        parser.config = _types.SimpleNamespace(data=tmp)
        for split in ("train", "val", "test"):
            o = parser._generate_dataparser_outputs(split)
            out[f"raf_{split}_mic"] = o.microphone_poses.numpy()
            out[f"raf_{split}_src"] = o.source_poses.numpy()
            out[f"raf_{split}_rot"] = o.source_rotations.numpy()
            out[f"raf_{split}_aabb"] = o.scene_box._kw["aabb"].numpy()
            out[f"raf_{split}_names"] = np.array(list(o.audios_filenames))
    with tempfile.TemporaryDirectory() as tmp:
        synth.write_tree(tmp, synth.soundspaces_tree())
        parser = object.__new__(ndp.SoundSpacesDataParser)
        parser.config = _types.SimpleNamespace(data=tmp)
        for split in ("train", "test"):
            o = parser._generate_dataparser_outputs(split)
            out[f"ss_{split}_mic"] = o.microphone_poses.numpy()
            out[f"ss_{split}_src"] = o.source_poses.numpy()
            out[f"ss_{split}_rot"] = o.microphone_rotations.numpy()
            out[f"ss_{split}_aabb"] = o.scene_box._kw["aabb"].numpy()
            out[f"ss_{split}_names"] = np.array(list(o.audios_filenames))
    # the 'inference' splits: pose files named by AVN_RENDER_POSES (NeRAF_dataparser.py:130-138, :311-322)
    import pickle
    with tempfile.TemporaryDirectory() as tmp:
        raf_inf, ss_inf = synth.inference_pose_files()
        p1 = os.path.join(tmp, "raf_poses.npy")
        np.save(p1, raf_inf, allow_pickle=True)
        os.environ["AVN_RENDER_POSES"] = p1
        parser = object.__new__(ndp.RAFDataParser)
        parser.config = _types.SimpleNamespace(data=tmp)
        o = parser._generate_dataparser_outputs("inference")
        out["raf_inf_mic"], out["raf_inf_src"], out["raf_inf_rot"] = (np.asarray(o.microphone_poses), np.asarray(o.source_poses),
                                                                      np.asarray(o.source_rotations))
        out["raf_inf_aabb"] = o.scene_box._kw["aabb"].numpy()
        synth.write_tree(tmp, synth.soundspaces_tree())
        p2 = os.path.join(tmp, "ss_poses.pkl")
        with open(p2, "wb") as f:
            pickle.dump(ss_inf, f)
        os.environ["AVN_RENDER_POSES"] = p2
        parser = object.__new__(ndp.SoundSpacesDataParser)
        parser.config = _types.SimpleNamespace(data=tmp)
        o = parser._generate_dataparser_outputs("inference")
        out["ss_inf_mic"], out["ss_inf_src"], out["ss_inf_rot"] = (np.asarray(o.microphone_poses), np.asarray(o.source_poses),
                                                                   np.asarray(o.microphone_rotations))
        out["ss_inf_aabb"] = o.scene_box._kw["aabb"].numpy()
    np.savez_compressed(os.path.join(OUT, "g6_dataparsers.npz"), **out)
    print("g6", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    assert os.path.isdir(REF), "container-only tool: /root/reference is absent"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    r3, nf, ne, nh, nm = import_reference()
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6"]
    if "g2" in which:
        g2_nacf(nf)
    if "g3" in which:
        g3_stft_loss(ne)
    if "g4" in which:
        g4_grid(nm)
    if "g5" in which:
        g5_helper(nh)
    if "g6" in which:
        g6_dataparsers()
    if "g1" in which:
        g1_resnet3d(r3)
    if any(w.startswith("g1v") for w in which):        # "g1v" = all three, "g1v:g1_resnet3d_64_2048" = one
        sel = [w.split(":", 1)[1] for w in which if w.startswith("g1v:")]
        g1v_resnet3d_variants(r3, sel or None)
