#!/usr/bin/env python3
"""ResNet3D forward (train-mode BatchNorm) of one fixed grid, repeated: relative deviation of the 1024-feature from the first
evaluation.  Alone the only source is the order of the BatchNorm-statistic atomics (1e-3 level through flipped ReLU gates); `--pair`
runs two such processes at once on the one GPU."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--pair", action="store_true"); ap.add_argument("--iters", type=int, default=400); ap.add_argument("--tag", default="A")
ap.add_argument("--size", type=int, default=64); ap.add_argument("--graphs", default=None)
ap.add_argument("--pad-mib", type=int, default=0, help="allocate this much device memory first (shifts every later device address)")
ap.add_argument("--hold", type=float, default=0.0, help="after the reference evaluation: touch /tmp/probe_ref_ready and wait this many seconds (so that a neighbour can be started against a clean reference)")
ap.add_argument("--eval", action="store_true", help="eval-mode BatchNorm (running statistics: no statistic atomics, deterministic)")
a = ap.parse_args()
if a.graphs is not None:
    os.environ["NERAF_GRAPHS"] = a.graphs
if a.pair:
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(a.iters), "--tag", t, "--size", str(a.size)] + (["--graphs", a.graphs] if a.graphs else [])
                           + (["--pad-mib", str(a.pad_mib)] if t == "B" and a.pad_mib else [])) for t in ("A", "B")]
    sys.exit(max(p.wait() for p in ps))
import numpy as np, torch
from neraf_amd import synth
from neraf_amd.resnet3d import ResNet3D_helper
dev = torch.device("cuda:0")
pad = torch.empty(a.pad_mib << 20, dtype=torch.uint8, device=dev) if a.pad_mib else None
net = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=1.0 / a.size, N_features=1024)
net.backbone_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.resnet3d_state_dict(7).items()})
net.to(dev).train()
if a.eval:
    net.eval()
x = torch.from_numpy(synth.uniform("probe.grid", (1, 7, a.size, a.size, a.size), 0.0, 1.0)).to(dev)
import ctypes as C
from neraf_amd import _lib
lib = _lib.load()
bb = net.backbone_net


def ws_tensor(kind, index, dtype):
    off, rows, cols = C.c_size_t(), C.c_int(), C.c_int()
    assert lib.neraf_resnet3d_debug_locate(C.byref(bb._desc), kind, index, C.byref(off), C.byref(rows), C.byref(cols)) == 0
    nbytes = rows.value * cols.value * torch.empty((), dtype=dtype).element_size()
    return bb._ws[off.value:off.value + nbytes].view(dtype).reshape(rows.value, cols.value)


def snapshot():
    """pre-BN outputs of all 43 convolutions (kind 5) and their published batch statistics (kind 6)"""
    return [ws_tensor(5, i, torch.float16).clone() for i in range(43)], [ws_tensor(6, i, torch.float32).clone() for i in range(43)]


def snapshot_acts():
    """post-activation tensors a1 / a2 / out of the 13 blocks (kinds 0, 1, 2)"""
    return [[ws_tensor(k, b, torch.float16).clone() for k in (0, 1, 2)] for b in range(13)]


reported = 0
with torch.no_grad():
    ref = net(x).flatten().clone()
    ref_pre, ref_stat = snapshot()
    ref_act = snapshot_acts()
    ref_pool = ws_tensor(3, 0, torch.float16).clone()
    if a.hold > 0:
        import time
        torch.cuda.synchronize(); open("/tmp/probe_ref_ready", "w").close(); time.sleep(a.hold)
    devs = []
    for it in range(a.iters):
        f = net(x).flatten()
        d = float((f - ref).norm() / ref.norm())
        devs.append(d)
        if d > 0.05:
            print(f"[{a.tag}] evaluation {it}: feature deviates by rel-L2 {d:.3f}", flush=True)
            if reported < 3:
                reported += 1
                pre, stat = snapshot()
                rel = [float((p.float() - r.float()).norm() / (r.float().norm() + 1e-30)) for p, r in zip(pre, ref_pre)]
                rels = [float((p - r).norm() / (r.norm() + 1e-30)) for p, r in zip(stat, ref_stat)]
                act = snapshot_acts()
                arel = [[round(float((p.float() - r.float()).norm() / (r.float().norm() + 1e-30)), 4) for p, r in zip(pb, rb)] for pb, rb in zip(act, ref_act)]
                print(f"[{a.tag}]   activations (a1, a2, out) per block: {arel}", flush=True)
                # shape of the damage in the first bad activation: which channels, which rows, which workgroups of bn_apply_kernel
                # (grid-stride over 16-byte chunks, min(2048, chunks/256) workgroups of 256 lanes)
                hit = next(((b, k) for b in range(13) for k in range(3) if arel[b][k] > 0.02), None)
                if hit:
                    got, want = act[hit[0]][hit[1]].float(), ref_act[hit[0]][hit[1]].float()
                    bad = (got - want).abs() > 0.05 * want.abs().max()
                    rows, cols = bad.shape
                    chunks = bad.reshape(rows, cols // 8, 8).any(-1).flatten().nonzero().flatten().cpu().numpy()
                    total = rows * (cols // 8)
                    grid = min(2048, (total + 255) // 256)
                    wg = (chunks // 256) % grid
                    bad_rows = bad.any(1).nonzero().flatten().cpu().numpy(); bad_cols = bad.any(0).nonzero().flatten().cpu().numpy()
                    print(f"[{a.tag}]   block {hit[0]} tensor {'a1 a2 out'.split()[hit[1]]} [{rows} x {cols}]: {int(bad.sum())} bad elements "
                          f"({float(bad.float().mean()):.4f}), {len(chunks)} bad chunks in {len(np.unique(wg))} of {grid} workgroups "
                          f"{np.unique(wg)[:12].tolist()}, passes {np.unique(chunks // (256 * grid))[:8].tolist()}, rows {len(bad_rows)} "
                          f"[{bad_rows[:4].tolist()}..{bad_rows[-2:].tolist()}], channels {len(bad_cols)} [{bad_cols[:6].tolist()}..]; "
                          f"got range [{float(got.min()):.3g}, {float(got.max()):.3g}] want [{float(want.min()):.3g}, {float(want.max()):.3g}]", flush=True)
                # was the damaged part of a1 / a2 normalised with ZERO statistics (sum = sum of squares = 0: scale = gamma / sqrt(eps))?
                hit01 = next(((b, k) for b in range(13) for k in range(2) if arel[b][k] > 0.02), None)
                if hit01:
                    b, k = hit01
                    idx = 1
                    for bb_ in range(b):
                        idx += 4 if bb_ in (0, 3, 7) else 3
                    ci = idx + k
                    bnm = bb._host_tables()["pairs"][ci][1]
                    xpre = ws_tensor(5, ci, torch.float16).float()
                    got, want = act[b][k].float(), ref_act[b][k].float()
                    badrow = ((got - want).abs() > 0.05 * want.abs().max()).any(1)
                    zero_stats = torch.relu(xpre * (bnm.weight.float() * (1e-5 ** -0.5)) + bnm.bias.float())
                    r = badrow.nonzero().flatten()
                    e0 = float((got[r] - zero_stats[r]).norm() / (got[r].norm() + 1e-30)); e1 = float((got[r] - want[r]).norm() / (got[r].norm() + 1e-30))
                    if 0 < len(r) <= 64:
                        # are the damaged rows some OTHER rows' values (of this tensor, or of any activation of the same width)?
                        msgs = []
                        for rr in r[:4].tolist():
                            best = (1e30, None)
                            for b2 in range(13):
                                for k2 in range(3):
                                    cand = ref_act[b2][k2].float()
                                    if cand.shape[1] != got.shape[1]:
                                        continue
                                    dist = (cand - got[rr]).norm(dim=1) / (got[rr].norm() + 1e-30)
                                    j = int(dist.argmin())
                                    if float(dist[j]) < best[0]:
                                        best = (float(dist[j]), (b2, k2, j))
                            nz = int((got[rr] != 0).sum())
                            msgs.append(f"row {rr}: {nz} non-zeros, max {float(got[rr].max()):.3g} (reference max {float(want[rr].max()):.3g}), nearest reference row anywhere {best[1]} at rel distance {best[0]:.3f}; "
                                        f"x row max |x| {float(xpre[rr].abs().max()):.3g}")
                        print(f"[{a.tag}]     " + " | ".join(msgs), flush=True)
                    print(f"[{a.tag}]   block {b} a{k + 1} (conv {ci}): {len(r)} damaged rows; |got - bn(x; zero statistics)| / |got| = {e0:.4f}, |got - reference| / |got| = {e1:.4f}", flush=True)
                first = next((i for i, v in enumerate(rel) if v > 0.02), None)
                if first is not None:
                    got, want = pre[first].float(), ref_pre[first].float()
                    bad = (got - want).abs() > 0.02 * want.abs().max()

                    def ranges(idx):
                        idx = idx.cpu().numpy(); out = []; lo = prev = None
                        for v in idx:
                            if lo is None: lo = prev = v
                            elif v == prev + 1: prev = v
                            else: out.append((int(lo), int(prev))); lo = prev = v
                        if lo is not None: out.append((int(lo), int(prev)))
                        return out[:12]
                    rr, cc = bad.any(1).nonzero().flatten(), bad.any(0).nonzero().flatten()
                    ij = bad.nonzero()[:6]
                    samples = [(int(i), int(j), round(float(got[i, j]), 3), round(float(want[i, j]), 3)) for i, j in ij]
                    # the same for the tensor that convolution READ (as it is now, after the forward): damaged input = sticky, clean = transient
                    src = {}
                    idx = 1
                    for b in range(13):
                        xin = (3, 0) if b == 0 else (2, b - 1)
                        src[idx] = xin; src[idx + 1] = (0, b); src[idx + 2] = (1, b)
                        if b in (0, 3, 7): src[idx + 3] = xin; idx += 4
                        else: idx += 3
                    if first in src:
                        k, b = src[first]
                        now = ws_tensor(k, b, torch.float16).float()
                        was = (ref_pool if k == 3 else ref_act[b][k]).float()
                        ibad = (now - was).abs() > 0.05 * was.abs().max()
                        print(f"[{a.tag}]   its input (kind {k}, block {b}) [{now.shape[0]} x {now.shape[1]}], {now.shape[1] * 2} B per row, as it is now: {int(ibad.sum())} bad, "
                              f"row ranges {ranges(ibad.any(1).nonzero().flatten())}", flush=True)
                    print(f"[{a.tag}]   conv {first} pre-BN [{got.shape[0]} x {got.shape[1]}]: {int(bad.sum())} bad; row ranges {ranges(rr)}; column ranges {ranges(cc)}; "
                          f"(row, col, got, want): {samples}", flush=True)
                firsts = next((i for i, v in enumerate(rels) if v > 0.02), None)
                print(f"[{a.tag}]   first conv whose pre-BN output deviates > 2e-2: {first}; first BN whose statistics deviate: {firsts}; "
                      f"pre-BN deviations conv 0..{min(43, (first or 0) + 6)}: {[round(v, 4) for v in rel[:(first or 0) + 6]]}; stats: {[round(v, 4) for v in rels[:(firsts or 0) + 4]]}", flush=True)
devs = np.array(devs)
print(f"[{a.tag}] workspace at {bb._ws.data_ptr():#x}", flush=True)
print(f"[{a.tag}] {a.size}^3 feature deviation over {a.iters} forwards: median {np.median(devs):.2e} p99 {np.quantile(devs, 0.99):.2e} max {devs.max():.2e}", flush=True)
