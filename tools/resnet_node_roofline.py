#!/usr/bin/env python3
"""Per-node roofline table of the ResNet3D forward + backward graphs: every launch of the two captured sequences with
its algorithmic FLOPs / designed bytes (the library's launch manifest, neraf_manifest_*), its duration (rocprofv3 kernel trace of the
same iterations) and how far it is from its own roofline -- max(bytes / 8 TB/s, FLOPs / 2.5 PF) / duration.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rn -- python3 $R/tools/resnet_node_roofline.py run 12
    python3 tools/resnet_node_roofline.py table /tmp/rn 12 > profiles/<tag>_resnet_node_roofline.txt

`run`: N iterations of the encoder alone on the 7 x 128^3 grid with a 4096-cell refresh window (the training step's call), then ONE
more iteration with the manifest enabled (un-graphed), written to /tmp/resnet_manifest.json.  `table`: zips the manifest with the
steady-state iterations of the trace by kernel-name prefix, in launch order."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM, MFMA = 8.0e12, 2.5e15


def run(n):
    import ctypes as C
    import numpy as np, torch
    from neraf_amd import _lib, synth
    from neraf_amd.resnet3d import ResNet3D_helper
    dev = torch.device("cuda:0")
    net = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=1 / 128, N_features=1024)
    net.backbone_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.resnet3d_state_dict(7).items()})
    net.to(dev).train()
    x = torch.from_numpy(synth.uniform("g1.grid128", (1, 7, 128, 128, 128), 0.0, 1.0)).to(dev)
    w = torch.ones(1024, device=dev)
    bb = net.backbone_net
    bb.grid_window = (0, 4096, 4)
    bb.grid_grad_sink = lambda d: None
    for _ in range(n):
        (net(x).flatten() * w).sum().backward()
    torch.cuda.synchronize()
    lib, h = _lib.load(), _lib.ctx(0)
    lib.neraf_manifest_enable(h, 1)
    (net(x).flatten() * w).sum().backward()
    torch.cuda.synchronize()
    nodes, name = [], C.create_string_buffer(160)
    fl, rb, wb = C.c_double(), C.c_double(), C.c_double()
    cnt = lib.neraf_manifest_get(h, -1, None, 0, None, None, None)
    for i in range(cnt):
        lib.neraf_manifest_get(h, i, name, 160, C.byref(fl), C.byref(rb), C.byref(wb))
        nodes.append({"name": name.value.decode(), "flops": fl.value, "rbytes": rb.value, "wbytes": wb.value})
    lib.neraf_manifest_enable(h, 0)
    json.dump(nodes, open("/tmp/resnet_manifest.json", "w"))
    print("done", n, "manifest nodes", len(nodes))


def table(trace_dir, n_iter):
    nodes = json.load(open("/tmp/resnet_manifest.json"))
    f = sorted(glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # the manifest iteration is the LAST one in the trace (un-graphed: same kernels, same order); steady-state iterations before it are
    # found by walking backwards with the manifest's prefixes
    prefixes = [nd["name"].split(" | ")[0] for nd in nodes]

    def match_from(end):
        """indices of the trace rows matching the manifest, scanning backwards from `end` (exclusive); None if it does not fit"""
        idx, j = [], end - 1
        for p in reversed(prefixes):
            while j >= 0 and p not in names[j]:
                j -= 1
            if j < 0:
                return None
            idx.append(j); j -= 1
        return idx[::-1]

    its, end = [], len(rows)
    for _ in range(min(n_iter, 6) + 1):
        m = match_from(end)
        if m is None:
            break
        its.append(m); end = m[0]
    its = its[1:5]                                   # drop the manifest iteration itself (direct launches), keep 4 replayed ones
    if not its:
        sys.exit("could not align the manifest with the trace")
    dur = [sum(int(rows[m[i]]["End_Timestamp"]) - int(rows[m[i]]["Start_Timestamp"]) for m in its) / len(its) / 1e3 for i in range(len(nodes))]
    print(f"# ResNet3D forward + backward on the 7 x 128^3 grid, refresh window 4096 cells: {len(nodes)} launches per iteration, "
          f"{sum(dur):.1f} us of kernel time (mean of {len(its)} replayed iterations)")
    print("# frac = max(bytes / 8 TB/s, FLOPs / 2.5 PFLOP/s) / duration: how close the launch runs to its OWN roofline (bytes as designed:")
    print("# operands once, results once, K-split slabs where written and read; FLOPs algorithmic, SURVEY 8d)")
    print(f"{'#':>4} {'us':>8} {'MB':>8} {'GFLOP':>8} {'TB/s':>6} {'TF/s':>7} {'bound':>5} {'frac':>6}  kernel | what")
    tot = collections.defaultdict(float)
    worst = []
    for i, (nd, d) in enumerate(zip(nodes, dur)):
        by, fl = nd["rbytes"] + nd["wbytes"], nd["flops"]
        t_b, t_f = by / HBM * 1e6, fl / MFMA * 1e6
        bound = "mfma" if t_f > t_b else "hbm"
        frac = max(t_b, t_f) / d if d > 0 else 0.0
        print(f"{i:4d} {d:8.2f} {by / 1e6:8.2f} {fl / 1e9:8.3f} {by / d / 1e6 if d else 0:6.2f} {fl / d / 1e6 if d else 0:7.1f} {bound:>5} {frac:6.3f}  {nd['name']}")
        k = nd["name"].split(" | ")[0].split("<")[0]
        tot[k] += d
        if bound == "hbm" and by > 16e6:
            worst.append((frac, d, i, nd["name"]))
    print("\n# time by kernel (us per iteration)")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"#   {v:8.1f}  {k}")
    print("\n# streaming launches (> 16 MB) furthest from the HBM roofline")
    for frac, d, i, nm in sorted(worst)[:12]:
        print(f"#   node {i:3d}  frac {frac:5.3f}  {d:7.2f} us  {nm}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    else:
        table(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12)
