#!/usr/bin/env python3
"""VERDICT r3 #8, the gather side of the training field query: what would ANY sort of the render batch's samples buy?

The fine samples of a training ray batch (4096 random pixels of random cameras x 48 samples, taken from the real sampler chain of the
bench workload) are handed to `neraf_field_query` as R*S single-sample rays in three orders:

  ray     the order the training kernel sees: consecutive lanes = consecutive samples of ONE ray
  morton  globally sorted by the Morton code of the level-8 cell of the contracted position (the upper bound of every
          within-workgroup or within-batch sort: the whole batch is one sorted run)
  random  a random permutation (what the gather costs with no locality at all)

Only the ORDER differs: same samples, same kernel, same results up to the permutation (checked).  Reported: kernel time per launch
(torch events around 50 launches; the library launches on torch's current stream), and what a sort would cost on top -- torch.sort of
the 196,608 keys plus the gathers of the permuted operands, as a stand-in for a hand-written radix pass.  With --order X --pmc the
script only launches that order 20 times (for `rocprofv3 --pmc FETCH_SIZE`, tools/gpu_r4_sort.sh)."""
import argparse, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def contract01(x):
    """nerfacto's L-inf scene contraction followed by the map of [-2, 2] onto [0, 1] (oracle/vision.py restates the same)."""
    n = x.abs().amax(-1, keepdim=True).clamp_min(1e-12)
    y = torch.where(n <= 1, x, (2 - 1 / n) * (x / n))
    return ((y + 2) / 4).clamp(0, 1)


def morton_keys(x01, res):
    c = (x01 * res).long().clamp(0, res - 1)
    def spread(v):
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", default="all", choices=["all", "ray", "morton", "random"])
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    js = bench.JointStep(dev, 4096, 2048, 1, rotate=1, camera_opt=False)
    vm = js.vm
    vm.train()
    vm.update_to_step(js.i)        # the bench's operating point (step 20000: proposal-weight annealing finished)
    with torch.no_grad():
        out = vm.get_outputs(js.bundle)
    fine = out["ray_samples_list"][-1]
    e = fine.e_bins if hasattr(fine, "e_bins") else fine.frustums.ends
    o = js.bundle.origins.float(); d = js.bundle.directions.float()
    R, S = e.shape[0], e.shape[1] - 1
    mid = 0.5 * (e[:, :-1] + e[:, 1:])                                     # [R, S]
    pos = (o[:, None, :] + d[:, None, :] * mid[..., None]).reshape(-1, 3).contiguous()
    dirs = d[:, None, :].expand(R, S, 3).reshape(-1, 3).contiguous()
    cam = js.bundle.camera_indices.reshape(-1, 1).expand(R, S).reshape(-1).to(torch.int32).contiguous()
    N = pos.shape[0]
    zero_e = torch.zeros((N, 2), dtype=torch.float32, device=dev)          # zero-length frustums: the sample IS the origin
    field = vm.field.module
    packed = field.packed(with_average=False)
    # level-8 resolution of the 16-level nerfacto grid (16 ... 2048): floor(16 * growth^8)
    import math
    res8 = int(math.floor(16 * math.exp(math.log(2048 / 16) / 15 * 8)))
    keys = morton_keys(contract01(pos), res8)
    perms = {"ray": torch.arange(N, device=dev), "morton": torch.argsort(keys), "random": torch.randperm(N, device=dev)}

    def run(order, reps):
        p = perms[order]
        po, di, ca = pos[p].contiguous(), dirs[p].contiguous(), cam[p].contiguous()
        rgb, den = field.query(po, di, zero_e, ca, packed=packed)          # warm-up + result
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            field.query(po, di, zero_e, ca, packed=packed)
        t1.record(); torch.cuda.synchronize()
        return t0.elapsed_time(t1) * 1e3 / reps, (rgb, den, p)

    if a.pmc:
        us, _ = run(a.order, 20)
        print(json.dumps({"order": a.order, "us_per_launch_under_pmc": round(us, 1)}))
        return
    res = {"samples": N, "rays": R, "samples_per_ray": S, "level8_res": res8}
    base = None
    for order in ("ray", "morton", "random"):
        us, (rgb, den, p) = run(order, a.reps)
        back_rgb = torch.empty_like(rgb); back_rgb[p] = rgb
        back_den = torch.empty_like(den); back_den[p] = den
        if base is None:
            base = (back_rgb, back_den)
        res[order] = {"us_per_launch": round(us, 1),
                      "max_abs_diff_vs_ray_order": float(max((back_rgb - base[0]).abs().max(), ((back_den - base[1]).abs() / (base[1].abs() + 1)).max()))}
    # the structured training-shaped call for reference (R rays x S samples, what the step launches; inference form)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    field.query(o.contiguous(), d.contiguous(), e.contiguous(), js.bundle.camera_indices_i32(), packed=packed)
    t0.record()
    for _ in range(a.reps):
        field.query(o.contiguous(), d.contiguous(), e.contiguous(), js.bundle.camera_indices_i32(), packed=packed)
    t1.record(); torch.cuda.synchronize()
    res["structured_R_x_S_call_us"] = round(t0.elapsed_time(t1) * 1e3 / a.reps, 1)
    # what a sort costs: keys + argsort + the permuted operands (a hand-written pass would fuse key generation into the sampler and
    # use one 196,608-key radix sort; this is the generous torch stand-in, reported as such)
    t0.record()
    for _ in range(a.reps):
        k = morton_keys(contract01(pos), res8)
        p = torch.argsort(k)
        _ = pos[p], dirs[p], cam[p]
    t1.record(); torch.cuda.synchronize()
    res["torch_keys_argsort_gather_us"] = round(t0.elapsed_time(t1) * 1e3 / a.reps, 1)
    t0.record()
    for _ in range(a.reps):
        p = torch.argsort(keys)
    t1.record(); torch.cuda.synchronize()
    res["torch_argsort_only_us"] = round(t0.elapsed_time(t1) * 1e3 / a.reps, 1)
    # distinct level-8 cells and lines touched, per 64 consecutive samples (one wavefront's worth), by order
    for order in ("ray", "morton", "random"):
        k = keys[perms[order]][: (N // 64) * 64].reshape(-1, 64)
        srt = k.sort(dim=1).values
        res[order]["distinct_level8_cells_per_64_samples"] = round(float((1 + (srt[:, 1:] != srt[:, :-1]).sum(1)).float().mean()), 2)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
