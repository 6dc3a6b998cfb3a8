#!/usr/bin/env python
"""Benchmark of the Salience-DETR encoder hot path on MI355X (contract: see task / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path (F0 pyramid plumbing -> F1-F3 hierarchical salience filtering ->
6-layer salience encoder, i.e. reference SalienceTransformer.forward lines 97-183) over ONE batch of
synthetic 800x1333 4-level pyramids resident in HBM, batch-per-GPU = 2, bf16 encoder
(BASELINE.json configs[1]).  Images are independent, so N GPUs run N replicas of the path on their own
batches with no data-path collective (weak scaling); the timed region is bracketed by
barrier + synchronize on both sides and the MAX over ranks is reported.

Rank 0 prints ONE JSON line: whole-job images/s, ms per encoder layer, the `roofline` object of the
dominant kernel (the fused MSDA gather, measured live with stream events around every launch of an
instrumented pass) and, at N=1, the `cpu_baseline` (the oracle's CPU port of the same path, timed on the
host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from salience_detr_amd import ms_deform_attn as msda_mod  # noqa: E402
from salience_detr_amd import pyramid, synthetic as syn  # noqa: E402
from salience_detr_amd.hot_path import build_hot_path  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MSDA_REPEATS = 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer (default, BASELINE.json configs[1]) or train: fp32 forward+backward of the hot path, "
                         "flat RCCL gradient all-reduce, AdamW step (configs[2])")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--value-dtype", choices=["same", "fp16"], default="fp16",
                    help="storage type of the head-major value maps sampled by the MSDA kernel in bf16 mode: fp16 "
                         "(default; 11-bit mantissa, gather via v_fma_mix_f32) or the activation dtype (bf16)")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--instrumented-steps", type=int, default=10)
    return ap.parse_args()


def make_inputs(batch, h, w, device, seed):
    sizes = [(h, w)] * batch
    canvas = syn.pad_to_32(h, w)
    _, masks = syn.make_masks(sizes)
    shapes = pyramid.level_shapes_of(masks)
    feats = syn.make_feats(batch, shapes, 256, seed=seed)
    pe = pyramid.PositionEmbeddingSine(128, temperature=10000, normalize=True, offset=-0.5)
    pos = [pe(m) for m in masks]
    cpu = (feats, masks, pos)
    dev = tuple([t.to(device) for t in ts] for ts in cpu)
    return sizes, canvas, shapes, cpu, dev


def algorithmic_bytes(B, Nv, Nq, M, D, L, P, value_bytes, proj_bytes, out_bytes, ref_dim=2):
    """Each input and output of the fused MSDA launch touched exactly once (SURVEY.md 8(d) formula with the
    fused kernel's actual operands: 3 projection values per sample instead of loc(2)+weight(1), plus the
    fp32 reference points)."""
    return B * (Nv * M * D * value_bytes + Nq * M * L * P * 3 * proj_bytes + Nq * L * ref_dim * 4
                + Nq * M * D * out_bytes)


def train_main(args, model, device, rank, world, dist):
    """configs[2]: one training step of the hot-path modules per batch of 2 images per GPU -- fp32 forward
    through the autograd path (HIP MSDA forward/backward op), the salience criterion (row N4: targets + focal loss
    on the salience maps, synthetic ground-truth boxes) plus a synthetic loss on `memory`, backward, ONE flat
    all-reduce of the ~38 MB of gradients over RCCL, AdamW."""
    from salience_detr_amd.data_parallel import FlatGradAllReducer, broadcast_parameters
    from salience_detr_amd.salience_criterion import SalienceCriterion
    sizes, canvas, level_shapes, _, (feats, masks, pos) = make_inputs(args.batch, args.height, args.width, device,
                                                                      seed=rank)
    model.train()
    if dist is not None:
        broadcast_parameters(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4)
    reducer = FlatGradAllReducer(params) if dist is not None else None
    w = None
    criterion = SalienceCriterion()
    strides = [(canvas[0] / h, canvas[1] / w_) for h, w_ in level_shapes]
    targets = []
    for i in range(args.batch):   # 12 deterministic boxes per image over all four scale ranges
        c = syn.det_rand(f"bench.box.c{i}", (12, 2), salt=rank) * 0.8 + 0.1
        wh = 0.02 + syn.det_rand(f"bench.box.wh{i}", (12, 2), salt=rank) ** 2 * 0.9
        targets.append({"boxes": torch.cat([c, wh], -1).to(device)})

    def step():
        nonlocal w
        opt.zero_grad(set_to_none=True)
        memory, score_maps = model(feats, masks, pos, image_sizes=sizes, canvas=canvas)
        if w is None:
            w = torch.randn_like(memory)
        loss = (memory * w).mean() + criterion(score_maps, targets, strides, sizes)["loss_salience"]
        loss.backward()
        if reducer is not None:
            reducer.all_reduce(average=True)
        opt.step()
        return loss

    for _ in range(max(args.warmup, 2)):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # dominant kernel of the training step: the MSDA backward scatter; timed with stream events
    evs, nbytes = [], []
    real_bwd = msda_mod.ms_deform_attn_backward

    def timed_bwd(value, shapes, lsi, loc, aw, grad_out, step_):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_bwd(value, shapes, lsi, loc, aw, grad_out, step_)
        e1.record()
        evs.append((e0, e1))
        B, Nv, M, D = value.shape
        Nq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
        nbytes.append(4 * B * (2 * Nv * M * D + 2 * Nq * M * L * P * 3 + Nq * M * D))  # SURVEY.md 8(d) backward
        return r

    msda_mod.ms_deform_attn_backward = timed_bwd
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    msda_mod.ms_deform_attn_backward = real_bwd
    tot_us = sum(e0.elapsed_time(e1) for e0, e1 in evs) * 1e3
    achieved = sum(nbytes) / tot_us / 1e3
    result = {
        "metric": "images/s (whole node) + ms/encoder-layer, ResNet50 800x1333",
        "value": round(world * args.batch * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
        "config": {"workload": "salience_detr_resnet50_800_1333 training step of the hot path (filtering + 6-layer "
                               "encoder fwd+bwd, salience focal loss + synthetic memory loss, AdamW), batch=%d per MI355X" % args.batch,
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "parallelism": "data parallel, one flat gradient all-reduce per step over RCCL"
                                  if world > 1 else "single GPU",
                   "grad_bytes": reducer.num_bytes if reducer is not None else sum(p.numel() * 4 for p in params)},
        "roofline": {"kernel": "sdetr::msda_col2im_kernel (MSDA backward scatter, fp32 atomics)", "bound": "hbm",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "avg_launch_us": round(tot_us / max(1, len(evs)), 1)},
        "loss": float(loss),
    }
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    model = build_hot_path()
    model.load_state_dict(syn.det_state_dict(model.state_dict()))
    model = model.to(device).eval()
    if args.mode == "train":
        return train_main(args, model, device, rank, world, dist)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.set_encoder_dtype(dtype, torch.float16 if (args.dtype == "bf16" and args.value_dtype == "fp16") else None)

    sizes, canvas, level_shapes, cpu_inputs, (feats, masks, pos) = make_inputs(
        args.batch, args.height, args.width, device, seed=rank)

    def step():
        with torch.no_grad():
            return model(feats, masks, pos, image_sizes=sizes, canvas=canvas)[0]

    for _ in range(max(args.warmup, 3)):
        out = step()
    torch.cuda.synchronize()

    graphed = False
    run = step
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # N > 1: the process group's helper threads exist by now; thread-local capture mode keeps anything they
            # might call from invalidating this thread's capture (no collective is captured: the data path has none)
            capture_kw = {"capture_error_mode": "thread_local"} if world > 1 else {}
            with torch.cuda.graph(g, **capture_kw):
                out = step()
            g.replay()
            torch.cuda.synchronize()
            run = g.replay
            graphed = True
        except Exception as e:  # keep the eager path measurable if capture is unavailable
            sys.stderr.write(f"[bench] hipGraph capture failed, timing eager launches: {e}\n")
            torch.cuda.synchronize()
    for _ in range(3):
        run()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    images_per_s = world * args.batch * args.steps / elapsed

    # ---- instrumented eager pass: events around every fused-MSDA launch and every layer boundary ----
    msda_events, layer_events, launches, launch_nq = [], [], [], []
    real_fused = msda_mod.msda_fused_forward

    kernels_used = []

    def record(real_call, value_hm, reference_points, proj, head_major, o, kernel):
        # the step's own launch is done (o); the SAME launch (same operands, straight after its producers) is then
        # repeated back to back between two events on the launch stream, so the measured time is kernel time
        # (what rocprofv3 --kernel-trace reports), not host launch gaps of the eager instrumented pass
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(MSDA_REPEATS):
            real_call()
        e1.record()
        msda_events.append((e0, e1))
        B, M, Nv, D = value_hm.shape
        nq = int(proj.shape[2] if head_major else proj.shape[1])
        launch_nq.append(nq)
        kernels_used.append(kernel)
        launches.append(algorithmic_bytes(B, Nv, nq, M, D, 4, 4, value_hm.element_size(), proj.element_size(),
                                          o.element_size(), reference_points.shape[-1]))

    def timed_fused(value_hm, spatial_shapes, level_start_index, reference_points, proj, num_levels, num_points,
                    order=None, out_dtype=None, proj_head_major=False):
        call = lambda: real_fused(value_hm, spatial_shapes, level_start_index, reference_points, proj, num_levels,
                                  num_points, order=order, out_dtype=out_dtype, proj_head_major=proj_head_major)
        o = call()
        record(call, value_hm, reference_points, proj, proj_head_major, o, "msda_gather_l4p4_kernel<half_t>")
        return o

    real_resident = msda_mod.msda_resident_forward

    def timed_resident(value_hm, level_shapes_, reference_points, proj_hm, out_dtype=None, chunks=0):
        call = lambda: real_resident(value_hm, level_shapes_, reference_points, proj_hm, out_dtype=out_dtype, chunks=chunks)
        o = call()
        record(call, value_hm, reference_points, proj_hm, True, o, "msda_resident_kernel<half_t>")
        return o

    def marker(layer_id):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        layer_events.append((layer_id, e))

    msda_mod.msda_fused_forward = timed_fused
    msda_mod.msda_resident_forward = timed_resident
    model.encoder.layer_marker = marker
    for _ in range(args.instrumented_steps):
        step()
    torch.cuda.synchronize()
    msda_mod.msda_fused_forward = real_fused
    msda_mod.msda_resident_forward = real_resident
    model.encoder.layer_marker = None

    nl = model.encoder.num_layers
    msda_us = [0.0] * nl
    for i, (e0, e1) in enumerate(msda_events):
        msda_us[i % nl] += e0.elapsed_time(e1) * 1e3 / args.instrumented_steps / MSDA_REPEATS
    layer_ms = [0.0] * nl
    for (l0, e0), (l1, e1) in zip(layer_events[:-1], layer_events[1:]):
        if l1 == l0 + 1:
            layer_ms[l0] += e0.elapsed_time(e1) / args.instrumented_steps
    bytes_per_layer = launches[:nl]
    total_bytes, total_us = sum(bytes_per_layer), sum(msda_us)
    achieved = total_bytes / total_us / 1e3  # GB/s
    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same workload (bench.py cannot
    # run the profiler on itself); null when the workload differs from the profiled one
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_msda_traffic.json")))
        nqs = launch_nq[:nl]
        if (args.dtype == "bf16" and args.value_dtype == tj.get("value_dtype", "same") and args.batch == tj["batch"]
                and all(str(n) in tj["per_num_query"] for n in nqs)):
            traffic = int(sum(tj["per_num_query"][str(n)]["hbm_bytes"] for n in nqs) / nl)
            traffic_src = "profiles/r01_msda_traffic.json"
    except (OSError, ValueError, KeyError):
        pass
    roofline = {
        "kernel": "sdetr::" + " / ".join(sorted(set(kernels_used[:nl])))
                  + " (fused softmax + sampling locations + bilinear gather)",
        "kernel_per_layer": kernels_used[:nl],
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
        "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(total_bytes / nl),
        "launches_per_step": nl, "num_queries_per_layer": launch_nq[:nl], "avg_launch_us": round(total_us / nl, 2),
        "per_layer_us": [round(u, 2) for u in msda_us],
        "per_layer_algorithmic_MB": [round(b / 1e6, 2) for b in bytes_per_layer],
    }

    result = {
        "metric": "images/s (whole node) + ms/encoder-layer, ResNet50 800x1333",
        "value": round(images_per_s, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "salience_detr_resnet50_800_1333 %s inference, batch=%d per MI355X: pyramid flatten + "
                               "hierarchical salience filtering + 6-layer salience encoder (MSDA)" % (args.dtype, args.batch),
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "image": [args.height, args.width], "levels": [list(s) for s in level_shapes],
                   "value_map_storage": ("fp16" if (args.dtype == "bf16" and args.value_dtype == "fp16") else args.dtype),
                   "parallelism": "replicas, images sharded across GPUs, no data-path collective",
                   "hipgraph": graphed},
        "ms_per_encoder_layer": {"mean": round(sum(layer_ms) / nl, 4), "per_layer": [round(x, 4) for x in layer_ms],
                                 "note": "eager instrumented pass (stream events at layer boundaries)"},
        "roofline": roofline,
    }

    # ---- CPU baseline: the oracle's port of the same path on the host cores (rank 0, N=1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import salience_ref as R  # checker / baseline only; never on the product path
        from oracle import msda_c
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        cf, cm, cp = cpu_inputs
        cores = torch.get_num_threads()
        times = []
        t_start = time.perf_counter()
        with torch.no_grad():
            R.hot_path(sd, cf, cm, cp)  # warm-up
            while len(times) < 5 and time.perf_counter() - t_start < 20.0:
                t1 = time.perf_counter()
                ref = R.hot_path(sd, cf, cm, cp)
                times.append(time.perf_counter() - t1)
        times.sort()
        med = times[len(times) // 2]
        err = (out.float().cpu() - ref["memory"]).abs()
        result["cpu_baseline"] = {
            "value": round(args.batch / med, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d timed passes of the same batch-%d 800x1333 hot path (oracle/salience_ref.py, fp32, "
                      "torch CPU ops + OpenMP C gather on %d threads), median" % (len(times), args.batch,
                                                                                  msda_c.num_threads()),
            "ms_per_pass": round(med * 1e3, 1),
        }
        result["parity_vs_cpu"] = {"max_abs": round(float(err.max()), 5), "mean_abs": round(float(err.mean()), 6),
                                   "note": "GPU %s output vs fp32 CPU oracle on the same batch" % args.dtype}

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
