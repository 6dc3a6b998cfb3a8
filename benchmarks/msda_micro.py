"""Kernel micro-benchmarks at the 800x1333 benchmark shape (B=2): MSDA variants, backward, top-k,
head-major re-layout.  Times with events on the launch stream; prints one line per case."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as FO
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
DEV = "cuda:0"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def alg_bytes(B, Nv, Nq, M_, D, LP, sv, so=4):
    return B * (Nv * M_ * D * sv + Nq * M_ * LP * 3 * 4 + Nq * M_ * D * so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--nq", type=int, nargs="*", default=[11363, 9090, 6817, 4545, 2272, 900])
    args = ap.parse_args()
    B, M_, D, P, L = args.B, 8, 32, 4, 4
    res = []
    for Nq in args.nq:
        value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, LEVELS, M_, D, P, seed=0)
        Nv = value.shape[1]
        dv, sh, ls = value.to(DEV), shapes.to(DEV), lsi.to(DEV)
        for order_name in ("random", "spatial"):
            if order_name == "spatial":
                # same queries, processed in token (spatial) order: sort by reference location
                key = (loc[:, :, 0, 0, 0, 1] * 1000).round() * 10 + loc[:, :, 0, 0, 0, 0]
                perm = key.argsort(1)
                loc_o = torch.gather(loc, 1, perm[:, :, None, None, None, None].expand_as(loc)).contiguous()
                aw_o = torch.gather(aw, 1, perm[:, :, None, None, None].expand_as(aw)).contiguous()
            else:
                loc_o, aw_o = loc, aw
            dl, da = loc_o.to(DEV), aw_o.to(DEV)
            t = timeit(lambda: M.ms_deform_attn_forward(dv, sh, ls, dl, da, 64))
            res.append(dict(case="fwd_ref_layout_f32", Nq=Nq, order=order_name, us=t,
                            GBps=alg_bytes(B, Nv, Nq, M_, D, 16, 4) / t / 1e3))
            for vdt, sv in ((torch.float32, 4), (torch.bfloat16, 2)):
                hm = M.value_to_head_major(dv.view(B, Nv, M_ * D), None, M_, vdt)
                t = timeit(lambda: M.msda_forward_head_major(hm, sh, ls, dl, da))
                res.append(dict(case=f"fwd_head_major_{str(vdt)[6:]}", Nq=Nq, order=order_name, us=t,
                                GBps=alg_bytes(B, Nv, Nq, M_, D, 16, sv) / t / 1e3))
            if order_name == "random":
                go = torch.randn(B, Nq, M_ * D, device=DEV)
                t = timeit(lambda: M.ms_deform_attn_backward(dv, sh, ls, dl, da, go, 64), reps=5)
                res.append(dict(case="bwd_ref_layout_f32", Nq=Nq, order=order_name, us=t))
        # fused bf16 path with raw projections
        proj = torch.cat([torch.randn(B, Nq, 256) * 2.0, torch.randn(B, Nq, 128)], -1).to(DEV)
        refp = (loc[:, :, 0, :, 0, :]).contiguous().to(DEV)
        hm = M.value_to_head_major(dv.view(B, Nv, M_ * D), None, M_, torch.bfloat16)
        for pdt in (torch.float32, torch.bfloat16):
            pj = proj.to(pdt)
            t = timeit(lambda: M.msda_fused_forward(hm, sh, ls, refp, pj, L, P, out_dtype=pdt))
            res.append(dict(case=f"fused_bf16value_proj{str(pdt)[6:]}", Nq=Nq, order="random", us=t,
                            GBps=B * (Nv * 256 * 2 + Nq * (384 + 256) * pj.element_size() + Nq * 32) / t / 1e3))
    # encoder-like dense query sets: direct fused kernel vs LDS-staged kernel
    for Nq in args.nq:
        for off in (1.5, 3.0):
            tok, ref, proj, shapes, lsi = syn.make_encoder_like_queries(B, Nq, LEVELS, 8, 4, seed=1, offset_px=off)
            hm = M.value_to_head_major(torch.randn(B, 22323, 256, device=DEV), None, 8, torch.bfloat16)
            sh, ls, rf, pj = shapes.to(DEV), lsi.to(DEV), ref.to(DEV), proj.to(torch.bfloat16).to(DEV)
            alg = B * (22323 * 256 * 2 + Nq * 384 * 2 + Nq * 32 + Nq * 256 * 2)
            t = timeit(lambda: M.msda_fused_forward(hm, sh, ls, rf, pj, 4, 4, out_dtype=torch.bfloat16))
            res.append(dict(case="enc_direct_bf16", Nq=Nq, offset_px=off, us=t, GBps=alg / t / 1e3))
            hm16 = hm.float().to(torch.float16)
            t = timeit(lambda: M.msda_fused_forward(hm16, sh, ls, rf, pj, 4, 4, out_dtype=torch.bfloat16))
            res.append(dict(case="enc_direct_f16value", Nq=Nq, offset_px=off, us=t, GBps=alg / t / 1e3))
    value = torch.randn(B, 22323, 256, device=DEV)
    for sdt in (torch.float32, torch.bfloat16):
        for ddt in (torch.float32, torch.bfloat16):
            v = value.to(sdt)
            t = timeit(lambda: M.value_to_head_major(v, None, 8, ddt))
            res.append(dict(case=f"to_head_major_{str(sdt)[6:]}_to_{str(ddt)[6:]}", us=t,
                            GBps=B * 22323 * 256 * (v.element_size() + (4 if ddt == torch.float32 else 2)) / t / 1e3))
    for N, k in ((16800, 6680), (4200, 3360), (1050, 1050), (273, 273), (11363, 11363), (11363, 300), (22323, 3600)):
        s = torch.randn(B, N, device=DEV)
        t = timeit(lambda: FO.masked_topk_desc(s, k))
        t2 = timeit(lambda: torch.topk(s, k, dim=1))
        res.append(dict(case="topk", N=N, k=k, us=t, torch_topk_us=t2))
    q = torch.randn(B, 22323, 256, device=DEV)
    idx = torch.stack([torch.randperm(22323)[:11363] for _ in range(B)]).to(DEV)
    t = timeit(lambda: FO.gather_rows(q, idx))
    e = idx[..., None].expand(-1, -1, 256)
    t2 = timeit(lambda: torch.gather(q, 1, e))
    res.append(dict(case="gather_rows_11363x256_f32", us=t, torch_gather_us=t2, GBps=B * 11363 * 2048 / t / 1e3))
    for r in res:
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()}))


if __name__ == "__main__":
    main()
