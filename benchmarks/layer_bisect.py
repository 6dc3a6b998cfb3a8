"""Where does a 16-bit encoder layer lose its accuracy?  One teacher-forced encoder layer at the benchmark shape (inputs and
top-300 set of the CPU oracle's fp32 run), in bf16 and in fp16 activations, with the intermediate tensors of the HIP path
held against the oracle's fp32 values step by step:

  slab     the offsets | logits projection the MSDA launch reads (after the top-300 attention updated its rows)
  sampled  the MSDA launch's output (before output_proj)
  out      the layer's output

    python benchmarks/layer_bisect.py [--layer 0] [--out gpurun_out/layer_bisect.json]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import salience_ref as R  # noqa: E402  (checker only)
from salience_detr_amd import ms_deform_attn as M  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402
from salience_detr_amd.hot_path import build_hot_path  # noqa: E402

DEV = "cuda:0"


def stats(got, want):
    d = (got.float().cpu() - want.float().cpu()).abs()
    return {"mean": round(d.mean().item(), 6), "p999": round(d.flatten().kthvalue(max(1, int(d.numel() * 0.999)))[0].item(), 5),
            "max": round(d.max().item(), 5), "scale": round(want.float().abs().mean().item(), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/layer_bisect.json")
    args = ap.parse_args()
    k = args.layer
    sizes = [(800, 1333), (800, 1333)]
    m = build_hot_path()
    m.load_state_dict(syn.det_state_dict(m.state_dict()))
    _, masks = syn.make_masks(sizes)
    shapes = [tuple(x.shape[-2:]) for x in masks]
    feats = syn.make_feats(len(sizes), shapes, 256, seed=0)
    pos = [syn.sine_position_embedding(x, 128) for x in masks]
    sd = {kk: v.detach().float().clone() for kk, v in m.state_dict().items()}
    with torch.no_grad():
        ref = R.hot_path(sd, feats, masks, pos)
    lin, sel = ref["layer_in"][k], ref["layer_sel"][k]
    prefix = f"encoder.layers.{k}"
    # the oracle's intermediate values (fp32, CPU)
    with torch.no_grad():
        q, qp = lin["query"], lin["query_pos"]
        E = q.shape[-1]
        sel_e = sel.unsqueeze(-1).expand(-1, -1, E)
        tgt, p = torch.gather(q, 1, sel_e), torch.gather(qp, 1, sel_e)
        tgt = R.layer_norm(sd, prefix + ".pre_norm", tgt + R.mha_self(sd, prefix + ".pre_attention", tgt + p, tgt, 8))
        q_att = q.scatter(1, sel_e, tgt)
        x = q_att + qp
        off = R.linear(sd, prefix + ".self_attn.sampling_offsets", x)
        lgt = R.linear(sd, prefix + ".self_attn.attention_weights", x)
        B, Nq, _ = x.shape
        want_slab = torch.cat([off.view(B, Nq, 8, 32), lgt.view(B, Nq, 8, 16)], -1).permute(0, 2, 1, 3)
        value = ref["feat_flatten"]
        v = R.linear(sd, prefix + ".self_attn.value_proj", value).masked_fill(ref["mask_flatten"][..., None], 0.0)
        aw = lgt.view(B, Nq, 8, 16).softmax(-1).view(B, Nq, 8, 4, 4)
        loc = R.sampling_locations(lin["ref"], off.view(B, Nq, 8, 4, 4, 2), ref["spatial_shapes"], 4)
        want_sampled = R.msda_core_c(v.view(B, -1, 8, 32).contiguous(), ref["spatial_shapes"], ref["level_start_index"],
                                     loc.contiguous(), aw.contiguous())
        want_out = ref["layer_out"][k]
    rows = []
    for name, act in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        # (a fresh module per type: .to(bfloat16) rounds the parameters in place)
        m = build_hot_path()
        m.load_state_dict(syn.det_state_dict(m.state_dict()))
        m = m.to(DEV).eval()
        m.set_encoder_dtype(act, torch.float16)
        enc = m.encoder
        layer = enc.layers[k]
        grabbed = {}
        real = M.msda_bordered_forward

        def grab(value_hm, levels, refp, proj, row_order=None, out_dtype=None, chunks=0):
            out = real(value_hm, levels, refp, proj, row_order=row_order, out_dtype=out_dtype, chunks=chunks)
            grabbed["slab"], grabbed["sampled"] = proj.clone(), out.clone()
            return out
        M.msda_bordered_forward = grab
        try:
            with torch.no_grad():
                maps = enc.project_values(ref["feat_flatten"].to(DEV).to(act), ref["mask_flatten"].to(DEV), shapes)
                qd = lin["query"].to(DEV).to(act).contiguous()
                q_in = qd.clone()
                out = layer.forward_sorted(qd, lin["query_pos"].to(DEV).to(act).contiguous(), lin["ref"].to(DEV).contiguous(),
                                           lin["fg"].to(DEV).contiguous(), maps[k], ref["spatial_shapes"].to(DEV),
                                           ref["level_start_index"].to(DEV), enc.enhance_mcsp, level_shapes=shapes,
                                           selection_hook=lambda s, forced=sel.to(DEV): forced)
        finally:
            M.msda_bordered_forward = real
        rec = {"act": name, "layer": k, "rows": Nq,
               "input_rounding": stats(q_in, lin["query"]),
               "slab": stats(grabbed["slab"], want_slab), "sampled": stats(grabbed["sampled"], want_sampled),
               "out": stats(out, want_out)}
        # the MSDA launch alone on the oracle's own operands rounded to the activation type (what the sampling itself adds)
        with torch.no_grad():
            alone = real(maps[k], shapes, lin["ref"].to(DEV).contiguous(), want_slab.to(act).contiguous().to(DEV), out_dtype=torch.float32)
        rec["sampled_from_oracle_slab"] = stats(alone, want_sampled)
        rows.append(rec)
        print(json.dumps(rec))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
