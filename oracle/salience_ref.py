"""oracle/salience_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (PyTorch fp32/fp64) restatement of the Salience-DETR encoder hot path as
plain functions over a ``state_dict`` (same key names as the reference's
``SalienceTransformer``).  Each function cites the reference lines it follows.
It is the parity checker for the HIP product path and the timed ``cpu_baseline``
("port") of bench.py; the product package never imports it.

Pinned against tests/golden/*.npz, which were produced by running the IMPORTED
reference (tests/golden/make_golden.py): tests/test_oracle_golden.py asserts
this file reproduces every captured intermediate (score maps, token budgets,
selected indices, per-layer outputs, final memory) to ~1e-5.

Tie rule (the reference leaves it to torch.topk/torch.sort, unspecified):
equal scores are ordered lower-index-first (stable descending sort).
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import msda_c

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- F0 helpers
def flatten_levels(xs: Sequence[torch.Tensor]) -> torch.Tensor:
    """models/bricks/base_transformer.py:22-27 -- [B,(C),H,W] x L -> [B,S,(C)]."""
    out = torch.cat([x.flatten(-2) for x in xs], -1)
    return out.transpose(1, 2).contiguous() if out.ndim == 3 else out


def level_pos_embed(sd: SD, pos: Sequence[torch.Tensor]) -> torch.Tensor:
    """base_transformer.py:29-33."""
    le = sd["level_embeds"]
    return flatten_levels([p + le[l].view(1, -1, 1, 1) for l, p in enumerate(pos)])


def valid_ratio(mask: torch.Tensor) -> torch.Tensor:
    """base_transformer.py:48-56 -- (w, h) fraction of valid pixels along row 0 / column 0."""
    _, h, w = mask.shape
    vh = (~mask[:, :, 0]).sum(1).float() / h
    vw = (~mask[:, 0, :]).sum(1).float() / w
    return torch.stack([vw, vh], -1)


def level_misc(masks: Sequence[torch.Tensor]):
    """base_transformer.py:35-46."""
    shapes = torch.tensor([tuple(m.shape[-2:]) for m in masks], dtype=torch.int64)
    sizes = shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    vr = torch.stack([valid_ratio(m) for m in masks], 1)
    return shapes, lsi, vr


def sine_position_embedding(mask: torch.Tensor, num_pos_feats: int, temperature: float = 10000.0,
                            scale: float = 2 * math.pi, eps: float = 1e-6, offset: float = -0.5):
    """models/bricks/position_encoding.py:48-67 with normalize=True (config: resnet50_800_1333.py:32)."""
    not_mask = (~mask).to(torch.float32)
    y = not_mask.cumsum(1)
    x = not_mask.cumsum(2)
    y = (y + offset) / (y[:, -1:, :] + eps) * scale
    x = (x + offset) / (x[:, :, -1:] + eps) * scale
    i = torch.arange(num_pos_feats)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px = x[..., None] / dim_t
    py = y[..., None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2).contiguous()


def linear(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def layer_norm(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def backbone_output_memory(sd: SD, memory: torch.Tensor, mask_flat: torch.Tensor, shapes: torch.Tensor):
    """base_transformer.py:74-112, first return value only (salience_transformer.py:112-114).

    A token survives iff it is not padding and its proposal (cx, cy, w, h) lies in (0.01, 0.99).
    """
    n = memory.shape[0]
    valid = []
    cur = 0
    for lvl, (h, w) in enumerate(shapes.tolist()):
        m = mask_flat[:, cur:cur + h * w].view(n, h, w)
        vh = (~m[:, :, 0]).sum(1).view(n, 1, 1).float()
        vw = (~m[:, 0, :]).sum(1).view(n, 1, 1).float()
        gy = torch.arange(h, dtype=torch.float32).view(1, h, 1)
        gx = torch.arange(w, dtype=torch.float32).view(1, 1, w)
        cx = ((gx + 0.5) / vw).expand(n, h, w)
        cy = ((gy + 0.5) / vh).expand(n, h, w)
        wh = 0.05 * 2.0 ** lvl
        ok = (cx > 0.01) & (cx < 0.99) & (cy > 0.01) & (cy < 0.99) & (0.01 < wh < 0.99)
        valid.append(ok.reshape(n, h * w))
        cur += h * w
    valid = torch.cat(valid, 1)
    keep = (~mask_flat) & valid
    x = memory * keep[..., None].to(memory.dtype)
    return layer_norm(sd, "enc_output_norm", linear(sd, "enc_output", x))


# ----------------------------------------------------------------------------- F1 mask predictor
def mask_predictor(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """salience_transformer.py:16-47.  The global half is the mean over ALL tokens of the level."""
    z = F.gelu(linear(sd, prefix + ".layer1.1", layer_norm(sd, prefix + ".layer1.0", x)))
    h = z.shape[-1] // 2
    z_local, z_global = z[..., :h], z[..., h:]
    z_global = z_global.mean(1, keepdim=True).expand(-1, z.shape[1], -1)
    z = torch.cat([z_local, z_global], -1)
    z = F.gelu(linear(sd, prefix + ".layer2.0", z))
    z = F.gelu(linear(sd, prefix + ".layer2.2", z))
    return linear(sd, prefix + ".layer2.4", z)


# ----------------------------------------------------------------------------- F2 / F3 filtering
def token_budgets(masks: Sequence[torch.Tensor], level_filter_ratio: torch.Tensor):
    """salience_transformer.py:116-121 (fp32 product truncated by .int())."""
    valid = torch.stack([(~m).sum((1, 2)) for m in masks], -1)
    focus = (valid * level_filter_ratio.to(torch.float32)).int()
    level_token_nums = focus.max(0)[0]
    return focus.sum(-1), level_token_nums, valid


def topk_desc_stable(score: torch.Tensor, k: int):
    s, i = torch.sort(score, dim=1, descending=True, stable=True)
    return s[:, :k], i[:, :k]


def level_filtering(sd: SD, bom: torch.Tensor, mask_flat: torch.Tensor, shapes: torch.Tensor,
                    lsi: torch.Tensor, level_token_nums: torch.Tensor, predictor_prefix="enc_mask_predictor"):
    """salience_transformer.py:123-154 -- high level -> low level score prediction + per-level top-k."""
    B = bom.shape[0]
    L = shapes.shape[0]
    alpha = sd["alpha"]
    score_maps: List[Optional[torch.Tensor]] = [None] * L
    level_inds: List[Optional[torch.Tensor]] = [None] * L
    level_score: List[Optional[torch.Tensor]] = [None] * L
    score = None
    for lvl in range(L - 1, -1, -1):
        h, w = shapes[lvl].tolist()
        s0 = int(lsi[lvl])
        mem = bom[:, s0:s0 + h * w]
        m = mask_flat[:, s0:s0 + h * w]
        if lvl != L - 1:
            up = F.interpolate(score, size=(h, w), mode="bilinear", align_corners=True)
            up = up.view(B, 1, h * w).transpose(1, 2)
            mem = mem + mem * up * alpha[lvl]
        sc = mask_predictor(sd, predictor_prefix, mem)  # [B, hw, 1]
        valid_score = sc.squeeze(-1).masked_fill(m, sc.min())
        score = sc.transpose(1, 2).reshape(B, 1, h, w)
        ls, li = topk_desc_stable(valid_score, int(level_token_nums[lvl]))
        score_maps[lvl] = score
        level_inds[lvl] = li + s0
        level_score[lvl] = ls
    return score_maps, level_inds, level_score


def salience_filtering(score_maps, level_inds, level_score, mask_flat, layer_filter_ratio: torch.Tensor):
    """salience_transformer.py:156-168 -- global sort + per-layer prefixes + foreground score."""
    sel_score = torch.cat(level_score, 1)
    order = torch.sort(sel_score, dim=1, descending=True, stable=True)[1]
    sel_inds = torch.cat(level_inds, 1).gather(1, order)
    n = sel_inds.shape[1]
    counts = (n * layer_filter_ratio.to(torch.float32)).to(torch.int64)
    foreground_inds = [sel_inds[:, :int(r)] for r in counts]
    fg = flatten_levels(score_maps).squeeze(-1)
    fg = fg.masked_fill(mask_flat, fg.min())
    return foreground_inds, fg


# ----------------------------------------------------------------------------- M2 / M4 MSDA
def msda_core_c(value, shapes, lsi, loc, aw):
    """Closed-form op via the plain-C oracle (forward only)."""
    out = msda_c.msda_forward(value.detach().numpy(), shapes.numpy(), lsi.numpy(),
                              loc.detach().numpy(), aw.detach().numpy())
    return torch.from_numpy(out)


def msda_core_torch(value, shapes, lsi, loc, aw):
    """Differentiable closed-form restatement (ms_deform_im2col_cuda.cuh:22-73, 226-288)
    used where autograd is needed (backward oracle for the module-level path)."""
    B, Nv, M, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    out = value.new_zeros(B, Nq, M, D)
    bidx = torch.arange(B).view(B, 1, 1, 1)
    midx = torch.arange(M).view(1, 1, M, 1)
    for l in range(L):
        H, W = shapes[l].tolist()
        x = loc[:, :, :, l, :, 0] * W - 0.5
        y = loc[:, :, :, l, :, 1] * H - 0.5
        inside = (y > -1) & (x > -1) & (y < H) & (x < W)
        x0 = torch.floor(x)
        y0 = torch.floor(y)
        lx, ly = x - x0, y - y0
        x0, y0 = x0.long(), y0.long()
        acc = 0
        for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx),
                            (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = y0 + dy, x0 + dx
            ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)) + int(lsi[l])
            v = value[bidx, idx, midx]  # [B,Nq,M,P,D]
            acc = acc + v * (wgt * ok.to(value.dtype))[..., None]
        out = out + (acc * aw[:, :, :, l, :, None]).sum(3)
    return out.reshape(B, Nq, M * D)


def sampling_locations(ref, offsets, shapes, num_points):
    """ms_deform_attn.py:339-355."""
    if ref.shape[-1] == 2:
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(offsets.dtype)
        return ref[:, :, None, :, None, :] + offsets / norm[None, None, None, :, None, :]
    if ref.shape[-1] == 4:
        return ref[:, :, None, :, None, :2] + offsets / num_points * ref[:, :, None, :, None, 2:] * 0.5
    raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(ref.shape[-1]))


def msda_module(sd: SD, prefix: str, query, ref, value, shapes, lsi, pad_mask, heads: int, levels: int,
                points: int, core=msda_core_c):
    """ms_deform_attn.py:286-377."""
    B, Nq, E = query.shape
    Nv = value.shape[1]
    v = linear(sd, prefix + ".value_proj", value)
    if pad_mask is not None:
        v = v.masked_fill(pad_mask[..., None], 0.0)
    v = v.view(B, Nv, heads, E // heads)
    off = linear(sd, prefix + ".sampling_offsets", query).view(B, Nq, heads, levels, points, 2)
    aw = linear(sd, prefix + ".attention_weights", query).view(B, Nq, heads, levels * points)
    aw = aw.softmax(-1).view(B, Nq, heads, levels, points)
    loc = sampling_locations(ref, off, shapes, points)
    out = core(v.contiguous(), shapes, lsi, loc.contiguous(), aw.contiguous())
    return linear(sd, prefix + ".output_proj", out)


# ----------------------------------------------------------------------------- E1-E4 encoder
def mha_self(sd: SD, prefix: str, qk: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True) forward, q = k = qk, value = v
    (salience_transformer.py:371-376), restated."""
    B, N, E = qk.shape
    hd = E // heads
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = F.linear(qk, w[:E], b[:E]).view(B, N, heads, hd).transpose(1, 2)
    k = F.linear(qk, w[E:2 * E], b[E:2 * E]).view(B, N, heads, hd).transpose(1, 2)
    vv = F.linear(v, w[2 * E:], b[2 * E:]).view(B, N, heads, hd).transpose(1, 2)
    att = (q * (1.0 / math.sqrt(hd))) @ k.transpose(-1, -2)
    o = att.softmax(-1) @ vv
    o = o.transpose(1, 2).reshape(B, N, E)
    return linear(sd, prefix + ".out_proj", o)


def encoder_reference_points(shapes: torch.Tensor, valid_ratios: torch.Tensor) -> torch.Tensor:
    """salience_transformer.py:418-432 -> [B, S, L, 2]."""
    refs = []
    for lvl, (h, w) in enumerate(shapes.tolist()):
        ry = (torch.arange(h, dtype=torch.float32) + 0.5).view(h, 1).expand(h, w).reshape(-1)
        rx = (torch.arange(w, dtype=torch.float32) + 0.5).view(1, w).expand(h, w).reshape(-1)
        ry = ry[None] / (valid_ratios[:, None, lvl, 1] * h)
        rx = rx[None] / (valid_ratios[:, None, lvl, 0] * w)
        refs.append(torch.stack((rx, ry), -1))
    ref = torch.cat(refs, 1)
    return ref[:, :, None] * valid_ratios[:, None]


def encoder_layer(sd: SD, prefix: str, query, query_pos, value, ref, shapes, lsi, pad_mask, score_tgt,
                  fg_pre, heads, levels, points, topk_sa, core=msda_core_c, collect_sel=None):
    """salience_transformer.py:353-396 (dropout = 0).  ``collect_sel``: list that receives the top-k index set."""
    E = query.shape[-1]
    mc = score_tgt.max(-1)[0] * fg_pre
    sel = topk_desc_stable(mc, topk_sa)[1]
    if collect_sel is not None:
        collect_sel.append(sel)
    sel_e = sel.unsqueeze(-1).expand(-1, -1, E)
    tgt = torch.gather(query, 1, sel_e)
    pos = torch.gather(query_pos, 1, sel_e)
    tgt = layer_norm(sd, prefix + ".pre_norm", tgt + mha_self(sd, prefix + ".pre_attention", tgt + pos, tgt, heads))
    query = query.scatter(1, sel_e, tgt)
    src2 = msda_module(sd, prefix + ".self_attn", query + query_pos, ref, value, shapes, lsi, pad_mask,
                       heads, levels, points, core)
    query = layer_norm(sd, prefix + ".norm1", query + src2)
    ffn = linear(sd, prefix + ".linear2", F.relu(linear(sd, prefix + ".linear1", query)))
    return layer_norm(sd, prefix + ".norm2", query + ffn)


def learned_background(sd: SD, prefix: str, masks: Sequence[torch.Tensor]) -> torch.Tensor:
    """position_encoding.py:70-99 applied per level and flattened (salience_transformer.py:488-492)."""
    outs = []
    for m in masks:
        B, h, w = m.shape
        xe = sd[prefix + ".col_embed.weight"][:w]  # [w, E/2]
        ye = sd[prefix + ".row_embed.weight"][:h]  # [h, E/2]
        pos = torch.cat([xe[None].expand(h, w, -1), ye[:, None].expand(h, w, -1)], -1)  # [h,w,E]
        outs.append(pos.reshape(1, h * w, -1).expand(B, -1, -1))
    return torch.cat(outs, 1)


def encoder(sd: SD, query, shapes, lsi, valid_ratios, query_pos, pad_mask, foreground_score,
            focus_token_nums, foreground_inds, masks, heads, levels, points, topk_sa, num_layers,
            prefix="encoder", core=msda_core_c, collect=None, collect_sel=None, collect_in=None, timings=None):
    """salience_transformer.py:434-497.  ``timings`` (dict, optional): wall seconds per encoder layer are added under
    ``encoder_layer_<k>`` (the CPU-baseline leg of bench.py reports them per stage)."""
    import time
    ref_all = encoder_reference_points(shapes, valid_ratios)
    B, S, Lr, two = ref_all.shape
    E = query.shape[-1]
    value = output = query
    inds_e = None
    for k in range(num_layers):
        t_layer = time.perf_counter()
        inds = foreground_inds[k]
        inds_e = inds.unsqueeze(-1).expand(-1, -1, E)
        q = torch.gather(output, 1, inds_e)
        qp = torch.gather(query_pos, 1, inds_e)
        fg = torch.gather(foreground_score, 1, inds)
        ref = torch.gather(ref_all.view(B, S, -1), 1, inds.unsqueeze(-1).expand(-1, -1, Lr * two)).view(B, -1, Lr, two)
        score_tgt = linear(sd, prefix + ".enhance_mcsp", q)
        if collect_in is not None:   # the layer's inputs, for teacher-forced per-layer checks
            collect_in.append(dict(query=q, query_pos=qp, ref=ref, fg=fg, inds=inds))
        q = encoder_layer(sd, f"{prefix}.layers.{k}", q, qp, value, ref, shapes, lsi, pad_mask, score_tgt, fg,
                          heads, levels, points, topk_sa, core, collect_sel=collect_sel)
        if collect is not None:
            collect.append(q)
        new = []
        for i in range(B):
            n = int(focus_token_nums[i])
            new.append(output[i].scatter(0, inds[i, :n].unsqueeze(-1).expand(-1, E), q[i, :n]))
        output = torch.stack(new)
        if timings is not None:
            timings[f"encoder_layer_{k}"] = timings.get(f"encoder_layer_{k}", 0.0) + time.perf_counter() - t_layer
    bg = learned_background(sd, prefix + ".background_embedding", masks).clone()
    bg.scatter_(1, inds_e, 0)
    bg = bg * (~pad_mask).unsqueeze(-1)
    return output + bg


# ----------------------------------------------------------------------------- whole hot path
def hot_path(sd: SD, feats, masks, pos, heads=8, points=4, topk_sa=300, num_layers=6, core=msda_core_c, timings=None):
    """SalienceTransformer.forward up to ``memory`` (salience_transformer.py:97-183).  ``timings`` (dict, optional)
    receives wall seconds of ``F0_F3_filtering``, ``encoder_layer_<k>`` and ``msda_core`` (the op alone, summed)."""
    import time
    t_start = time.perf_counter()
    if timings is not None:
        inner = core

        def core(*a, **kw):   # noqa: F811 -- the same op, timed
            t0 = time.perf_counter()
            r = inner(*a, **kw)
            timings["msda_core"] = timings.get("msda_core", 0.0) + time.perf_counter() - t0
            return r
    L = len(feats)
    feat_flat = flatten_levels(feats)
    mask_flat = flatten_levels(masks)
    pos_flat = level_pos_embed(sd, pos)
    shapes, lsi, vr = level_misc(masks)
    bom = backbone_output_memory(sd, feat_flat + pos_flat, mask_flat, shapes)
    focus, level_token_nums, _ = token_budgets(masks, sd["level_filter_ratio"])
    score_maps, level_inds, level_score = level_filtering(sd, bom, mask_flat, shapes, lsi, level_token_nums)
    fg_inds, fg_score = salience_filtering(score_maps, level_inds, level_score, mask_flat,
                                           sd["layer_filter_ratio"])
    if timings is not None:
        timings["F0_F3_filtering"] = timings.get("F0_F3_filtering", 0.0) + time.perf_counter() - t_start
    layer_out, layer_sel, layer_in = [], [], []
    memory = encoder(sd, feat_flat, shapes, lsi, vr, pos_flat, mask_flat, fg_score, focus, fg_inds, masks,
                     heads, L, points, topk_sa, num_layers, core=core, collect=layer_out, collect_sel=layer_sel,
                     collect_in=layer_in, timings=timings)
    return dict(feat_flatten=feat_flat, mask_flatten=mask_flat, lvl_pos_embed_flatten=pos_flat,
                spatial_shapes=shapes, level_start_index=lsi, valid_ratios=vr, backbone_output_memory=bom,
                focus_token_nums=focus, level_token_nums=level_token_nums, score_maps=score_maps,
                level_inds=level_inds, level_score=level_score, foreground_inds=fg_inds,
                foreground_score=fg_score, layer_out=layer_out, layer_sel=layer_sel, layer_in=layer_in, memory=memory)


# ----------------------------------------------------------------------------- D1 decoder (row N2)
def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """util/misc.py:31-35."""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def coordinate_sine_embed(pos: torch.Tensor, num_pos_feats: int, temperature: float = 10000.0) -> torch.Tensor:
    """position_encoding.py:105-132 with exchange_xy=True: per coordinate c, feature 2j is sin(2*pi*c / T^(2j/F)),
    feature 2j+1 the cosine; the x and y blocks are emitted in (y, x) order, further coordinates follow in order."""
    out = []
    j = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (j // 2) / num_pos_feats)
    for c in range(pos.shape[-1]):
        a = pos[..., c, None] * (2 * math.pi) / dim_t
        e = torch.zeros_like(a)
        e[..., 0::2] = a[..., 0::2].sin()
        e[..., 1::2] = a[..., 1::2].cos()
        out.append(e)
    if len(out) >= 2:
        out[0], out[1] = out[1], out[0]
    return torch.cat(out, -1)


def mlp(sd: SD, prefix: str, x: torch.Tensor, num_layers: int) -> torch.Tensor:
    """basic.py:6-26."""
    for i in range(num_layers):
        x = linear(sd, "{}.layers.{}".format(prefix, i), x)
        if i + 1 < num_layers:
            x = F.relu(x)
    return x


def decoder_layer(sd: SD, prefix: str, query, query_pos, ref_in, value, shapes, lsi, pad_mask, heads: int,
                  levels: int, points: int, core=msda_core_c):
    """salience_transformer.py:553-588 (eval mode: dropouts are identity, no attention mask)."""
    qk = query + query_pos
    query = layer_norm(sd, prefix + ".norm2", query + mha_self(sd, prefix + ".self_attn", qk, query, heads))
    q2 = msda_module(sd, prefix + ".cross_attn", query + query_pos, ref_in, value, shapes, lsi, pad_mask, heads,
                     levels, points, core=core)
    query = layer_norm(sd, prefix + ".norm1", query + q2)
    h = F.relu(linear(sd, prefix + ".linear1", query))
    return layer_norm(sd, prefix + ".norm3", query + linear(sd, prefix + ".linear2", h))


def decoder(sd: SD, query, reference_points, value, shapes, lsi, valid_ratios, pad_mask, num_layers: int,
            heads: int = 8, points: int = 4, core=msda_core_c, trace=None):
    """salience_transformer.py:625-674 -> (class logits [num_layers,B,Nq,C], boxes [num_layers,B,Nq,4]).
    ``trace`` (a list) receives per layer ``(query_in, reference_points_in, query_pos, ref_in, query_out)``."""
    E = query.shape[-1]
    levels = shapes.shape[0]
    ratio = torch.cat([valid_ratios, valid_ratios], -1)[:, None]
    classes, coords = [], []
    for i in range(num_layers):
        ref_in = reference_points.detach()[:, :, None] * ratio
        query_pos = mlp(sd, "ref_point_head", coordinate_sine_embed(ref_in[:, :, 0, :], E // 2), 2)
        query_in = query
        query = decoder_layer(sd, "layers.{}".format(i), query, query_pos, ref_in, value, shapes, lsi, pad_mask,
                              heads, levels, points, core=core)
        if trace is not None:
            trace.append((query_in, reference_points, query_pos, ref_in, query))
        normed = layer_norm(sd, "norm", query)
        classes.append(linear(sd, "class_head.{}".format(i), normed))
        coords.append((mlp(sd, "bbox_head.{}".format(i), normed, 3) + inverse_sigmoid(reference_points)).sigmoid())
        if i + 1 == num_layers:
            break
        # the next reference keeps the gradient through bbox_head(query) ("look forward twice", :666-668)
        reference_points = (mlp(sd, "bbox_head.{}".format(i), query, 3)
                            + inverse_sigmoid(reference_points.detach())).sigmoid()
    return torch.stack(classes), torch.stack(coords)


# ----------------------------------------------------------------------------- row N1: two-stage proposals
def encoder_output_proposals(sd: SD, memory: torch.Tensor, mask_flat: torch.Tensor, shapes: torch.Tensor):
    """base_transformer.py:74-112, both return values: (LayerNorm(Linear(memory * keep)), proposal logits with
    +inf on padding / out-of-range tokens)."""
    n = memory.shape[0]
    props = []
    cur = 0
    for lvl, (h, w) in enumerate(shapes.tolist()):
        m = mask_flat[:, cur:cur + h * w].view(n, h, w)
        vh = (~m[:, :, 0]).sum(1).view(n, 1, 1).float()
        vw = (~m[:, 0, :]).sum(1).view(n, 1, 1).float()
        gy = torch.arange(h, dtype=torch.float32).view(1, h, 1)
        gx = torch.arange(w, dtype=torch.float32).view(1, 1, w)
        cx = ((gx + 0.5) / vw).expand(n, h, w)
        cy = ((gy + 0.5) / vh).expand(n, h, w)
        wh = torch.ones(n, h, w) * 0.05 * 2.0 ** lvl
        props.append(torch.stack([cx, cy, wh, wh], -1).view(n, h * w, 4))
        cur += h * w
    prop = torch.cat(props, 1)
    valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
    logit = torch.log(prop / (1 - prop))
    logit = logit.masked_fill(mask_flat[..., None] | ~valid, float("inf"))
    x = memory * (~mask_flat[..., None]) * valid
    return layer_norm(sd, "enc_output_norm", linear(sd, "enc_output", x)), logit


def nms_greedy(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Restatement of ``torchvision.ops.nms`` (torchvision is a dependency of the reference that is NOT present in
    this image -- requirements.txt of the reference lists it unpinned; the algorithm below is the one published in
    torchvision/csrc/ops/cpu/nms_kernel.cpp, v0.13 ... v0.22: boxes are visited in descending score order (stable
    sort), a box is kept unless a previously kept box overlaps it with inter / (area_i + area_j - inter) >
    iou_threshold; widths/heights are x2-x1 / y2-y1 clamped at 0, all arithmetic in the box dtype).  Returns the kept
    indices in descending score order.  PARITY UNPINNED: no torchvision build here to check this against."""
    order = torch.sort(scores, descending=True, stable=True)[1].tolist()
    b = boxes.float().numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    thr = np.float32(iou_threshold)
    kept = np.zeros(len(order), dtype=np.int64)
    nk = 0
    for i in order:
        j = kept[:nk]
        iw = np.maximum(np.float32(0), np.minimum(b[i, 2], b[j, 2]) - np.maximum(b[i, 0], b[j, 0]))
        ih = np.maximum(np.float32(0), np.minimum(b[i, 3], b[j, 3]) - np.maximum(b[i, 1], b[j, 1]))
        inter = (iw * ih).astype(np.float32)
        if not np.any(inter / (area[i] + area[j] - inter) > thr):
            kept[nk] = i
            nk += 1
    kept = kept[:nk].tolist()
    return torch.tensor(kept, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """``torchvision.ops.batched_nms`` (torchvision/ops/boxes.py): NMS per category, the union of the kept indices
    returned in descending score order (ties in list order).  Same provenance note as ``nms_greedy``."""
    keep = torch.zeros(scores.shape[0], dtype=torch.bool)
    for c in torch.unique(idxs).tolist():
        cur = torch.nonzero(idxs == c)[:, 0]
        keep[cur[nms_greedy(boxes[cur], scores[cur], iou_threshold)]] = True
    kept = torch.nonzero(keep)[:, 0]
    return kept[torch.sort(scores[kept], descending=True, stable=True)[1]]


def nms_inputs(topk_index: torch.Tensor, shapes: torch.Tensor, lsi: torch.Tensor):
    """salience_transformer.py:249-277: unit boxes around the grid cell of every selected token and the
    (image, level) category id."""
    B, K = topk_index.shape
    flat = topk_index.reshape(-1)
    level = (flat[:, None] >= lsi[None, :]).sum(1) - 1
    width = shapes[level, 1]
    sp = flat - lsi[level]
    x = (sp % width).float()
    y = torch.div(sp, width, rounding_mode="trunc").float()
    boxes = torch.stack([x - 1.0, y - 1.0, x + 1.0, y + 1.0], -1)
    image = torch.arange(B).repeat_interleave(K)
    return boxes, level + lsi.shape[0] * image, image


def nms_on_topk_index(topk_scores, topk_index, shapes, lsi, num_proposals: int, iou_threshold: float = 0.3,
                      nms=batched_nms):
    """salience_transformer.py:249-295."""
    B, K = topk_scores.shape
    boxes, idxs, image = nms_inputs(topk_index, shapes, lsi)
    kept = nms(boxes, topk_scores.reshape(-1), idxs, iou_threshold)
    flat = topk_index.reshape(-1)
    per_image = [flat[kept[image[kept] == b]] for b in range(B)]
    n = min([num_proposals] + [int(t.shape[0]) for t in per_image])
    return torch.stack([t[:n] for t in per_image])


def two_stage_proposals(sd: SD, memory, mask_flat, shapes, lsi, num_proposals: int, iou_threshold: float = 0.3):
    """salience_transformer.py:194-212 -> dict(enc_outputs_class, enc_outputs_coord (selected), topk_scores,
    topk_index (before NMS), index (after NMS), class_all, coord_all)."""
    om, logit = encoder_output_proposals(sd, memory, mask_flat, shapes)
    cls_all = linear(sd, "encoder_class_head", om)
    coord_all = (mlp(sd, "encoder_bbox_head", om, 3) + logit).sigmoid()
    k = min(num_proposals * 4, cls_all.shape[1])
    topk_scores, topk_index = topk_desc_stable(cls_all.max(-1)[0], k)
    index = nms_on_topk_index(topk_scores, topk_index, shapes, lsi, num_proposals, iou_threshold)
    C = cls_all.shape[-1]
    return dict(class_all=cls_all, coord_all=coord_all, topk_scores=topk_scores, topk_index=topk_index, index=index,
                enc_outputs_class=cls_all.gather(1, index[..., None].expand(-1, -1, C)),
                enc_outputs_coord=coord_all.gather(1, index[..., None].expand(-1, -1, 4)))


def _sub(sd: SD, prefix: str) -> SD:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------- row N3: RepVGGPluX neck
def silu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def conv_norm(sd: SD, prefix: str, x: torch.Tensor, stride: int = 1, groups: int = 1, act: bool = True,
              eps: float = 1e-5, new_stats: Optional[SD] = None, momentum: float = 0.1) -> torch.Tensor:
    """Conv2dNormActivation (models/bricks/misc.py:61-103): bias-free convolution with "same" padding, BatchNorm2d,
    optional SiLU.  ``prefix.0`` is the convolution, ``prefix.1`` the norm.  Eval mode (``new_stats`` None): the
    running statistics.  Training mode (``new_stats`` a dict): the batch statistics over (B, H, W) -- biased variance
    for the normalisation -- and the updated running statistics (momentum 0.1, UNBIASED variance, as nn.BatchNorm2d;
    single process, i.e. what SyncBatchNorm computes over all ranks' pixels) are written into ``new_stats``."""
    w = sd[prefix + "0.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=(w.shape[-1] - 1) // 2, groups=groups)
    if new_stats is None:
        mean, var = sd[prefix + "1.running_mean"], sd[prefix + "1.running_var"]
    else:
        mean = y.mean((0, 2, 3))
        var = y.var((0, 2, 3), unbiased=False)
        n = y.numel() // y.shape[1]
        with torch.no_grad():
            new_stats[prefix + "1.running_mean"] = (1 - momentum) * sd[prefix + "1.running_mean"] + momentum * mean
            new_stats[prefix + "1.running_var"] = (1 - momentum) * sd[prefix + "1.running_var"] \
                + momentum * var * (n / max(n - 1, 1))
    inv = torch.rsqrt(var + eps) * sd[prefix + "1.weight"]
    y = (y - mean.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1) + sd[prefix + "1.bias"].view(1, -1, 1, 1)
    return silu(y) if act else y


def attention_pool_gate(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """SqueezeAndExcitation (models/bricks/basic.py:29-54): the pooled context is NOT a mean -- every pixel gets the
    logit ``conv_mask(x)``, a softmax over all H*W pixels weights the pixels, the weighted channel sums go through
    C -> C/16 -> ReLU -> C -> sigmoid, and that gate scales x."""
    B, C, H, W = x.shape
    logit = F.conv2d(x, sd[prefix + "conv_mask.weight"], sd[prefix + "conv_mask.bias"]).view(B, H * W)
    context = torch.einsum("bcp,bp->bc", x.view(B, C, H * W), logit.softmax(-1))
    hidden = torch.relu(context @ sd[prefix + "se_module.0.weight"].view(-1, C).t())
    gate = torch.sigmoid(hidden @ sd[prefix + "se_module.2.weight"].view(C, -1).t())
    return gate.view(B, C, 1, 1) * x


def repvgg_block(sd: SD, prefix: str, x: torch.Tensor, groups: int, new_stats: Optional[SD] = None) -> torch.Tensor:
    """RepVggPluXBlock.forward (models/necks/repnet.py:61-64) with in == out channels (identity shortcut, alpha = 1):
    grouped 3x3 + grouped 1x1, each with its own BatchNorm, SiLU, the attention-pooled gate, plus x."""
    y = conv_norm(sd, prefix + "conv1.", x, groups=groups, act=False, new_stats=new_stats) \
        + conv_norm(sd, prefix + "conv2.", x, groups=groups, act=False, new_stats=new_stats)
    return attention_pool_gate(sd, prefix + "se_module.", silu(y)) + x


def csp_layer(sd: SD, prefix: str, x: torch.Tensor, groups: int, num_blocks: int = 3,
              new_stats: Optional[SD] = None) -> torch.Tensor:
    """CSPRepPluXLayer.forward (repnet.py:120-123), expansion 1 (conv3 is the identity)."""
    y = conv_norm(sd, prefix + "conv1.", x, new_stats=new_stats)
    for j in range(num_blocks):
        y = repvgg_block(sd, f"{prefix}bottlenecks.{j}.", y, groups, new_stats)
    return y + conv_norm(sd, prefix + "conv2.", x, new_stats=new_stats)


def neck(sd: SD, feats: Sequence[torch.Tensor], groups: int = 4, new_stats: Optional[SD] = None) -> List[torch.Tensor]:
    """RepVGGPluXNetwork.forward (repnet.py:211-245) on NCHW levels, fine to coarse; ``sd`` holds the neck's own keys
    (``lateral_convs.*``, ``layer_blocks.*``, ``downsample_blocks.*``, ``pan_blocks.*``).  ``new_stats`` None: eval
    mode (what the HIP path implements); a dict: training mode, see ``conv_norm`` (restated for the next round's
    training form of the neck; differentiable through torch autograd)."""
    L = len(feats)
    inner = [feats[-1]]
    for idx in range(L - 1, 0, -1):  # top-down
        high = conv_norm(sd, f"lateral_convs.{idx - 1}.", inner[0], new_stats=new_stats)
        inner[0] = high
        up = F.interpolate(high, size=feats[idx - 1].shape[-2:], mode="nearest")
        inner.insert(0, csp_layer(sd, f"layer_blocks.{idx - 1}.", torch.cat([up, feats[idx - 1]], 1), groups,
                                  new_stats=new_stats))
    outs = [inner[0]]
    for idx in range(L - 1):  # bottom-up
        down = conv_norm(sd, f"downsample_blocks.{idx}.", outs[-1], stride=2, new_stats=new_stats)
        outs.append(csp_layer(sd, f"pan_blocks.{idx}.", torch.cat([down, inner[idx + 1]], 1), groups,
                              new_stats=new_stats))
    return outs


def neck_on_memory(sd: SD, memory: torch.Tensor, shapes: torch.Tensor, groups: int = 4) -> torch.Tensor:
    """salience_transformer.py:185-192: token-major memory -> NCHW levels -> neck -> token-major memory."""
    B, _, C = memory.shape
    sizes = [int(h) * int(w) for h, w in shapes.tolist()]
    levels = [m.transpose(1, 2).reshape(B, C, int(h), int(w))
              for m, (h, w) in zip(memory.split(sizes, 1), shapes.tolist())]
    return torch.cat([o.flatten(2).transpose(1, 2) for o in neck(sd, levels, groups)], 1)


def transformer(sd: SD, feats, masks, pos, num_proposals: int, heads=8, points=4, topk_sa=300, enc_layers=6,
                dec_layers=6, core=msda_core_c):
    """SalienceTransformer.forward, inference (no denoising queries; salience_transformer.py:97-226).  The neck runs
    iff ``sd`` carries ``neck.*`` keys."""
    hp = hot_path(sd, feats, masks, pos, heads, points, topk_sa, enc_layers, core=core)
    memory = hp["memory"]
    if any(k.startswith("neck.") for k in sd):
        memory = neck_on_memory(_sub(sd, "neck."), memory, hp["spatial_shapes"])
    ts = two_stage_proposals(sd, memory, hp["mask_flatten"], hp["spatial_shapes"], hp["level_start_index"],
                             num_proposals)
    B = memory.shape[0]
    target = sd["tgt_embed.weight"][None].expand(B, -1, -1)
    ref = ts["enc_outputs_coord"]
    cls, box = decoder(_sub(sd, "decoder."), target, ref, memory,
                       hp["spatial_shapes"], hp["level_start_index"], hp["valid_ratios"], hp["mask_flatten"],
                       dec_layers, heads, points, core=core)
    return dict(outputs_classes=cls, outputs_coords=box, salience_score=hp["score_maps"], memory=memory, **ts)


# ----------------------------------------------------------------------------- row N4: salience criterion
def salience_targets(boxes_xyxy: Sequence[torch.Tensor], level_shapes, strides, limit_range, noise_scale: float = 0.0,
                     noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """models/detectors/salience_detr.py:36-45, 61-116: per level and image the scale-independent salience confidence
    of every pixel centre, [B, S].  ``boxes_xyxy``: per image [m,4] in input-image pixels; ``strides``: (sy, sx) per
    level; ``noise`` [B,S] stands for the reference's ``torch.rand_like`` draws when ``noise_scale`` > 0."""
    out = []
    for lvl, ((h, w), (sy, sx)) in enumerate(zip(level_shapes, strides)):
        cy = (torch.linspace(0.5, h - 0.5, h, dtype=torch.float32) * sy).view(h, 1).expand(h, w).reshape(-1)
        cx = (torch.linspace(0.5, w - 0.5, w, dtype=torch.float32) * sx).view(1, w).expand(h, w).reshape(-1)
        lo, hi = limit_range[lvl]
        per_image = []
        for gt in boxes_xyxy:
            if gt.shape[0] == 0:
                per_image.append(torch.zeros(h * w))
                continue
            l = cx[:, None] - gt[None, :, 0]
            t = cy[:, None] - gt[None, :, 1]
            r = gt[None, :, 2] - cx[:, None]
            b = gt[None, :, 3] - cy[:, None]
            d = torch.stack([l, t, r, b], -1)
            dmin, dmax = d.min(-1)[0], d.max(-1)[0]
            inside = dmin > 0
            in_level = (dmax > lo) & (dmax <= hi)
            dx = (l - r) / (l + r)
            dy = (t - b) / (t + b)
            conf = 1 - torch.sqrt(dx ** 2 + dy ** 2) / 2
            conf = torch.where(inside, conf, torch.zeros_like(conf))
            m = conf.max(-1)[0]
            pos = (inside & in_level).any(-1)
            per_image.append(torch.where(pos, m, torch.zeros_like(m)))
        out.append(torch.stack(per_image))
    target = torch.cat(out, 1)
    if noise_scale:
        target = (1 - noise_scale) * target + noise_scale * noise
    return target


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2.0):
    """models/bricks/losses.py:4-13 (the weight keeps its gradient)."""
    prob = inputs.sigmoid()
    weight = (1 - alpha) * prob ** gamma * (1 - targets) + targets * alpha * (1 - prob) ** gamma
    loss = F.binary_cross_entropy_with_logits(inputs, targets.to(inputs.dtype), reduction="none") * weight
    return (loss.sum(1) / max(loss.shape[1], 1)).sum() / num_boxes


def salience_criterion(foreground_mask: Sequence[torch.Tensor], boxes_cxcywh: Sequence[torch.Tensor], strides, image_sizes,
                       limit_range=((-1, 64), (64, 128), (128, 256), (256, 99999)), noise_scale: float = 0.0,
                       alpha: float = 0.25, gamma: float = 2.0, noise: Optional[torch.Tensor] = None):
    """SalienceCriterion.forward (salience_detr.py:27-59) -> (loss_salience, mask_targets [B,S])."""
    xyxy = []
    for bx, (ih, iw) in zip(boxes_cxcywh, image_sizes):
        cx, cy, w, h = bx.unbind(-1)
        xyxy.append(torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)
                    * torch.tensor([iw, ih, iw, ih], dtype=bx.dtype))
    shapes = [tuple(m.shape[-2:]) for m in foreground_mask]
    target = salience_targets(xyxy, shapes, strides, limit_range, noise_scale, noise)
    logits = torch.cat([m.flatten(-2) for m in foreground_mask], -1).squeeze(1)
    num_pos = (target > 0.5 * noise_scale).sum().clamp(min=1)
    loss = sigmoid_focal_loss(logits, target, num_pos, alpha, gamma) * logits.shape[1]
    return loss, target
