"""GPU: row N2 -- the decoder (self-attention, MSDA cross-attention with box references, FFN, iterative box
refinement) against the reference's golden outputs and, at the full 900-query / 800x1333 size, against the oracle.

Bars: fp32 logits 1e-3, boxes 1e-4 (north_star tolerance); bf16 is a closeness check with its bound in the test.
"""
import os

import numpy as np
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import synthetic as syn
from salience_detr_amd.salience_decoder import SalienceTransformerDecoder, SalienceTransformerDecoderLayer
from test_decoder_cpu import _t, build_decoder

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("query", "ref", "memory", "shapes", "lsi", "valid_ratios", "mask")


@pytest.fixture(scope="module")
def cases():
    return np.load(os.path.join(G, "decoder_cases.npz"))


def _inputs(d, tag, device="cuda"):
    return [_t(d[f"{tag}.{k}"]).to(device) for k in KEYS]


@pytest.mark.parametrize("tag", ["small", "e256"])
def test_decoder_matches_reference_golden(cases, tag):
    d = cases
    dec, _, _ = build_decoder(d, tag)
    dec = dec.cuda()
    with torch.no_grad():
        cls, box = dec(*_inputs(d, tag))
    assert (cls.cpu() - _t(d[f"{tag}.classes"])).abs().max() < 1e-3
    assert (box.cpu() - _t(d[f"{tag}.boxes"])).abs().max() < 1e-4


def test_decoder_without_padding_mask_and_with_attention_mask(cases):
    d = cases
    dec, sd, heads = build_decoder(d, "small")
    dec = dec.cuda()
    q, ref, mem, shapes, lsi, vr, _ = _inputs(d, "small")
    with torch.no_grad():
        cls, box = dec(q, ref, mem, shapes, lsi, vr, None, None)
    ocls, obox = R.decoder(sd, q.cpu(), ref.cpu(), mem.cpu(), shapes.cpu(), lsi.cpu(), vr.cpu(), None, dec.num_layers,
                           heads=heads)
    assert (cls.cpu() - ocls).abs().max() < 1e-3 and (box.cpu() - obox).abs().max() < 1e-4
    # an all-False attention mask (nothing blocked) must not change the result (denoising groups pass one, :575)
    blocked = torch.zeros(q.shape[1], q.shape[1], dtype=torch.bool, device="cuda")
    with torch.no_grad():
        cls2, box2 = dec(q, ref, mem, shapes, lsi, vr, None, blocked)
    assert (cls2 - cls).abs().max() < 1e-5 and (box2 - box).abs().max() < 1e-6


def test_decoder_autograd_path_matches_native_and_oracle_grads(cases):
    d = cases
    dec, sd, heads = build_decoder(d, "small")
    dec = dec.cuda()
    args = _inputs(d, "small")
    with torch.no_grad():
        cls_n, box_n = dec(*args)
    q = args[0].clone().requires_grad_(True)
    cls_g, box_g = dec(q, *args[1:])
    assert (cls_g - cls_n).abs().max() < 1e-4 and (box_g - box_n).abs().max() < 1e-5
    gc, gb = syn.det_randn("dec.gc", tuple(cls_g.shape)).cuda(), syn.det_randn("dec.gb", tuple(box_g.shape)).cuda()
    ((cls_g * gc).sum() + (box_g * gb).sum()).backward()
    # oracle gradients through the differentiable closed-form sampler
    qo = args[0].cpu().clone().requires_grad_(True)
    cpu = [a.cpu() for a in args[1:]]
    ocls, obox = R.decoder(sd, qo, cpu[0], cpu[1], cpu[2], cpu[3], cpu[4], cpu[5], dec.num_layers, heads=heads,
                           core=R.msda_core_torch)
    ((ocls * gc.cpu()).sum() + (obox * gb.cpu()).sum()).backward()
    scale = qo.grad.abs().max()
    assert (q.grad.cpu() - qo.grad).abs().max() < 2e-3 * scale
    w = dec.layers[0].cross_attn.value_proj.weight
    assert w.grad is not None and torch.isfinite(w.grad).all() and w.grad.abs().max() > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_query_sine_embed_and_box_refine_kernels(cases, dtype):
    from salience_detr_amd.filter_ops import box_refine, decoder_query_sine_embed
    B, Nq, L, F = 3, 37, 4, 128
    ref = torch.cat([syn.det_rand("k.c", (B, Nq, 2)) * 1.2 - 0.1, syn.det_rand("k.w", (B, Nq, 2))], -1)
    ref[0, 0] = torch.tensor([0.0, 1.0, 1e-4, 0.9999])        # the clamps of inverse_sigmoid
    ref[0, 1] = torch.tensor([-0.2, 1.3, 0.5, 1e-3])
    vr = syn.det_rand("k.vr", (B, L, 2)) * 0.5 + 0.5
    ref_in, sine = decoder_query_sine_embed(ref.cuda(), vr.cuda(), F, dtype)
    want_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]
    assert torch.equal(ref_in.cpu(), want_in)
    want = R.coordinate_sine_embed(want_in[:, :, 0, :], F)
    assert sine.dtype == dtype and sine.shape == (B, Nq, 4 * F)
    assert (sine.float().cpu() - want).abs().max() < (2e-6 if dtype == torch.float32 else 4e-3)
    # the reference's own helper vectors (F = 16, fp32)
    d = cases
    pos = _t(d["helper.pos"]).reshape(1, -1, 4)
    _, s16 = decoder_query_sine_embed(pos.cuda(), torch.ones(1, 1, 2, device="cuda"), 16, torch.float32)
    assert (s16.cpu().reshape(d["helper.sine16"].shape) - _t(d["helper.sine16"])).abs().max() < 2e-6
    delta = syn.det_randn("k.d", (2, B, Nq, 4)) * 2
    got = box_refine(delta.to(dtype).cuda(), ref.cuda())
    want = (delta.to(dtype).float() + R.inverse_sigmoid(ref)).sigmoid()
    assert got.dtype == torch.float32 and got.shape == (2, B, Nq, 4)
    assert (got.cpu() - want).abs().max() < 1e-6
    assert (box_refine(delta[1].cuda(), ref.cuda()).cpu() - want[1]).abs().max() < (1e-6 if dtype == torch.float32 else 1e-2)
    x = _t(d["helper.isig_x"])
    pad = torch.zeros(12)
    pad[:10] = x
    sig = box_refine(torch.zeros(3, 4, device="cuda"), pad.view(3, 4).cuda()).cpu().view(-1)[:10]
    assert (sig - _t(d["helper.isig_y"]).sigmoid()).abs().max() < 1e-6


def _full_size(dtype):
    E, heads, d_ffn, layers, classes, Nq, B = 256, 8, 2048, 6, 91, 900, 2
    level_shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    layer = SalienceTransformerDecoderLayer(embed_dim=E, d_ffn=d_ffn, n_heads=heads, dropout=0.0)
    dec = SalienceTransformerDecoder(layer, layers, classes)
    sd = syn.det_state_dict(dec.state_dict())
    dec.load_state_dict(sd)
    shapes = torch.tensor(level_shapes, dtype=torch.int64)
    sizes = shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    Nv = int(sizes.sum())
    query = syn.det_randn("decfull.query", (B, Nq, E))
    memory = syn.det_randn("decfull.memory", (B, Nv, E))
    ref = torch.cat([syn.det_rand("decfull.cxcy", (B, Nq, 2)), syn.det_rand("decfull.wh", (B, Nq, 2)) * 0.5 + 0.01], -1)
    vr = torch.ones(B, 4, 2)
    vr[1] = 0.8
    mask = torch.zeros(B, Nv, dtype=torch.bool)
    mask[1, ::5] = True
    return dec.eval(), sd, (query, ref, memory, shapes, lsi, vr, mask)


def test_decoder_full_size_fp32_matches_oracle():
    """Six refinement layers with random (gain ~1) weights amplify rounding differences ~3x per layer, so the
    full-depth comparison is made two ways: (1) layer by layer with the oracle's inputs (no amplification), at the
    1e-3 bar; (2) end to end against an fp64 run of the oracle, where the HIP path's fp32 error must not exceed
    twice the fp32 oracle's own error (or 1e-3)."""
    dec, sd, args = _full_size(torch.float32)
    trace = []
    ocls, obox = R.decoder(sd, *args, 6, heads=8, trace=trace)
    sd64 = {k: v.double() for k, v in sd.items()}
    args64 = [a.double() if a.is_floating_point() else a for a in args]
    tcls, tbox = R.decoder(sd64, *args64, 6, heads=8)
    dec = dec.cuda()
    gpu = [a.cuda() for a in args]
    with torch.no_grad():
        cls, box = dec(*gpu)
        assert cls.shape == (6, 2, 900, 91) and box.shape == (6, 2, 900, 4)
        # (1) per layer, teacher-forced with the oracle's layer inputs
        for i, (q_in, ref_pts, q_pos, ref_in, q_out) in enumerate(trace):
            got = dec.layers[i](query=q_in.cuda(), query_pos=q_pos.cuda(), reference_points=ref_in.cuda(),
                                value=gpu[2], spatial_shapes=gpu[3], level_start_index=gpu[4],
                                key_padding_mask=gpu[6])
            assert (got.cpu() - q_out).abs().max() < 1e-3, i
    assert (cls[0].cpu() - ocls[0]).abs().max() < 1e-4 and (box[0].cpu() - obox[0]).abs().max() < 1e-5
    # (2) end to end against fp64
    e_ref_c, e_ref_b = (ocls.double() - tcls).abs().max(), (obox.double() - tbox).abs().max()
    e_gpu_c, e_gpu_b = (cls.cpu().double() - tcls).abs().max(), (box.cpu().double() - tbox).abs().max()
    print("fp32 oracle vs fp64: cls %.3g box %.3g; HIP vs fp64: cls %.3g box %.3g" % (e_ref_c, e_ref_b, e_gpu_c, e_gpu_b))
    assert e_gpu_c <= max(1e-3, 2 * e_ref_c) and e_gpu_b <= max(1e-4, 2 * e_ref_b)


def test_decoder_full_size_bf16_close_to_fp32():
    """bf16 weights/activations with fp16 value maps (the encoder's bench configuration).  Full-depth outputs are not
    comparable (see the fp32 test: rounding differences grow ~3x per layer with these weights), so every bf16 layer is
    fed the fp32 run's own layer inputs: outputs (LayerNorm scale, |x| ~ 1) stay within 0.4 in the worst element (a handful of outliers; the memory is white noise, so sampled values are steep in the box coordinates) and
    0.015 on average; the first layer's logits / boxes within 0.3 / 0.03."""
    dec, sd, args = _full_size(torch.float32)
    dec = dec.cuda()
    gpu = [a.cuda() for a in args]
    seen = []
    hooks = [l.register_forward_hook(lambda m, a, kw, out: seen.append((kw, out)), with_kwargs=True) for l in dec.layers]
    with torch.no_grad():
        cls32, box32 = dec(*gpu)
    for h in hooks:
        h.remove()
    dec = dec.bfloat16()
    for layer in dec.layers:
        layer.cross_attn.value_dtype = torch.float16
    mem16 = gpu[2].bfloat16()
    with torch.no_grad():
        cls, box = dec(gpu[0].bfloat16(), gpu[1], mem16, *gpu[3:])
        assert cls.dtype == torch.bfloat16 and box.dtype == torch.float32
        dc, db = (cls[0].float() - cls32[0]).abs(), (box[0] - box32[0]).abs()
        assert dc.max() < 0.3 and db.max() < 0.03, (float(dc.max()), float(db.max()))
        for i, (kw, out32) in enumerate(seen):
            got = dec.layers[i](query=kw["query"].bfloat16(), query_pos=kw["query_pos"].bfloat16(),
                                reference_points=kw["reference_points"], value=mem16,
                                spatial_shapes=kw["spatial_shapes"], level_start_index=kw["level_start_index"],
                                key_padding_mask=kw["key_padding_mask"])
            d = (got.float() - out32).abs()
            print("layer", i, float(d.max()), float(d.mean()))
            assert d.max() < 0.4 and d.mean() < 0.015, (i, float(d.max()), float(d.mean()))


def test_fp16_mode_against_fp16_operand_arithmetic():
    """BASELINE.json configs[4] asks for fp16 (the reference's --mixed-precision fp16, main.py:24-56).  Since round 5 the
    build runs that request with IEEE-half activations (libsalience_hip_f16.so).  On the configuration's own stress case --
    the 900-query decoder over the 22 000-token memory -- every layer, fed the fp32 run's inputs, is compared with the SAME
    layer evaluated on fp16-rounded parameters and fp16-rounded inputs with fp32 arithmetic (what an fp16-activation path
    carries between its kernels).  The fp16 mode stays within fp16 rounding of that (LayerNorm-scale outputs: mean <= 2.5e-3,
    max <= 0.08); the bf16 mode that used to stand in for it (rounds 2-4) is printed beside it: ~8x farther."""
    import copy
    dec, sd, args = _full_size(torch.float32)
    dec = dec.cuda()
    gpu = [a.cuda() for a in args]
    seen = []
    hooks = [l.register_forward_hook(lambda m, a, kw, out: seen.append((kw, out)), with_kwargs=True) for l in dec.layers]
    with torch.no_grad():
        dec(*gpu)
    for h in hooks:
        h.remove()
    h16 = lambda t: t.half().float() if t.is_floating_point() else t
    emu = copy.deepcopy(dec)
    with torch.no_grad():
        for p in emu.parameters():
            p.copy_(h16(p))
    served = copy.deepcopy(dec).half()
    old = copy.deepcopy(dec).bfloat16()
    for layer in list(served.layers) + list(old.layers):
        layer.cross_attn.value_dtype = torch.float16
    mem = gpu[2]
    with torch.no_grad():
        for i, (kw, out32) in enumerate(seen):
            common = dict(reference_points=kw["reference_points"], spatial_shapes=kw["spatial_shapes"],
                          level_start_index=kw["level_start_index"], key_padding_mask=kw["key_padding_mask"])
            e = emu.layers[i](query=h16(kw["query"]), query_pos=h16(kw["query_pos"]), value=h16(mem), **common)
            s = served.layers[i](query=kw["query"].half(), query_pos=kw["query_pos"].half(), value=mem.half(),
                                 **common)
            assert s.dtype == torch.float16
            s = s.float()
            o = old.layers[i](query=kw["query"].bfloat16(), query_pos=kw["query_pos"].bfloat16(), value=mem.bfloat16(),
                              **common).float()
            d_sub, d_emu, d_old = (s - e).abs(), (e - out32).abs(), (o - e).abs()
            print("layer", i, "fp16 mode vs fp16-operand: max %.4f mean %.5f;  fp16-operand vs fp32: max %.4f mean %.6f;  "
                  "bf16 stand-in of rounds 2-4 vs fp16-operand: max %.4f mean %.5f"
                  % (float(d_sub.max()), float(d_sub.mean()), float(d_emu.max()), float(d_emu.mean()),
                     float(d_old.max()), float(d_old.mean())))
            assert d_sub.max() < 0.08 and d_sub.mean() < 2.5e-3, (i, float(d_sub.max()), float(d_sub.mean()))
            assert d_sub.mean() < 0.5 * d_old.mean()    # (a real gain over the substitution it replaces)
