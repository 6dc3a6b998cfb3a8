"""The fp16-activation flavour of the library (round 5): ``libsalience_hip_f16.so`` = the same sources built with
``-DSDETR_ACT_F16`` (csrc/common.h), selected by ``_hip.lib(torch.float16)`` when an operator is handed IEEE-half
activations -- BASELINE.json configs[4], the reference's ``--mixed-precision fp16`` (main.py:24-56).

Every operator of the encoder / decoder layer is run here on the SAME inputs in both 16-bit activation types and held
against the fp32 evaluation of the block on the rounded operands.  fp16 carries three more mantissa bits than bf16, so the
fp16 run has to be the CLOSER one by a clear margin -- a wrong conversion anywhere (a bf16 shift applied to a half, an
MFMA of the other type) gives errors of order one.  The end-to-end bars are tests/test_encoder_timed_mode_gpu.py (against
the reference's own fp16 autocast) and tests/test_decoder_gpu.py::test_fp16_mode_against_fp16_operand_arithmetic.
"""
import numpy as np
import pytest
import torch

from salience_detr_amd import _hip, filter_ops as F, ms_deform_attn as M
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, BF = torch.float16, torch.bfloat16


def _both(fn):
    """fn(dtype) -> (got fp32, reference fp32): returns the two max / mean errors (fp16, bf16)."""
    out = {}
    for dt in (H, BF):
        got, ref = fn(dt)
        d = (got.float().cpu() - ref.float().cpu()).abs()
        out[dt] = (d.max().item(), d.mean().item(), ref.float().abs().max().item())
    return out


def _closer(out, ratio=0.4):
    (hmax, hmean, scale), (bmax, bmean, _) = out[H], out[BF]
    assert hmean <= ratio * bmean + 1e-7, out
    assert hmax <= 8e-3 * (scale + 1), out            # fp16 rounding of O(1) rows: 2^-11 relative
    return out


def test_both_libraries_load_and_are_distinct():
    a, b = _hip.lib(), _hip.lib(torch.float16)
    assert a is not b and a is _hip.lib(torch.bfloat16) and a is _hip.lib(torch.float32)
    assert a.sdetr_abi_version() == b.sdetr_abi_version() == 1
    x = torch.zeros(2, 4, 256, dtype=H, device=DEV)
    assert _hip.lib(x) is b


@pytest.mark.parametrize("B,n", [(2, 31), (2, 4545)])
def test_token_linear_and_class_head(B, n):
    def run(dt):
        x = (syn.det_randn(f"h.tlx{n}", (B, n, 256)) * 1.3).to(dt).to(DEV)
        pos = syn.det_randn(f"h.tlp{n}", (B, n, 256)).to(dt).to(DEV)
        w = (syn.det_randn("h.tlw", (384, 256)) * 0.06).to(dt).to(DEV)
        b = syn.det_randn("h.tlb", (384,)).to(dt).to(DEV)
        with torch.no_grad():
            got = F.token_linear(x, w, b, x_add=pos)
            assert got.dtype == dt
            ref = torch.nn.functional.linear((x + pos).float(), w.float(), b.float())
        return got, ref
    _closer(_both(run))

    def cls(dt):
        x = (syn.det_randn(f"h.clx{n}", (B, n, 256)) * 1.3).to(dt).to(DEV)
        head = torch.nn.Linear(256, 91).to(DEV).to(dt)
        fg = syn.det_randn(f"h.fg{n}", (B, n)).to(DEV)
        with torch.no_grad():
            got = F.class_head_max_times(x, head, fg)
            ref = torch.nn.functional.linear(x.float(), head.weight.float(), head.bias.float()).max(-1)[0] * fg
        return got, ref
    out = _both(cls)           # fp32 accumulators in both flavours: exact up to the summation order
    assert out[H][0] <= 2e-4 * (out[H][2] + 1) and out[BF][0] <= 2e-4 * (out[BF][2] + 1), out


@pytest.mark.parametrize("T,splits", [(8000, 1), (4545, 3)])
def test_fused_ffn(T, splits):
    def run(dt):
        torch.manual_seed(T)
        lin1, lin2, norm = torch.nn.Linear(256, 2048), torch.nn.Linear(2048, 256), torch.nn.LayerNorm(256)
        lin1.bias.data.normal_(0, 0.5)
        lin2.bias.data.normal_(0, 0.5)
        norm.weight.data = 1 + 0.3 * syn.det_randn("h.fg", (256,))
        norm.bias.data = 0.3 * syn.det_randn("h.fb", (256,))
        mods = [m.to(DEV).to(dt) for m in (lin1, lin2, norm)]
        x = (syn.det_randn(f"h.fx{T}", (T, 256)) * 1.5).to(dt).to(DEV)
        with torch.no_grad():
            assert F.fused_ffn_applies(x, mods[0], mods[1], mods[2], torch.nn.ReLU())
            got = F.fused_ffn(x, *mods, hidden_splits=splits)
            assert got.dtype == dt
            xf = x.float()
            h = torch.relu(torch.nn.functional.linear(xf, mods[0].weight.float(), mods[0].bias.float()))
            y = xf + torch.nn.functional.linear(h, mods[1].weight.float(), mods[1].bias.float())
            ref = torch.nn.functional.layer_norm(y, (256,), mods[2].weight.float(), mods[2].bias.float(), mods[2].eps)
        return got, ref
    _closer(_both(run))


@pytest.mark.parametrize("rows", [2272, 11363])
def test_layer_end_operator_and_row_bookkeeping(rows):
    """``attn_tail_ffn_advance`` (output_proj + norm1 + feed-forward + norm2 + advance_rows [+ next class score])."""
    B, S, n0 = 2, 22323, 11363
    nxt = max(1, rows * 4 // 5)

    def run(dt):
        torch.manual_seed(rows)
        mk = lambda m: m.to(DEV).to(dt)
        wo, n1 = mk(torch.nn.Linear(256, 256)), mk(torch.nn.LayerNorm(256))
        l1, l2, n2 = mk(torch.nn.Linear(256, 2048)), mk(torch.nn.Linear(2048, 256)), mk(torch.nn.LayerNorm(256))
        head = mk(torch.nn.Linear(256, 91))
        sampled = (syn.det_randn(f"h.s{rows}", (B, rows, 256)) * 0.8).to(DEV).to(dt)
        query = (syn.det_randn(f"h.q{rows}", (B, rows, 256)) * 0.9).to(DEV).to(dt)
        tokens = syn.det_randn("h.tok", (B, S, 256)).to(DEV).to(dt)
        sidx = torch.stack([torch.randperm(S)[:n0] for _ in range(B)]).to(DEV)
        count = torch.tensor([rows, rows - 7], device=DEV)
        fg = syn.det_randn(f"h.fg{rows}", (B, n0)).to(DEV)
        res = torch.full((B, n0, 256), -3.0, dtype=dt, device=DEV)
        with torch.no_grad():
            out = F.attn_tail_ffn_advance(sampled, query, wo, n1, l1, l2, n2, res, nxt, tokens, sidx, count,
                                          next_class_head=head, foreground=fg)
            nq, score = out
            x = torch.nn.functional.layer_norm(query.float() + torch.nn.functional.linear(sampled.float(), wo.weight.float(), wo.bias.float()),
                                               (256,), n1.weight.float(), n1.bias.float(), n1.eps)
            h = torch.relu(torch.nn.functional.linear(x, l1.weight.float(), l1.bias.float()))
            y = torch.nn.functional.layer_norm(x + torch.nn.functional.linear(h, l2.weight.float(), l2.bias.float()),
                                               (256,), n2.weight.float(), n2.bias.float(), n2.eps)
            want = torch.stack([torch.where((torch.arange(nxt, device=DEV) < count[b])[:, None], y[b, :nxt],
                                            tokens[b, sidx[b, :nxt]].float()) for b in range(B)])
            # the live rows were recorded, the others left alone
            for b in range(B):
                c = int(count[b])
                assert (res[b, :c].float() - y[b, :c]).abs().max().item() < 0.05
                assert (res[b, c:rows] == -3.0).all()
            if score is not None:
                logits = torch.nn.functional.linear(nq.float(), head.weight.float(), head.bias.float())
                assert (score - logits.max(-1)[0] * fg[:, :nxt]).abs().max().item() < 2e-3
        return nq, want
    _closer(_both(run), ratio=0.5)


def test_top300_self_attention_with_carried_projection():
    B, rows, N = 2, 6817, 300

    def run(dt):
        torch.manual_seed(3)
        mha = torch.nn.MultiheadAttention(256, 8, batch_first=True).to(DEV).to(dt)
        norm = torch.nn.LayerNorm(256).to(DEV).to(dt)
        q0 = (syn.det_randn("h.tpq", (B, rows, 256)) * 0.8).to(DEV).to(dt)
        pos = (syn.det_randn("h.tpp", (B, rows + 50, 256)) * 0.5).to(DEV).to(dt)
        sel = torch.stack([torch.randperm(rows)[:N] for _ in range(B)]).to(DEV)
        w = (syn.det_randn("h.tpw", (384, 256)) * 0.06).to(DEV).to(dt)
        b = (syn.det_randn("h.tpb", (384,)) * 0.2).to(DEV).to(dt)
        q = q0.clone()
        with torch.no_grad():
            slab = F.topk_self_attention_(q, pos, sel, mha, norm, projection=(w, b))
            # fp32 evaluation on the rounded operands
            x = torch.gather(q0, 1, sel[..., None].expand(-1, -1, 256)).float()
            p = torch.gather(pos[:, :rows], 1, sel[..., None].expand(-1, -1, 256)).float()
            m32 = torch.nn.MultiheadAttention(256, 8, batch_first=True).to(DEV)
            m32.load_state_dict({k: v.float() for k, v in mha.state_dict().items()})
            a = m32(x + p, x + p, x)[0]
            y = torch.nn.functional.layer_norm(x + a, (256,), norm.weight.float(), norm.bias.float(), norm.eps)
            want = q0.float().scatter(1, sel[..., None].expand(-1, -1, 256), y)
            if slab is not None:
                assert slab.dtype == dt and slab.shape == (B, 8, rows, 48)
                proj = torch.nn.functional.linear(want + pos[:, :rows].float(), w.float(), b.float())   # [B,rows,384]
                ref_slab = proj.view(B, rows, 8, 48).permute(0, 2, 1, 3)
                assert (slab.float() - ref_slab).abs().max().item() < (0.02 if dt == H else 0.15) * (ref_slab.abs().max().item() + 1)
        return q, want
    _closer(_both(run), ratio=0.5)


def test_layer_norm_finalize_and_flatten_in_half():
    B, n = 2, 777
    x = syn.det_randn("h.lnx", (B, n, 256)).to(DEV)
    norm = torch.nn.LayerNorm(256).to(DEV)
    norm.weight.data = 1 + 0.3 * syn.det_randn("h.lng", (256,)).to(DEV)
    with torch.no_grad():
        ref = norm(x)
        for dt in (H, BF):
            got = F.fused_layer_norm(x.to(dt), norm)
            assert got.dtype == dt
            tol = 4e-3 if dt == H else 3e-2
            assert (got.float() - norm(x.to(dt).float())).abs().max().item() < tol
        got = F.fused_layer_norm(x, norm, out_dtype=H)
        assert got.dtype == H and (got.float() - ref).abs().max().item() < 4e-3
    # F0: the 16-bit copies of the flattened pyramid in the requested activation type
    _, masks = syn.make_masks([(64, 96), (64, 80)])
    shapes = [tuple(m.shape[-2:]) for m in masks]
    feats = [f.to(DEV) for f in syn.make_feats(2, shapes, 256, 1)]
    pos = [syn.sine_position_embedding(m, 128).to(DEV) for m in masks]
    le = syn.det_randn("h.le", (4, 256)).to(DEV)
    out32 = F.pyramid_flatten(feats, pos, [m.to(DEV) for m in masks], le)
    for dt in (H, BF):
        out = F.pyramid_flatten(feats, pos, [m.to(DEV) for m in masks], le, want_bf16=dt)
        assert out[4].dtype == dt and torch.equal(out[4], out32[0].to(dt)) and torch.equal(out[5], out32[1].to(dt))


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_bordered_msda_with_half_projection_slab(ref_dim):
    from tests.test_msda_timed_kernels_gpu import LEVELS_FULL, _case, _expected, _head_major_slab, HEADS, D
    B, Nq = 2, 2272
    value, shapes, lsi, proj, ref = _case(B, Nq, LEVELS_FULL, ref_dim, seed=77)
    hm = M.value_to_head_major(value.view(B, -1, HEADS * D).to(DEV), None, HEADS, torch.float16)
    hb = M.to_bordered(hm, LEVELS_FULL)
    for dt in (H, BF):
        p16 = proj.float().to(dt)
        expect = _expected(value.to(torch.float16).float(), shapes, lsi, ref, p16.float())
        slab = _head_major_slab(p16).to(DEV)
        assert slab.dtype == dt
        out = M.msda_bordered_forward(hb, LEVELS_FULL, ref.to(DEV), slab, out_dtype=torch.float32)
        assert M.last_forward_kernel(dt) == M.KERNEL_BORDERED
        assert np.abs(out.cpu().numpy() - expect).max() < 2e-4
        out16 = M.msda_bordered_forward(hb, LEVELS_FULL, ref.to(DEV), slab, out_dtype=dt)
        assert out16.dtype == dt
        # (fp16 outputs: exact products + one fp16 rounding; bf16 outputs: the packed-fp16 corner sums of PK = 2, bars of
        # tests/test_msda_bordered_gpu.py)
        tol, slack = (2.0 ** -11, 1e-3) if dt == H else (2.0 ** -8, 4e-3)
        assert (np.abs(out16.float().cpu().numpy() - expect) <= np.abs(expect) * tol + slack).all()


def test_value_projection_from_half_tokens():
    B, Nv, groups, heads = 2, 3000, 2, 8
    pad = (torch.rand(B, Nv) < 0.1).to(DEV)
    for dt in (H, BF):
        tokens = syn.det_randn("h.vtok", (B, Nv, 256)).to(DEV).to(dt)
        w = (syn.det_randn("h.vw", (groups * heads * 32, 256)) * 0.05).to(DEV).to(dt)
        bias = (syn.det_randn("h.vb", (groups * heads * 32,)) * 0.1).to(DEV).to(dt)
        with torch.no_grad():
            maps = F.value_proj_head_major(tokens, w, bias, pad, heads, groups, torch.float16)
            v = torch.nn.functional.linear(tokens.float(), w.float(), bias.float()).masked_fill(pad[..., None], 0.0)
            want = v.view(B, Nv, groups, heads, 32).permute(2, 0, 3, 1, 4)
        assert maps.dtype == torch.float16 and maps.shape == (groups, B, heads, Nv, 32)
        assert (maps.float() - want).abs().max().item() < 2e-3 * (want.abs().max().item() + 1)
