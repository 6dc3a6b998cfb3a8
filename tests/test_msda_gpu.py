"""GPU parity of the MSDA kernels (through the C ABI) against the oracle and the golden vectors.

Bars: fp64 1e-10; fp32 forward 1e-4 abs (north_star: <= 1e-3 fp32); bf16-stored value compared
against the oracle fed the SAME bf16-rounded value (fp32 accumulate) at 1e-4.
"""
import os

import numpy as np
import pytest
import torch

from oracle import msda_c
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _t(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dt) if dt is not None else t


@pytest.fixture(scope="module")
def op_cases():
    return np.load(os.path.join(G, "msda_op_cases.npz"))


@pytest.mark.parametrize("name", ["tiny", "degenerate", "hotlike", "odd_d"])
@pytest.mark.parametrize("tag,dt,tol", [("f64", torch.float64, 1e-10), ("f32", torch.float32, 1e-4)])
def test_reference_layout_op_golden(op_cases, name, tag, dt, tol):
    d = op_cases
    v, loc, aw, go = (_t(d[f"{name}.{k}"], dt).to(DEV) for k in ("value", "loc", "aw", "gout"))
    shapes, lsi = _t(d[f"{name}.shapes"]).to(DEV), _t(d[f"{name}.lsi"]).to(DEV)
    out = M.ms_deform_attn_forward(v, shapes, lsi, loc, aw, 64)
    gv, gl, ga = M.ms_deform_attn_backward(v, shapes, lsi, loc, aw, go, 64)
    torch.cuda.synchronize()
    for got, key in ((out, "out"), (gv, "gv"), (gl, "gl"), (ga, "ga")):
        ref = _t(d[f"{name}.{key}_{tag}"])
        err = (got.cpu() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (key, err)


def test_autograd_function_matches_golden(op_cases):
    d, name = op_cases, "hotlike"
    v, loc, aw = (_t(d[f"{name}.{k}"]).to(DEV).requires_grad_(True) for k in ("value", "loc", "aw"))
    shapes, lsi = _t(d[f"{name}.shapes"]).to(DEV), _t(d[f"{name}.lsi"]).to(DEV)
    out = M.MultiScaleDeformableAttnFunction.apply(v, shapes, lsi, loc, aw, 64)
    out.backward(_t(d[f"{name}.gout"]).to(DEV))
    assert (v.grad.cpu() - _t(d[f"{name}.gv_f32"])).abs().max() < 1e-4
    assert (loc.grad.cpu() - _t(d[f"{name}.gl_f32"])).abs().max() < 1e-3
    assert (aw.grad.cpu() - _t(d[f"{name}.ga_f32"])).abs().max() < 1e-4


LEVELS_SMALL = [(20, 30), (10, 15), (5, 8), (3, 4)]
LEVELS_FULL = [(100, 168), (50, 84), (25, 42), (13, 21)]


@pytest.mark.parametrize("B,Nq,levels,M_,D,P", [
    (2, 333, LEVELS_SMALL, 8, 32, 4),
    (1, 77, LEVELS_SMALL, 4, 16, 3),
    (3, 50, LEVELS_SMALL[:2], 2, 64, 5),   # L*P = 10
    (1, 40, LEVELS_SMALL, 2, 8, 9),        # L*P = 36 > one LDS chunk
    (2, 2272, LEVELS_FULL, 8, 32, 4),      # encoder layer 5 at the benchmark shape
    (2, 11363, LEVELS_FULL, 8, 32, 4),     # encoder layer 0 at the benchmark shape: the largest call of the step
])
def test_forward_backward_vs_c_oracle(B, Nq, levels, M_, D, P):
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, levels, M_, D, P, seed=1, spread_px=6.0)
    go = syn.det_randn("gout", (B, Nq, M_ * D))
    ref = msda_c.msda_forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy())
    rgv, rgl, rga = msda_c.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(),
                                         go.numpy())
    dv, dloc, daw = value.to(DEV), loc.to(DEV), aw.to(DEV)
    out = M.ms_deform_attn_forward(dv, shapes.to(DEV), lsi.to(DEV), dloc, daw, 64)
    gv, gl, ga = M.ms_deform_attn_backward(dv, shapes.to(DEV), lsi.to(DEV), dloc, daw, go.to(DEV), 64)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4
    assert np.abs(gv.cpu().numpy() - rgv).max() < 2e-4 * max(1.0, np.abs(rgv).max())
    # d(out)/d(loc) jumps where a sampling point sits exactly on a pixel boundary; a 1-ulp difference
    # in loc*size-0.5 (fma contraction on the GPU) flips floor() there.  Exclude those samples.
    px = loc * torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()[None, None, None, :, None, :] - 0.5
    smooth = ((px - px.round()).abs() > 1e-3).all(-1, keepdim=True).expand_as(loc).numpy()
    assert smooth.mean() > 0.99
    assert np.abs((gl.cpu().numpy() - rgl) * smooth).max() < 2e-4 * max(1.0, np.abs(rgl).max())
    assert np.abs(ga.cpu().numpy() - rga).max() < 2e-4 * max(1.0, np.abs(rga).max())


@pytest.mark.parametrize("vdt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Nq,levels,M_,D,P", [(2, 301, LEVELS_SMALL, 8, 32, 4), (1, 65, LEVELS_SMALL, 4, 16, 2)])
def test_head_major_explicit_vs_oracle(vdt, B, Nq, levels, M_, D, P):
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, levels, M_, D, P, seed=2, spread_px=5.0)
    vq = value.to(vdt).float()  # what the kernel actually reads
    ref = msda_c.msda_forward(vq.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy())
    Nv = value.shape[1]
    hm = M.value_to_head_major(value.view(B, Nv, M_ * D).to(DEV), None, M_, vdt)
    assert hm.shape == (B, M_, Nv, D) and hm.dtype == vdt
    assert torch.equal(hm.float().cpu(), vq.permute(0, 2, 1, 3))
    out = M.msda_forward_head_major(hm, shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV))
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4


def _fused_reference(value_q, shapes, lsi, ref_pts, proj, M_, L, P):
    B, Nq = proj.shape[:2]
    off = proj[..., :M_ * L * P * 2].view(B, Nq, M_, L, P, 2)
    aw = proj[..., M_ * L * P * 2:M_ * L * P * 3].view(B, Nq, M_, L * P).softmax(-1).view(B, Nq, M_, L, P)
    if ref_pts.shape[-1] == 2:
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
        loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref_pts[:, :, None, :, None, :2] + off / P * ref_pts[:, :, None, :, None, 2:] * 0.5
    return msda_c.msda_forward(value_q.numpy(), shapes.numpy(), lsi.numpy(), loc.contiguous().numpy(),
                               aw.contiguous().numpy())


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("vdt,pdt", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                     (torch.bfloat16, torch.bfloat16), (torch.float16, torch.bfloat16)])
def test_fused_forward_vs_oracle(ref_dim, vdt, pdt):
    B, Nq, M_, D, P, levels = 2, 257, 8, 32, 4, LEVELS_SMALL
    L = len(levels)
    value, shapes, lsi, _, _ = syn.make_msda_inputs(B, Nq, levels, M_, D, P, seed=3)
    Nv = value.shape[1]
    proj = torch.cat([syn.det_randn("off", (B, Nq, M_ * L * P * 2)) * 3.0,
                      syn.det_randn("lgt", (B, Nq, M_ * L * P)),
                      syn.det_randn("pad", (B, Nq, 5))], -1).to(pdt)   # row stride > 3*M*L*P
    if ref_dim == 2:
        ref_pts = syn.det_rand("ref", (B, Nq, L, 2)) * 1.1 - 0.05
    else:
        ref_pts = torch.cat([syn.det_rand("ref", (B, Nq, L, 2)), syn.det_rand("wh", (B, Nq, L, 2)) * 0.4 + 0.02], -1)
    vq = value.to(vdt).float()
    expect = _fused_reference(vq, shapes, lsi, ref_pts, proj.float(), M_, L, P)
    hm = M.value_to_head_major(value.view(B, Nv, M_ * D).to(DEV), None, M_, vdt)
    order = torch.stack([torch.randperm(Nq, generator=torch.Generator().manual_seed(b)) for b in range(B)]).int()
    for ordr in (None, order.to(DEV)):
        out = M.msda_fused_forward(hm, shapes.to(DEV), lsi.to(DEV), ref_pts.to(DEV), proj.to(DEV), L, P,
                                   order=ordr, out_dtype=torch.float32)
        assert np.abs(out.cpu().numpy() - expect).max() < 2e-4


def test_value_to_head_major_mask_and_stride():
    B, Nv, M_, D = 2, 130, 8, 32
    wide = syn.det_randn("wide", (B, Nv, 3 * M_ * D)).to(DEV)
    mask = torch.zeros(B, Nv, dtype=torch.bool)
    mask[1, 100:] = True
    mask[0, 3] = True
    sl = wide[:, :, M_ * D:2 * M_ * D]  # column slice: row stride 3*E
    hm = M.value_to_head_major(sl, mask.to(DEV), M_, torch.bfloat16)
    expect = sl.cpu().masked_fill(mask[..., None], 0.0).view(B, Nv, M_, D).permute(0, 2, 1, 3).to(torch.bfloat16)
    assert torch.equal(hm.cpu(), expect)
    # grouped (all encoder layers in one launch) + fp16 storage with saturation
    big = wide.clone()
    big[0, 0, 0] = 1e6
    hm3 = M.value_to_head_major(big, mask.to(DEV), M_, torch.float16, num_groups=3)
    assert hm3.shape == (3, B, M_, Nv, D) and hm3.dtype == torch.float16
    ref3 = big.cpu().masked_fill(mask[..., None], 0.0).clamp(-65504, 65504).view(B, Nv, 3, M_, D).permute(2, 0, 3, 1, 4)
    assert torch.equal(hm3.cpu(), ref3.to(torch.float16))


def test_full_size_properties():
    """Benchmark-shape invariants that need no oracle: linearity in value and partition of unity."""
    B, Nq, M_, D, P, levels = 2, 11363, 8, 32, 4, LEVELS_FULL
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, levels, M_, D, P, seed=4)
    sh, ls, dl, da = shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV)
    v1 = value.to(DEV)
    v2 = syn.det_randn("v2", tuple(value.shape)).to(DEV)
    o1 = M.ms_deform_attn_forward(v1, sh, ls, dl, da, 64)
    o2 = M.ms_deform_attn_forward(v2, sh, ls, dl, da, 64)
    o12 = M.ms_deform_attn_forward(v1 + 2.0 * v2, sh, ls, dl, da, 64)
    assert (o12 - (o1 + 2.0 * o2)).abs().max() < 1e-4
    # constant value map + all sampling points strictly inside -> output == constant (weights sum to 1)
    inner = loc.clamp(0.04, 0.96).to(DEV)  # 0.04*13-0.5 >= 0 on the coarsest (13x21) level
    oc = M.ms_deform_attn_forward(torch.full_like(v1, 3.0), sh, ls, inner, da, 64)
    assert (oc - 3.0).abs().max() < 1e-5
    # head-major bf16 fused path agrees with the reference-layout op on the same (bf16-rounded) value
    Nv = value.shape[1]
    hm = M.value_to_head_major(v1.view(B, Nv, M_ * D), None, M_, torch.bfloat16)
    ref = M.ms_deform_attn_forward(hm.float().permute(0, 2, 1, 3).contiguous(), sh, ls, dl, da, 64)
    out = M.msda_forward_head_major(hm, sh, ls, dl, da)
    assert (out - ref).abs().max() < 1e-4


def test_error_behaviour():
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(2, 9, LEVELS_SMALL, 8, 32, 4, seed=5)
    with pytest.raises(RuntimeError):  # CPU tensors: no fallback
        M.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
    dv, ds, dl, dloc, daw = (t.to(DEV) for t in (value, shapes, lsi, loc, aw))
    with pytest.raises(RuntimeError):  # non-contiguous
        M.ms_deform_attn_forward(dv.transpose(0, 1).contiguous().transpose(0, 1), ds, dl, dloc, daw, 64)
    three = syn.make_msda_inputs(3, 9, LEVELS_SMALL, 8, 32, 4, seed=5)
    with pytest.raises(RuntimeError):  # batch 3 not divisible by min(3, 2)
        M.ms_deform_attn_forward(*(t.to(DEV) for t in three), 2)
    with pytest.raises(RuntimeError):
        M.ms_deform_attn_forward(dv.half(), ds, dl, dloc.half(), daw.half(), 64)
    mod = M.MultiScaleDeformableAttention(256, 4, 8, 4).to(DEV)
    q = torch.zeros(2, 9, 256, device=DEV)
    val = torch.zeros(2, int(shapes.prod(1).sum()), 256, device=DEV)
    with pytest.raises(ValueError):
        mod(q, torch.zeros(2, 9, 4, 3, device=DEV), val, ds, dl, None)
    with pytest.raises(RuntimeError):
        mod(q.cpu(), torch.zeros(2, 9, 4, 2), val.cpu(), shapes, lsi, None)


def test_module_golden_both_paths():
    d = np.load(os.path.join(G, "msda_module_cases.npz"))
    E, Lv, H, P = d["dims"].tolist()
    mod = M.MultiScaleDeformableAttention(E, Lv, H, P)
    mod.load_state_dict({k[3:]: _t(d[k]) for k in d.files if k.startswith("sd.")})
    mod = mod.to(DEV).eval()
    q, val, mask = _t(d["query"]).to(DEV), _t(d["value"]).to(DEV), _t(d["mask"]).to(DEV)
    shapes, lsi = _t(d["shapes"]).to(DEV), _t(d["lsi"]).to(DEV)
    for ref_key, out_key, m in (("ref2", "out2", mask), ("ref4", "out4", mask), ("ref2", "out2_nomask", None)):
        ref_pts = _t(d[ref_key]).to(DEV)
        with torch.no_grad():
            native = mod(q, ref_pts, val, shapes, lsi, m)
        autograd = mod(q.clone().requires_grad_(True), ref_pts, val, shapes, lsi, m)
        for got in (native, autograd):
            assert (got.detach().cpu() - _t(d[out_key])).abs().max() < 1e-4, (ref_key, out_key)


def _value_hm_bf16(B, levels, M_=8, D=32, seed=9):
    Nv = sum(h * w for h, w in levels)
    value = syn.det_randn("tiled.value", (B, Nv, M_ * D), salt=seed)
    hm = M.value_to_head_major(value.to(DEV), None, M_, torch.bfloat16)
    return value.to(torch.bfloat16).float().view(B, Nv, M_, D), hm


def test_head_major_projection_layout_equals_token_rows():
    """The fused kernel fed per-head projection slabs [B,M,Nq,48] (written directly by the token-resident linear
    kernel) returns what it returns for the same numbers in token rows [B,Nq,384]."""
    from salience_detr_amd import filter_ops as FO
    from salience_detr_amd import ms_deform_attn as M_
    torch.manual_seed(3)
    levels = [(25, 42), (13, 21), (7, 11), (4, 6)]
    B, Nq, M, D, L, P = 2, 777, 8, 32, 4, 4
    Nv = sum(h * w for h, w in levels)
    shapes = torch.tensor(levels, dtype=torch.int64, device=DEV)
    lsi = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    value_hm = (torch.randn(B, M, Nv, D, device=DEV)).to(torch.float16)
    ref = torch.rand(B, Nq, L, 2, device=DEV)
    x = torch.randn(B, Nq, 256, device=DEV).to(torch.bfloat16)
    pos = torch.randn(B, Nq, 256, device=DEV).to(torch.bfloat16)
    attn = M_.MultiScaleDeformableAttention(256, L, M, P).to(DEV).to(torch.bfloat16)
    attn.sampling_offsets.weight.data.normal_(0, 0.02)
    attn.attention_weights.weight.data.normal_(0, 0.05)
    with torch.no_grad():
        w, b = attn._fused_query_projection()
        wh, bh = attn._fused_query_projection_head_major()
        rows = FO.token_linear(x, w, b, x_add=pos)                                   # [B,Nq,384]
        slabs = FO.token_linear(x, wh, bh, x_add=pos, group_features=48)             # [B,M,Nq,48]
        # same numbers, two layouts
        off = rows[..., :256].view(B, Nq, M, 32).permute(0, 2, 1, 3)
        logit = rows[..., 256:].view(B, Nq, M, 16).permute(0, 2, 1, 3)
        assert torch.equal(slabs, torch.cat([off, logit], -1))
        o_rows = M_.msda_fused_forward(value_hm, shapes, lsi, ref, rows, L, P, out_dtype=torch.bfloat16)
        o_hm = M_.msda_fused_forward(value_hm, shapes, lsi, ref, slabs, L, P, out_dtype=torch.bfloat16, proj_head_major=True)
    assert torch.equal(o_rows, o_hm)


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("B,Nq", [(2, 300), (1, 11363), (3, 1)])
def test_sampling_prep_op_matches_the_torch_expressions(B, Nq, ref_dim):
    """softmax + offset normalisation + reference-point add as one launch each way (csrc/sampling_prep.hip) against the
    module's torch formulation (ms_deform_attn.py:322-349), values and gradients."""
    from salience_detr_amd.ms_deform_attn import _SamplingPrep
    M, L, P = 8, 4, 4
    shapes = torch.tensor([[100, 168], [50, 84], [25, 42], [13, 21]], device="cuda")
    off = (syn.det_randn(f"sp.off{Nq}", (B, Nq, M, L, P, 2)) * 3).cuda()
    lg = (syn.det_randn(f"sp.lg{Nq}", (B, Nq, M, L * P)) * 2).cuda()
    ref = syn.det_rand(f"sp.ref{Nq}", (B, Nq, L, ref_dim)).cuda()
    gl = syn.det_randn(f"sp.gl{Nq}", (B, Nq, M, L, P, 2)).cuda()
    gw = syn.det_randn(f"sp.gw{Nq}", (B, Nq, M, L, P)).cuda()

    def torch_form(o, g):
        w = g.softmax(-1).view(B, Nq, M, L, P)
        if ref_dim == 2:
            norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
            loc = ref[:, :, None, :, None, :] + o / norm[None, None, None, :, None, :]
        else:
            loc = ref[:, :, None, :, None, :2] + o / P * ref[:, :, None, :, None, 2:] * 0.5
        return loc, w

    o1, g1 = off.clone().requires_grad_(True), lg.clone().requires_grad_(True)
    loc1, w1 = torch_form(o1, g1)
    ((loc1 * gl).sum() + (w1 * gw).sum()).backward()
    o2, g2 = off.clone().requires_grad_(True), lg.clone().requires_grad_(True)
    assert _SamplingPrep.applies(o2, g2, ref, L, P)
    loc2, w2 = _SamplingPrep.apply(o2, g2, ref, shapes, L, P)
    ((loc2 * gl).sum() + (w2 * gw).sum()).backward()
    assert (loc2 - loc1).abs().max() <= 1e-6 * max(1.0, loc1.abs().max().item())
    assert (w2 - w1).abs().max() <= 5e-7
    assert (o2.grad - o1.grad).abs().max() <= 1e-6 * max(1.0, o1.grad.abs().max().item())
    assert (g2.grad - g1.grad).abs().max() <= 2e-6 * max(1.0, g1.grad.abs().max().item())

