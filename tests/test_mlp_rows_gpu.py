"""GPU: the decoder's small MLPs in one launch (csrc/mlp_rows.hip; models/bricks/basic.py:6-26) against the same chain in
fp32 on the 16-bit parameters and rows, with the library-GEMM chain in the rows' type as the error scale.
"""
import pytest
import torch

from salience_detr_amd import filter_ops as F
from salience_detr_amd.salience_decoder import MLP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mlp(in_dim, out_dim, layers, dtype, seed):
    torch.manual_seed(seed)
    m = MLP(in_dim, 256, out_dim, layers)
    with torch.no_grad():
        for l in m.layers:
            l.bias.normal_(0, 0.2)
        m.layers[-1].weight.normal_(0, 0.05)        # (bbox heads are zero-initialised: give the last layer something to do)
    return m.to(DEV).to(dtype).eval()


def _fp32_chain(m, x):
    y = x.float()
    for i, l in enumerate(m.layers):
        y = torch.nn.functional.linear(y, l.weight.float(), l.bias.float())
        if i + 1 < len(m.layers):
            y = torch.relu(y).to(x.dtype).float()     # the hidden activations are stored in the rows' type
    return y


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("in_dim,out_dim,layers", [(512, 256, 2), (256, 256, 2), (256, 4, 3), (256, 32, 3), (256, 1, 3)])
@pytest.mark.parametrize("rows", [(1, 1), (1, 31), (3, 11), (2, 900)])
def test_mlp_rows_matches_the_fp32_chain(dtype, in_dim, out_dim, layers, rows):
    m = _mlp(in_dim, out_dim, layers, dtype, seed=in_dim + out_dim)
    g = torch.Generator().manual_seed(rows[1])
    x = (torch.randn(*rows, in_dim, generator=g) * 1.5).to(dtype).to(DEV)
    assert F.mlp_rows_applies(x, m.layers) is False             # autograd on: the differentiable chain
    with torch.no_grad():
        assert F.mlp_rows_applies(x, m.layers)
        got = m(x)
        lib = x
        for i, l in enumerate(m.layers):
            lib = l(lib)
            if i + 1 < len(m.layers):
                lib = torch.relu(lib)
    want = _fp32_chain(m, x)
    assert got.shape == want.shape and got.dtype == dtype
    err, base = (got.float() - want).abs().max().item(), (lib.float() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= max(2.0 * base, 2.0 ** -7 * scale), (err, base, scale)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_two_row_sources_equal_the_stacked_rows(dtype):
    """bbox_head on the normed and the raw queries of a decoder layer: ``forward(x, x_second)`` is bit for bit
    ``forward(stack((x, x_second)))`` -- also when a 32-row tile holds rows of both sources (2 x 900 = 56 tiles + 8 rows)."""
    m = _mlp(256, 4, 3, dtype, seed=9)
    g = torch.Generator().manual_seed(4)
    a = torch.randn(2, 900, 256, generator=g).to(dtype).to(DEV)
    b = torch.randn(2, 900, 256, generator=g).to(dtype).to(DEV)
    with torch.no_grad():
        both = m(a, b)
        stacked = m(torch.stack((a, b)))
        alone = m(b)
    assert both.shape == (2, 2, 900, 4)
    assert torch.equal(both, stacked) and torch.equal(both[1], alone)


def test_mlp_rows_refuses_what_it_does_not_take():
    m = _mlp(256, 4, 3, torch.bfloat16, seed=1)
    x = torch.randn(4, 256)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            F.mlp_rows(x.to(torch.bfloat16), m.layers)                       # CPU rows
        with pytest.raises(RuntimeError):
            F.mlp_rows(x.to(DEV), m.layers)                                  # fp32 rows
        wide = _mlp(256, 64, 3, torch.bfloat16, seed=2)
        assert not F.mlp_rows_applies(x.to(torch.bfloat16).to(DEV), wide.layers)   # 64 outputs: the library chain
        assert wide(x.to(torch.bfloat16).to(DEV)).shape == (4, 64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,pos_features", [(768, 512), (384, 384), (91, 0), (256, 32), (33, 64)])
@pytest.mark.parametrize("rows", [(1, 1), (1, 33), (2, 900)])
def test_rows_linear_matches_the_fp32_products(dtype, n, pos_features, rows):
    """One Linear whose first ``pos_features`` outputs see ``x + pos`` (the decoder's q | k | v in-projection and the
    cross-attention's offset | weight projection) against fp32 products on the rounded operands -- ``x + pos`` rounded to
    the rows' type first, as the elementwise add of the reference does."""
    torch.manual_seed(n + rows[1])
    lin = torch.nn.Linear(256, n).to(DEV).to(dtype)
    with torch.no_grad():
        lin.bias.normal_(0, 0.3)
    x = (torch.randn(*rows, 256) * 1.5).to(dtype).to(DEV)
    pos = torch.randn(*rows, 256).to(dtype).to(DEV)
    assert not F.rows_linear_applies(x, lin.weight, lin.bias)           # autograd on
    with torch.no_grad():
        assert F.rows_linear_applies(x, lin.weight, lin.bias)
        got = F.rows_linear(x, lin.weight, lin.bias, pos=pos if pos_features else None, pos_features=pos_features)
        lib = torch.cat((lin(x + pos)[..., :pos_features], lin(x)[..., pos_features:]), -1)
    w, b = lin.weight.float(), lin.bias.float()
    xs = (x + pos).float()
    want = torch.cat(((xs @ w.t() + b)[..., :pos_features], (x.float() @ w.t() + b)[..., pos_features:]), -1)
    assert got.shape == want.shape == tuple(rows) + (n,) and got.dtype == dtype
    err, base = (got.float() - want).abs().max().item(), (lib.float() - want).abs().max().item()
    assert err <= max(2.0 * base, 2.0 ** -7 * want.abs().max().item()), (err, base)


def test_rows_linear_refuses_what_it_does_not_take():
    lin = torch.nn.Linear(256, 1024).to(DEV).to(torch.bfloat16)
    x = torch.randn(4, 256, dtype=torch.bfloat16, device=DEV)
    with torch.no_grad():
        assert not F.rows_linear_applies(x, lin.weight, lin.bias)         # 1024 outputs
        with pytest.raises(RuntimeError):
            F.rows_linear(x, lin.weight, lin.bias)
        ok = torch.nn.Linear(256, 64).to(DEV).to(torch.bfloat16)
        with pytest.raises(RuntimeError):
            F.rows_linear(x, ok.weight, ok.bias, pos=x[:2], pos_features=64)   # pos of another shape
        with pytest.raises(RuntimeError):
            F.rows_linear(x.cpu(), ok.weight, ok.bias)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,ncls", [((2, 900), 91), ((1, 33), 91), ((1, 1), 7), ((2, 50), 224)])
def test_decoder_head_matches_the_module_by_module_path(dtype, rows, ncls):
    """LayerNorm + class head + both bbox chains + refinement in one launch against (a) the same stages launched one by
    one and (b) the stages in fp32 on the 16-bit parameters."""
    from salience_detr_amd.salience_decoder import inverse_sigmoid
    torch.manual_seed(ncls + rows[1])
    norm = torch.nn.LayerNorm(256).to(DEV)
    cls = torch.nn.Linear(256, ncls).to(DEV)
    with torch.no_grad():
        norm.weight.normal_(1.0, 0.2)
        norm.bias.normal_(0, 0.2)
        cls.bias.normal_(0, 0.5)
    norm, cls = norm.to(dtype), cls.to(dtype)
    bbox = _mlp(256, 4, 3, dtype, seed=5)
    q = (torch.randn(*rows, 256) * 2.0 + 0.3).to(dtype).to(DEV)
    ref = torch.rand(*rows, 4).to(DEV)
    ref[..., 0, 0] = 0.0          # a coordinate on the clamp
    ref[..., 0, 1] = 1.0
    with torch.no_grad():
        assert F.decoder_head_applies(q, norm, cls, bbox.layers)
        logits, boxes = F.decoder_head(q, norm, cls, bbox.layers, ref, True)
        only, boxes1 = F.decoder_head(q, norm, cls, bbox.layers, ref, False)
        normed = F.fused_layer_norm(q, norm)
        lib_logits = cls(normed)
        lib_boxes = F.box_refine(bbox(normed, q), ref)
    assert logits.shape == tuple(rows) + (ncls,) and boxes.shape == (2,) + tuple(rows) + (4,) and boxes1.shape[0] == 1
    assert torch.equal(only, logits) and torch.equal(boxes1[0], boxes[0])
    # fp32 on the 16-bit parameters, the normed rows rounded to the rows' type as both paths store them
    n32 = torch.nn.functional.layer_norm(q.float(), (256,), norm.weight.float(), norm.bias.float(), norm.eps).to(dtype).float()
    want_logits = n32 @ cls.weight.float().t() + cls.bias.float()
    want_boxes = torch.stack([(_fp32_chain(bbox, x.to(dtype)).to(dtype).float() + inverse_sigmoid(ref)).sigmoid()
                              for x in (n32, q.float())])
    err, base = (logits.float() - want_logits).abs().max().item(), (lib_logits.float() - want_logits).abs().max().item()
    assert err <= max(2.0 * base, 2.0 ** -6 * want_logits.abs().max().item()), (err, base)
    berr, bbase = (boxes - want_boxes).abs().max().item(), (lib_boxes - want_boxes).abs().max().item()
    assert berr <= max(2.0 * bbase, 2e-3), (berr, bbase)
    assert (boxes - lib_boxes).abs().max().item() <= 4e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Nq", [(2, 900), (1, 33), (3, 1)])
def test_ref_point_head_equals_sine_embed_then_chain(dtype, B, Nq):
    """Sine embedding made in the chain's tile fill: bit for bit the two launches it replaces (the same arithmetic on the
    same rounded features), and the scaled reference points with it."""
    m = _mlp(512, 256, 2, dtype, seed=3)
    g = torch.Generator().manual_seed(B * 1000 + Nq)
    ref = torch.rand(B, Nq, 4, generator=g).to(DEV)
    vr = (torch.rand(B, 4, 2, generator=g) * 0.4 + 0.6).to(DEV)
    with torch.no_grad():
        assert F.ref_point_head_applies(m.layers, dtype, 128)
        ref_in, pos = F.ref_point_head(ref, vr, m.layers, dtype)
        want_in, sine = F.decoder_query_sine_embed(ref, vr, 128, dtype)
        want = F.mlp_rows(sine, m.layers)
    assert torch.equal(ref_in, want_in)
    assert pos.shape == (B, Nq, 256) and torch.equal(pos, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows", [(1, 1), (1, 33), (2, 900)])
def test_rows_linear_ln_matches_gemm_then_add_layernorm(dtype, rows):
    """``norm(residual + linear(x))`` in one launch against the two launches it replaces (library GEMM, fused add +
    LayerNorm) and against fp32 on the 16-bit parameters with the Linear's output rounded as both paths store it."""
    torch.manual_seed(rows[1])
    lin = torch.nn.Linear(256, 256).to(DEV)
    norm = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        lin.bias.normal_(0, 0.3)
        norm.weight.normal_(1.0, 0.2)
        norm.bias.normal_(0, 0.2)
    lin, norm = lin.to(dtype), norm.to(dtype)
    x = (torch.randn(*rows, 256) * 1.5).to(dtype).to(DEV)
    res = (torch.randn(*rows, 256) * 1.5).to(dtype).to(DEV)
    assert not F.rows_linear_ln_applies(x, lin, norm)                 # autograd on
    with torch.no_grad():
        assert F.rows_linear_ln_applies(x, lin, norm)
        got = F.rows_linear_ln(x, lin, norm, res)
        lib = F.fused_layer_norm(res, norm, residual=lin(x))
    y = (x.float() @ lin.weight.float().t() + lin.bias.float()).to(dtype).float() + res.float()
    want = torch.nn.functional.layer_norm(y, (256,), norm.weight.float(), norm.bias.float(), norm.eps)
    assert got.shape == x.shape and got.dtype == dtype
    err, base = (got.float() - want).abs().max().item(), (lib.float() - want).abs().max().item()
    assert err <= max(2.0 * base, 2.0 ** -6 * want.abs().max().item()), (err, base)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            F.rows_linear_ln(x, lin, norm, res[..., :128])
