"""GPU: bit-exact parity of the masked top-k / sort and row movers against the oracle's rule
(stable descending sort = ties to the lower index), incl. ties, masks, ragged and maximum sizes."""
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import filter_ops as F
from salience_detr_amd import pyramid
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_topk(score, k, mask=None):
    if mask is not None:
        score = score.masked_fill(mask, score.min())
    return R.topk_desc_stable(score, k)


@pytest.mark.parametrize("B,N,k", [(1, 1, 1), (2, 7, 3), (2, 273, 273), (2, 1050, 1050), (2, 4200, 3360),
                                   (2, 16800, 6680), (2, 11363, 300), (2, 11363, 11363), (3, 5000, 1),
                                   (2, 22323, 3600), (1, 16384, 16384), (1, 40000, 20000), (1, 67200, 26880)])
def test_topk_exact(B, N, k):
    score = syn.det_randn(f"score{N}", (B, N))
    v, i = F.masked_topk_desc(score.to(DEV), k)
    rv, ri = R.topk_desc_stable(score, k)
    assert torch.equal(i.cpu(), ri)
    assert torch.equal(v.cpu(), rv)


@pytest.mark.parametrize("B,N,k", [(1, 45330, 300), (2, 36264, 300), (2, 27198, 300), (3, 18132, 300), (1, 17409, 1),
                                   (2, 65536, 2048), (1, 89250, 900), (1, 300000, 1500)])
@pytest.mark.parametrize("kind", ["random", "ties", "constant", "inf_tail"])
def test_long_rows_small_k_take_the_sliced_selection(B, N, k, kind):
    """Round 5: rows beyond the one-launch sort's 17 408 keys with k << N (the per-layer top-300 of the reference's 5scale
    pyramid, salience_transformer.py:366-367 on 45 330 rows): per-slice top-k + one sort of the survivors
    (filter_ops._sliced_topk) -- bit-exact against the stable descending sort, ties by position, with an index offset."""
    g = torch.Generator().manual_seed(N + k)
    score = torch.randn(B, N, generator=g)
    if kind == "ties":
        score = (score * 2).round() / 2                     # ~15 distinct values: every slice is full of ties
    elif kind == "constant":
        score = torch.full((B, N), 0.25)
    elif kind == "inf_tail":
        score[:, N - 5000:] = float("-inf")
        score[:, 7] = float("inf")
    called = []
    real = F._sliced_topk
    F._sliced_topk = lambda *a, **kw: (called.append(1), real(*a, **kw))[1]
    try:
        v, i = F.masked_topk_desc(score.to(DEV), k, index_offset=11)
        _, i2 = F.masked_topk_desc(score.to(DEV), k, want_scores=False)
    finally:
        F._sliced_topk = real
    assert len(called) == 2
    rv, ri = R.topk_desc_stable(score, k)
    # (300 000 keys: 37 slices x 1500 survivors are themselves longer than one workgroup's row and are sliced again)
    assert torch.equal(i.cpu(), ri + 11) and torch.equal(v.cpu(), rv) and torch.equal(i2.cpu(), ri)


@pytest.mark.parametrize("B,N,k", [(1, 67200, 16700), (2, 67200, 16800), (2, 40000, 20000), (1, 89250, 89250), (2, 24577, 6000),
                                   (2, 16800, 6680), (2, 16801, 6680), (1, 8192, 4000), (3, 8199, 8199), (2, 24576, 5000)])
def test_long_rows_large_k_take_sorted_slices_and_a_merge(B, N, k):
    """The finest level of a pyramid (top 6680 of 16 800 at 800 x 1333 -- since round 6 --, 16 700 of 67 200 on the
    reference's 5scale pyramid; mask + whole-array minimum as fill, an index offset, outputs into column slices of wider
    buffers): eight fully sorted slices + a stable merge, two chip-wide launches (filter_ops._sorted_slices_topk ->
    sdetr_masked_topk_sliced_f32) instead of one workgroup per row / the quadratic rank over the whole row -- bit-exact
    against the oracle, rows that do not divide into eight equal slices included."""
    score = syn.det_randn(f"ls{N}", (B, N))
    n7 = score[:, 3::7].shape[1]
    score[:, ::7][:, :n7] = score[:, 3::7]                                  # exact duplicates across slices
    mask = torch.zeros(B, N, dtype=torch.bool)
    mask[:, N - N // 9:] = True                                             # a padded tail: floods the bottom with the fill value
    mask[0, ::101] = True
    called = []
    real = F._sorted_slices_topk
    F._sorted_slices_topk = lambda *a, **kw: (called.append(1), real(*a, **kw))[1]
    try:
        fill = score.min().reshape(1).to(DEV)
        v, i = F.masked_topk_desc(score.to(DEV), k, mask=mask.to(DEV), fill_with_global_min=True, fill_value=fill, index_offset=500)
        wide_s = torch.zeros(B, k + 10, device=DEV)
        wide_i = torch.zeros(B, k + 10, dtype=torch.int64, device=DEV)
        F.masked_topk_desc(score.to(DEV), k, mask=mask.to(DEV), fill_with_global_min=True, fill_value=fill, index_offset=500,
                           out=(wide_s[:, 3:3 + k], wide_i[:, 3:3 + k]))
        v0, i0 = F.masked_topk_desc(score.to(DEV), k)                       # no mask
    finally:
        F._sorted_slices_topk = real
    assert len(called) == 3
    rv, ri = _oracle_topk(score, k, mask)
    assert torch.equal(i.cpu(), ri + 500) and torch.equal(v.cpu(), rv)
    assert torch.equal(wide_i[:, 3:3 + k].cpu(), ri + 500) and torch.equal(wide_s[:, 3:3 + k].cpu(), rv)
    rv0, ri0 = R.topk_desc_stable(score, k)
    assert torch.equal(i0.cpu(), ri0) and torch.equal(v0.cpu(), rv0)


@pytest.mark.parametrize("B,N,k", [(2, 273, 273), (2, 4200, 3360), (2, 16800, 6680)])
def test_masked_topk_with_ties(B, N, k):
    score = syn.det_randn("ms", (B, N))
    score[:, ::5] = score[:, 1::5][:, : score[:, ::5].shape[1]]  # plant exact duplicates
    mask = torch.zeros(B, N, dtype=torch.bool)
    mask[1, N // 2:] = True  # second image heavily padded: fill value ties everywhere
    mask[0, ::97] = True
    v, i = F.masked_topk_desc(score.to(DEV), k, mask=mask.to(DEV), fill_with_global_min=True, index_offset=1000)
    rv, ri = _oracle_topk(score, k, mask)
    assert torch.equal(i.cpu(), ri + 1000)
    assert torch.equal(v.cpu(), rv)


@pytest.mark.parametrize("B,N,k", [(2, 1024, 256), (2, 11363, 300), (3, 4096, 1024), (2, 9090, 300), (2, 2272, 300),
                                   (2, 6817, 300), (2, 4545, 300), (1, 24576, 1024), (2, 2560, 1024), (2, 5000, 1),
                                   (1, 24577, 300), (2, 1023, 300), (2, 16800, 6680), (2, 17408, 6000), (1, 17409, 6000)])
@pytest.mark.parametrize("kind", ["random", "ties", "constant", "masked", "sigmoid", "two_values", "padded_tail"])
def test_small_k_selection(B, N, k, kind):
    """k well below the row length (the encoder layers' top-300, the finest level's top-6680 of 16 800): the one-launch
    histogram sort (csrc/topk.hip, topk_hsort_kernel: rows of 1024 ... 17 408 keys) and, outside its shape, prefilter +
    rank.  Heavy ties, a constant row, +-inf / +-0, a mask whose fill value floods the
    row, scores squeezed into a narrow band (class score x foreground score), two key values only, a padded tail that
    ties at the minimum, a payload and an index offset -- bit-exact against the stable descending sort."""
    g = torch.Generator().manual_seed(N + k)
    score = torch.randn(B, N, generator=g)
    mask = None
    if kind == "ties":
        score = (score * 3).round() / 3
    elif kind == "constant":
        score[:] = 0.25
        score[0, 5] = -0.0
    elif kind == "masked":
        mask = torch.rand(B, N, generator=g) < 0.7
    elif kind == "sigmoid":
        score = torch.sigmoid(score - 4.0) * torch.sigmoid(torch.randn(B, N, generator=g))
    elif kind == "two_values":
        score = (score > 1.0).float() * 0.5 + 0.125
    elif kind == "padded_tail":
        score = torch.sigmoid(score)
        score[:, N // 3:] = float(score.min())
    if kind not in ("sigmoid", "two_values", "padded_tail"):
        score[0, :4] = torch.tensor([0.0, -0.0, float("inf"), -float("inf")])
    payload = torch.stack([torch.randperm(5 * N, generator=torch.Generator().manual_seed(b))[:N] for b in range(B)])
    kw = dict(mask=mask.to(DEV), fill_with_global_min=True) if mask is not None else {}
    v, i = F.masked_topk_desc(score.to(DEV), k, index_offset=7, **kw)
    rv, ri = _oracle_topk(score, k, mask)
    assert torch.equal(i.cpu(), ri + 7) and torch.equal(v.cpu(), rv)
    _, ip = F.masked_topk_desc(score.to(DEV), k, payload=payload.to(DEV), want_scores=False, **kw)
    assert torch.equal(ip.cpu(), payload.gather(1, ri))


def test_sort_with_payload_and_special_values():
    B, N = 2, 11363
    score = syn.det_randn("sp", (B, N))
    score[0, :4] = torch.tensor([0.0, -0.0, float("inf"), -float("inf")])
    payload = torch.stack([torch.randperm(50000, generator=torch.Generator().manual_seed(b))[:N] for b in range(B)])
    _, got = F.masked_topk_desc(score.to(DEV), N, payload=payload.to(DEV), want_scores=False)
    order = torch.sort(score, dim=1, descending=True, stable=True)[1]
    assert torch.equal(got.cpu(), payload.gather(1, order))


def test_topk_errors():
    s = torch.zeros(2, 10, device=DEV)
    with pytest.raises(RuntimeError):
        F.masked_topk_desc(s, 11)
    with pytest.raises(RuntimeError):
        F.masked_topk_desc(s.cpu(), 3)
    with pytest.raises(RuntimeError):
        F.masked_topk_desc(s, 3, mask=torch.zeros(2, 10, dtype=torch.bool, device=DEV))
    v, i = F.masked_topk_desc(s, 0)
    assert i.shape == (2, 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.int64])
@pytest.mark.parametrize("C", [256, 8, 6])
def test_gather_scatter_rows(dtype, C):
    if dtype == torch.bfloat16 and C % 2:
        pytest.skip("rows must be a multiple of 4 bytes")
    B, S, n = 2, 1000, 613
    src = (syn.det_randn("rows", (B, S, C)) * 100).to(dtype)
    idx = torch.stack([torch.randperm(S, generator=torch.Generator().manual_seed(b))[:n] for b in range(B)])
    got = F.gather_rows(src.to(DEV), idx.to(DEV))
    assert torch.equal(got.cpu(), torch.gather(src, 1, idx[..., None].expand(-1, -1, C)))
    dst = torch.zeros(B, S, C, dtype=dtype)
    new = (syn.det_randn("new", (B, n, C)) * 100).to(dtype)
    count = torch.tensor([n, 100])
    expect = dst.clone()
    for b in range(B):
        expect[b, idx[b, :count[b]]] = new[b, :count[b]]
    d = dst.to(DEV)
    F.scatter_rows_(d, idx.to(DEV), new.to(DEV), count.to(DEV))
    assert torch.equal(d.cpu(), expect)
    d2 = dst.to(DEV)
    F.scatter_rows_(d2, idx.to(DEV), new.to(DEV))
    expect2 = dst.clone().scatter(1, idx[..., None].expand(-1, -1, C), new)
    assert torch.equal(d2.cpu(), expect2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_class_max_times(dtype):
    B, Nq, C = 2, 1234, 91
    score = syn.det_randn("cmt", (B, Nq, C)).to(dtype)
    fg = syn.det_randn("cmt.fg", (B, Nq))
    got = F.class_max_times(score.to(DEV), fg.to(DEV))
    assert torch.equal(got.cpu(), score.max(-1)[0].float() * fg)


@pytest.mark.parametrize("sizes", [[(64, 96), (48, 80)], [(800, 1333)], [(300, 500), (333, 480), (100, 37)]])
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("one_launch", [True, False])
def test_pyramid_flatten_matches_reference_plumbing(sizes, bf16, one_launch):
    from salience_detr_amd import pyramid
    E = 64
    _, masks = syn.make_masks(sizes)
    shapes = pyramid.level_shapes_of(masks)
    feats = syn.make_feats(len(sizes), shapes, E, seed=2)
    pos = [syn.det_randn(f"pos{l}", tuple(f.shape)) for l, f in enumerate(feats)]
    le = syn.det_randn("le", (4, E))
    feat, posf, enc_in, mask, fb, pb, vr = F.pyramid_flatten([f.to(DEV) for f in feats], [p.to(DEV) for p in pos],
                                                             [m.to(DEV) for m in masks], le.to(DEV), want_bf16=bf16,
                                                             one_launch=one_launch)
    assert (vr.cpu() - R.level_misc(masks)[2]).abs().max() < 1e-7
    ref_feat = R.flatten_levels(feats)
    ref_mask = R.flatten_levels(masks)
    ref_pos = R.level_pos_embed({"level_embeds": le}, pos)
    assert torch.equal(feat.cpu(), ref_feat) and torch.equal(mask.cpu(), ref_mask)
    assert torch.equal(posf.cpu(), ref_pos)
    # keep-mask: same tokens survive as in the oracle's backbone_output_memory (identity enc_output / no LN)
    sd = {"enc_output.weight": torch.eye(E), "enc_output.bias": torch.zeros(E),
          "enc_output_norm.weight": torch.ones(E), "enc_output_norm.bias": torch.zeros(E)}
    shapes_t = torch.tensor(shapes)
    import torch.nn.functional as TF
    expect = TF.layer_norm(enc_in.cpu(), (E,))
    assert (R.backbone_output_memory(sd, ref_feat + ref_pos, ref_mask, shapes_t) - expect).abs().max() < 1e-5
    if bf16:
        assert torch.equal(fb.cpu(), ref_feat.to(torch.bfloat16)) and torch.equal(pb.cpu(), ref_pos.to(torch.bfloat16))
    else:
        assert fb is None and pb is None


@pytest.mark.parametrize("xdt,pdt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                     (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("C", [256, 32, 64])
def test_fused_layer_norm(xdt, pdt, C):
    B, N = 2, 777
    big = syn.det_randn("ln.x", (B, N + 50, C)).to(xdt)
    x = big[:, 20:20 + N]                        # batch-strided view, like one level of [B,S,C]
    res = syn.det_randn("ln.r", (B, N, C)).to(xdt)
    scale = syn.det_randn("ln.s", (B, N))
    alpha = torch.tensor([0.3])
    ln = torch.nn.LayerNorm(C)
    ln.weight.data = 1 + 0.1 * syn.det_randn("ln.w", (C,))
    ln.bias.data = 0.1 * syn.det_randn("ln.b", (C,))
    ln_d = torch.nn.LayerNorm(C).to(DEV).to(pdt)
    ln_d.load_state_dict({k: v.to(pdt) for k, v in ln.state_dict().items()})
    lw, lb = ln_d.weight.float().cpu(), ln_d.bias.float().cpu()
    tol = 1e-4 if xdt == torch.float32 else 2e-2  # fp32: rsqrt + summation order vs ATen, bar is 1e-3
    ref = lambda t: torch.nn.functional.layer_norm(t, (C,), lw, lb, ln.eps)
    got = F.fused_layer_norm(x.to(DEV), ln_d)
    assert (got.float().cpu() - ref(x.float())).abs().max() < tol
    got = F.fused_layer_norm(x.to(DEV), ln_d, residual=res.to(DEV))
    assert (got.float().cpu() - ref(x.float() + res.float())).abs().max() < tol
    got = F.fused_layer_norm(x.to(DEV), ln_d, row_scale=scale.to(DEV), alpha=alpha.to(DEV), out_dtype=torch.float32)
    xm = x.float() + x.float() * scale[..., None] * alpha
    assert (got.cpu() - ref(xm)).abs().max() < 2e-4


def test_column_mean_strided():
    z = syn.det_randn("cm", (2, 16800, 256)).to(DEV)
    got = F.column_mean(z[..., 128:])
    assert got.shape == (2, 1, 128)
    assert (got.cpu() - z[..., 128:].cpu().double().mean(1, keepdim=True).float()).abs().max() < 1e-6
    again = F.column_mean(z[..., 128:])
    assert torch.equal(got, again)  # deterministic


def _head_reference(pred, x, up=None, alpha=None, enc=None, enc_norm=None):
    """fp32 torch restatement of salience_transformer.py:36-47 (+ :143, + base_transformer.py:110-111), on CPU in
    float64-free plain fp32 so that it is the same arithmetic the oracle uses."""
    mem = enc_norm(enc(x)) if enc is not None else x
    y = mem if up is None else mem + mem * up.unsqueeze(-1) * alpha
    return pred(y).squeeze(-1), mem


@pytest.mark.parametrize("n,hw", [(1, (1, 1)), (63, (7, 9)), (64, (8, 8)), (77, (7, 11)), (1050, (25, 42)),
                                  (4200, (50, 84))])
@pytest.mark.parametrize("mode", ["plain", "row_scale", "coarse", "enc+coarse"])
@pytest.mark.parametrize("x3", [True, False], ids=["bf16x3", "f32mfma"])
def test_salience_head_matches_fp32_reference(n, hw, mode, x3, monkeypatch):
    """Both stage-1 kernels (fp32-input MFMA; bf16 MFMA on exact three-way splits) against the fp32 reference at the
    same bar: the split loses nothing the fp32 kernel keeps."""
    from salience_detr_amd.salience_filtering import MaskPredictor
    monkeypatch.setattr(F, "salience_head_bf16x3", x3)
    torch.manual_seed(n)
    B, C = 2, 256
    pred = MaskPredictor(C, C)
    for p in pred.parameters():          # non-trivial biases / norm parameters
        if p.dim() == 1:
            p.data = syn.det_randn(f"hp{p.numel()}", tuple(p.shape)) * 0.3 + (1.0 if p.numel() == C else 0.0)
    enc, enc_norm = torch.nn.Linear(C, C), torch.nn.LayerNorm(C)
    enc_norm.weight.data = 1 + 0.2 * syn.det_randn("eg", (C,))
    enc_norm.bias.data = 0.2 * syn.det_randn("eb", (C,))
    # the level sits inside a longer [B,S,C] buffer, as in the hot path
    full = syn.det_randn(f"hx{n}", (B, n + 13, C)) * 1.5
    x = full[:, 5:5 + n]
    alpha = torch.tensor([0.27])
    ch, cw = max(1, (hw[0] + 1) // 2), max(1, (hw[1] + 1) // 2)
    coarse = syn.det_randn(f"hc{n}", (B, 1, ch, cw))
    up = None
    if mode == "row_scale":
        up = syn.det_randn(f"hu{n}", (B, n))
    elif "coarse" in mode:
        up = torch.nn.functional.interpolate(coarse, size=hw, mode="bilinear", align_corners=True).reshape(B, n)
    with torch.no_grad():
        ref, ref_mem = _head_reference(pred, x, up, alpha if up is not None else None,
                                       enc if mode.startswith("enc") else None, enc_norm)
    pd, ed, nd = pred.to(DEV), enc.to(DEV), enc_norm.to(DEV)
    xd = full.to(DEV)[:, 5:5 + n]
    flat = torch.full((B, n + 3), -7.0, device=DEV)
    mem_out = torch.zeros(B, n + 2, C, device=DEV)
    kw = {}
    if mode == "row_scale":
        kw = dict(row_scale=up.to(DEV), alpha=alpha.to(DEV))
    elif "coarse" in mode:
        kw = dict(coarse_score=coarse.to(DEV), level_hw=hw, alpha=alpha.to(DEV))
    if mode.startswith("enc"):
        kw.update(enc_output=ed, enc_output_norm=nd, memory_out=mem_out[:, 1:1 + n])
    with torch.no_grad():
        got = F.salience_head(xd, pd, score_flat=flat[:, 2:2 + n], **kw)
        if mode == "row_scale":   # the module's own forward takes the same kernels
            via_module = pd(xd, row_scale=up.to(DEV), alpha=alpha.to(DEV)).squeeze(-1)
            assert torch.equal(via_module, got)
    scale = ref.abs().max().item() + 1.0
    assert (got.cpu() - ref).abs().max().item() <= 2e-5 * scale
    assert torch.equal(flat[:, 2:2 + n], got) and (flat[:, :2] == -7).all() and (flat[:, 2 + n:] == -7).all()
    if mode.startswith("enc"):
        assert (mem_out[:, 1:1 + n].cpu() - ref_mem).abs().max().item() <= 2e-5 * (ref_mem.abs().max().item() + 1)
        assert (mem_out[:, 0] == 0).all() and (mem_out[:, 1 + n:] == 0).all()


def test_salience_head_is_deterministic_and_rejects_bad_input():
    from salience_detr_amd.salience_filtering import MaskPredictor
    pred = MaskPredictor(256, 256).to(DEV)
    x = syn.det_randn("hdet", (2, 16700, 256)).to(DEV)
    with torch.no_grad():
        a, b = F.salience_head(x, pred), F.salience_head(x, pred)
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        F.salience_head(x.cpu(), pred)
    with pytest.raises(RuntimeError):
        F.salience_head(x[..., :128].contiguous(), pred)
    small = MaskPredictor(64, 64).to(DEV)          # other widths run the generic native path
    xs = syn.det_randn("hsm", (2, 50, 64)).to(DEV)
    with torch.no_grad():
        ref = small.cpu()(xs.cpu())
        got = small.to(DEV)(xs)
    assert (got.cpu() - ref).abs().max().item() <= 1e-4


def test_salience_head_is_bit_reproducible_with_enc_output_and_modulation():
    """Round 6: a schedule of stage 1 in which the last step of a LayerNorm's 16-lane sum was consumed behind a workgroup
    barrier gave ~20 wrong rows of 33 400 per launch, different ones every run (csrc/salience_head_core.h).  Twenty launches
    of the full form (enc_output + enc_output_norm inside, coarse-to-fine modulation) at the finest level's size must agree
    bit for bit -- scores and the enc_output memory."""
    from salience_detr_amd.salience_filtering import MaskPredictor
    torch.manual_seed(0)
    B, h, w = 2, 100, 167
    n = h * w
    pred = MaskPredictor(256, 256).to(DEV)
    enc, norm = torch.nn.Linear(256, 256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    alpha = torch.tensor([0.2], device=DEV)
    x = syn.det_randn("hrep", (B, n, 256)).to(DEV)
    coarse = syn.det_randn("hrep_c", (B, 1, (h + 1) // 2, (w + 1) // 2)).to(DEV)
    mem = torch.empty(B, n, 256, device=DEV)
    kw = dict(coarse_score=coarse, level_hw=(h, w), alpha=alpha, enc_output=enc, enc_output_norm=norm, memory_out=mem)
    with torch.no_grad():
        first = F.salience_head(x, pred, **kw).clone()
        first_mem = mem.clone()
        for _ in range(20):
            again = F.salience_head(x, pred, **kw)
            assert torch.equal(again, first)
            assert torch.equal(mem, first_mem)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prefix_row_movers_match_gather_scatter_semantics(dtype):
    """advance_rows / select_stack / encoder_finalize against the reference's gather / scatter formulation
    (salience_transformer.py:366-376, 454-495), incl. images whose focus count is below the layer's row count."""
    B, S, C, n0 = 3, 500, 256, 300
    counts = [300, 240, 120]
    torch.manual_seed(5)
    tokens = syn.det_randn("pt", (B, S, C)).to(dtype)
    perm = torch.stack([torch.randperm(S)[:n0] for _ in range(B)])
    focus = torch.tensor([300, 200, 90])
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[1, 400:] = True
    pad[2, 100:160] = True
    bg = syn.det_randn("pbg", (S, C)).to(dtype)
    outs = [syn.det_randn(f"py{k}", (B, c, C)).to(dtype) for k, c in enumerate(counts)]

    # reference formulation on the CPU: scatter every layer's live rows into token space, gather the next layer
    ref_out = tokens.clone()
    ref_next = []
    for k, c in enumerate(counts):
        for b in range(B):
            live = min(c, int(focus[b]))
            ref_out[b, perm[b, :live]] = outs[k][b, :live]
        if k + 1 < len(counts):
            ref_next.append(torch.stack([ref_out[b, perm[b, :counts[k + 1]]] for b in range(B)]))
    keep = torch.ones(B, S)
    keep.scatter_(1, perm[:, :counts[-1]], 0.0)
    keep = keep * (~pad)
    ref_final = (ref_out.float() + bg.float().unsqueeze(0) * keep.unsqueeze(-1)).to(dtype)

    tok_d, perm_d, focus_d = tokens.to(DEV), perm.to(DEV), focus.to(DEV)
    result = torch.zeros(B, n0, C, dtype=dtype, device=DEV)
    for k, c in enumerate(counts):
        nxt_rows = counts[k + 1] if k + 1 < len(counts) else 0
        nxt = F.advance_rows(outs[k].to(DEV), result, nxt_rows, tok_d, perm_d, focus_d)
        if nxt_rows:
            assert torch.equal(nxt.cpu(), ref_next[k])
        else:
            assert nxt is None
    final = F.encoder_finalize(tok_d, result, perm_d, focus_d, bg.to(DEV), pad.to(DEV), counts[-1])
    assert torch.equal(final.cpu(), ref_final)
    # without a count every row is live
    r2 = torch.zeros(B, n0, C, dtype=dtype, device=DEV)
    n2 = F.advance_rows(outs[1].to(DEV), r2, 100, tok_d, perm_d, None)
    assert torch.equal(n2.cpu(), outs[1][:, :100]) and torch.equal(r2[:, :240].cpu(), outs[1])

    # select_stack: query is a [B,c,C] buffer, pos a row prefix of a longer sorted buffer
    q = outs[1].to(DEV)
    pos_long = syn.det_randn("ppos", (B, n0, C)).to(dtype).to(DEV)
    sel = torch.stack([torch.randperm(240)[:37] for _ in range(B)]).to(DEV)
    st = F.select_stack(q, pos_long[:, :240], sel)
    qs = torch.stack([q[b, sel[b]] for b in range(B)])
    ps = torch.stack([pos_long[b, sel[b]] for b in range(B)])
    assert torch.equal(st[:, 37:], qs) and torch.equal(st[:, :37], qs + ps)
    # gather_rows from a row prefix; class_max_times with a strided scale
    g = F.gather_rows(pos_long[:, :240], sel)
    assert torch.equal(g, ps)
    score = syn.det_randn("pcs", (B, 240, 91)).to(dtype).to(DEV)
    fg_long = syn.det_randn("pfg", (B, n0)).to(DEV)
    got = F.class_max_times(score, fg_long[:, :240])
    assert torch.equal(got, score.float().max(-1)[0] * fg_long[:, :240])


def test_layer_norm_scatter_destination():
    B, n, m, C = 2, 37, 200, 256
    x = syn.det_randn("lsx", (B, 2 * n, C)).to(DEV)
    r = syn.det_randn("lsr", (B, n, C)).to(DEV)
    norm = torch.nn.LayerNorm(C).to(DEV)
    idx = torch.stack([torch.randperm(m)[:n] for _ in range(B)]).to(DEV)
    dst = syn.det_randn("lsd", (B, m, C)).to(DEV)
    want = dst.clone()
    plain = F.fused_layer_norm(x[:, n:], norm, residual=r)
    for b in range(B):
        want[b, idx[b]] = plain[b]
    out = F.fused_layer_norm(x[:, n:], norm, residual=r, scatter_index=idx, scatter_into=dst)
    assert out is dst and torch.equal(dst, want)


@pytest.mark.parametrize("T", [1, 31, 128, 129, 1000, 4545])
@pytest.mark.parametrize("hidden,splits", [(64, None), (64, 2), (2048, None), (2048, 1), (2048, 3), (2048, 7), (2048, 64)])
def test_fused_ffn_matches_fp32_reference(T, hidden, splits):
    """One-launch bf16 feed-forward vs the same block evaluated in fp32 on the bf16-rounded parameters; the
    framework's own bf16 path (two GEMMs + LayerNorm) sets the error scale."""
    torch.manual_seed(T + hidden)
    lin1, lin2, norm = torch.nn.Linear(256, hidden), torch.nn.Linear(hidden, 256), torch.nn.LayerNorm(256)
    lin1.bias.data.normal_(0, 0.5)
    lin2.bias.data.normal_(0, 0.5)
    norm.weight.data = 1 + 0.3 * syn.det_randn("fg", (256,))
    norm.bias.data = 0.3 * syn.det_randn("fb", (256,))
    x = (syn.det_randn(f"fx{T}", (T, 256)) * 1.5).to(torch.bfloat16)
    mods = [m.to(DEV).to(torch.bfloat16) for m in (lin1, lin2, norm)]
    xd = x.to(DEV)
    with torch.no_grad():
        f1, f2, fn = [m.float() for m in (torch.nn.Linear(256, hidden), torch.nn.Linear(hidden, 256), torch.nn.LayerNorm(256))]
        for dst, src in zip((f1, f2, fn), mods):
            dst.load_state_dict({k: v.float().cpu() for k, v in src.state_dict().items()})
        ref = fn(x.float() + f2(torch.relu(f1(x.float()))))
        assert F.fused_ffn_applies(torch.empty(8000, 256, dtype=torch.bfloat16, device=DEV), mods[0], mods[1], mods[2],
                                   torch.nn.ReLU())   # (small token counts are routed to the library GEMMs)
        got = F.fused_ffn(xd, *mods, hidden_splits=splits).float().cpu()
        base = mods[2](xd + mods[1](torch.relu(mods[0](xd)))).float().cpu()
    err, base_err = (got - ref).abs().max().item(), (base - ref).abs().max().item()
    assert err <= max(1.5 * base_err, 0.03), (err, base_err)
    assert (got - ref).abs().mean().item() <= 6e-3


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("B,n", [(1, 1), (2, 31), (2, 128), (3, 333), (2, 4545)])
def test_token_linear_kernels_match_framework_bf16_path(B, n):
    """Token-resident linear kernels vs the framework's own bf16 ops on the same rounded parameters; the fp32
    evaluation of the same block bounds both."""
    torch.manual_seed(B * 1000 + n)
    x = _bf(syn.det_randn(f"tlx{n}", (B, n, 256)) * 1.3).to(DEV)
    long_pos = _bf(syn.det_randn(f"tlp{n}", (B, n + 7, 256))).to(DEV)
    pos = long_pos[:, :n]
    lin = torch.nn.Linear(256, 384).to(DEV).to(torch.bfloat16)
    lin.bias.data = _bf(syn.det_randn("tlb", (384,))).to(DEV)
    with torch.no_grad():
        # plain and with the fused addend
        got = F.token_linear(x, lin.weight, lin.bias)
        ref32 = torch.nn.functional.linear(x.float(), lin.weight.float(), lin.bias.float())
        assert (got.float() - ref32).abs().max().item() <= 0.02 * (ref32.abs().max().item() + 1)
        got2 = F.token_linear(x, lin.weight, lin.bias, x_add=pos)
        ref2 = torch.nn.functional.linear((x + pos).float(), lin.weight.float(), lin.bias.float())
        assert (got2.float() - ref2).abs().max().item() <= 0.02 * (ref2.abs().max().item() + 1)
        # class head + max * scale
        head = torch.nn.Linear(256, 91).to(DEV).to(torch.bfloat16)
        head.bias.data = _bf(syn.det_randn("tlc", (91,)) - 2.0).to(DEV)
        fg_long = syn.det_randn(f"tlf{n}", (B, n + 5)).to(DEV)
        mc = F.class_head_max_times(x, head, fg_long[:, :n])
        logits32 = torch.nn.functional.linear(x.float(), head.weight.float(), head.bias.float())
        # the maximum comes from the fp32 accumulators (no bf16 rounding of the logits: the score picks the top-300)
        want = logits32.max(-1)[0] * fg_long[:, :n]
        assert (mc - want).abs().max().item() <= 2e-4 * (want.abs().max().item() + 1)
        frame = F.class_max_times(head(x), fg_long[:, :n])
        assert (mc - frame).abs().max().item() <= 0.02 * (want.abs().max().item() + 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_value_proj_head_major_matches_two_step_path(dtype):
    from salience_detr_amd.ms_deform_attn import value_to_head_major
    B, Nv, heads, groups = 2, 1234, 8, 3
    value = _bf(syn.det_randn("vpx", (B, Nv, 256))).to(DEV)
    w = _bf(syn.det_randn("vpw", (groups * 256, 256)) * 0.06).to(DEV)
    b = _bf(syn.det_randn("vpb", (groups * 256,))).to(DEV)
    pad = torch.zeros(B, Nv, dtype=torch.bool, device=DEV)
    pad[1, 1000:] = True
    with torch.no_grad():
        got = F.value_proj_head_major(value, w, b, pad, heads, groups, dtype)
        ref32 = torch.nn.functional.linear(value.float(), w.float(), b.float()).masked_fill(pad[..., None], 0.0)
        want = ref32.view(B, Nv, groups, heads, 32).permute(2, 0, 3, 1, 4)
        two_step = value_to_head_major(torch.nn.functional.linear(value, w, b), pad, heads, dtype, num_groups=groups)
    assert got.shape == (groups, B, heads, Nv, 32) and got.dtype == dtype
    tol = 0.02 * (want.abs().max().item() + 1)
    assert (got.float() - want).abs().max().item() <= tol
    assert (got.float() - two_step.float()).abs().max().item() <= tol
    assert (got[:, 1, :, 1000:] == 0).all()


def test_reference_points_and_fill_min_kernels():
    from salience_detr_amd.salience_encoder import SalienceTransformerEncoder
    shapes = [(25, 42), (13, 21), (7, 11), (4, 6)]
    S = sum(h * w for h, w in shapes)
    B = 3
    vr = (0.5 + 0.5 * torch.rand(B, 4, 2)).to(DEV)
    st, lsi = pyramid.shape_tensors(shapes, DEV)
    want = SalienceTransformerEncoder.get_reference_points(shapes, vr, device=DEV)       # [B,S,L,2]
    got = F.encoder_reference_points(vr, st, lsi, S)
    assert torch.equal(got, want)
    idx_long = torch.stack([torch.randperm(S)[:500] for _ in range(B)]).to(DEV)
    idx = idx_long[:, :333]                                                                # a row prefix view
    got_i = F.encoder_reference_points(vr, st, lsi, 0, index=idx)
    assert torch.equal(got_i, torch.stack([want[b, idx[b]] for b in range(B)]))
    score = syn.det_randn("fm", (B, S)).to(DEV)
    mask = torch.rand(B, S, device=DEV) < 0.3
    mins = torch.tensor([0.5, score.min().item(), 1.0, 2.0], device=DEV)
    assert torch.equal(F.masked_fill_min(score, mask, mins), torch.where(mask, score.min(), score))


def test_topk_into_column_slices_with_strided_mask_and_given_fill():
    B, S = 2, 900
    score_all = syn.det_randn("tkc", (B, S)).to(DEV)
    mask_all = torch.rand(B, S, device=DEV) < 0.2
    out_s = torch.full((B, 700), -9.0, device=DEV)
    out_i = torch.full((B, 700), -1, dtype=torch.int64, device=DEV)
    off = 0
    for start, n, k in ((0, 600, 400), (600, 300, 300)):
        sc = score_all[:, start:start + n].contiguous()
        fill = sc.min().reshape(1)
        v, i = F.masked_topk_desc(sc, k, mask=mask_all[:, start:start + n], fill_with_global_min=True, index_offset=start,
                                  fill_value=fill, out=(out_s[:, off:off + k], out_i[:, off:off + k]))
        rv, ri = R.topk_desc_stable(sc.cpu().masked_fill(mask_all[:, start:start + n].cpu(), sc.min().item()), k)
        assert torch.equal(out_i[:, off:off + k].cpu(), ri + start) and torch.equal(out_s[:, off:off + k].cpu(), rv)
        off += k


@pytest.mark.parametrize("B,n", [(1, 1), (2, 300), (2, 1000), (3, 129)])
def test_token_linear_ln_matches_reference_and_scatters(B, n):
    torch.manual_seed(n)
    lin = torch.nn.Linear(256, 256).to(DEV).to(torch.bfloat16)
    lin.bias.data = _bf(syn.det_randn("tnb", (256,))).to(DEV)
    norm = torch.nn.LayerNorm(256).to(DEV).to(torch.bfloat16)
    norm.weight.data = _bf(1 + 0.3 * syn.det_randn("tng", (256,))).to(DEV)
    norm.bias.data = _bf(0.3 * syn.det_randn("tnbb", (256,))).to(DEV)
    x = _bf(syn.det_randn(f"tnx{n}", (B, n, 256))).to(DEV)
    long_res = _bf(syn.det_randn(f"tnr{n}", (B, 2 * n + 3, 256))).to(DEV)
    res = long_res[:, n:2 * n]                                   # a row range of a longer buffer
    with torch.no_grad():
        ref = torch.nn.functional.layer_norm(res.float() + torch.nn.functional.linear(x.float(), lin.weight.float(), lin.bias.float()),
                                             (256,), norm.weight.float(), norm.bias.float(), norm.eps)
        base = norm(res + lin(x)).float()
        got = F.token_linear_ln(x, lin, norm, residual=res)
        err, base_err = (got.float() - ref).abs().max().item(), (base - ref).abs().max().item()
        assert err <= max(1.5 * base_err, 0.03), (err, base_err)
        m = n + 40
        idx = torch.stack([torch.randperm(m)[:n] for _ in range(B)]).to(DEV)
        dst = _bf(syn.det_randn(f"tnd{n}", (B, m, 256))).to(DEV)
        want = dst.clone()
        for b in range(B):
            want[b, idx[b]] = got[b]
        out = F.token_linear_ln(x, lin, norm, residual=res, scatter_index=idx, scatter_into=dst)
        assert out is dst and torch.equal(dst, want)


@pytest.mark.parametrize("sizes", [[5], [300, 200, 100, 7], [6680, 3360, 1050, 273], [1, 1, 1], [0, 4, 0, 9], [12000, 5000, 2000]])
def test_merge_of_sorted_segments_is_the_stable_sort(sizes):
    B = 2
    torch.manual_seed(sum(sizes))
    segs = []
    for n in sizes:
        v = torch.randn(B, n)
        v[:, : n // 3] = v[:, :1] if n else v[:, :0]            # ties inside a segment
        segs.append(torch.sort(v, dim=1, descending=True, stable=True)[0])
    score = torch.cat(segs, 1)
    if len(sizes) > 1 and sizes[0] and sizes[1]:
        score[:, sizes[0]:sizes[0] + max(1, sizes[1] // 4)] = score[:, :1]     # ties across segments
        score[:, sizes[0]:sizes[0] + sizes[1]] = torch.sort(score[:, sizes[0]:sizes[0] + sizes[1]], dim=1, descending=True, stable=True)[0]
    n = score.shape[1]
    payload = torch.stack([torch.randperm(n) for _ in range(B)]) + 1000
    starts = [sum(sizes[:i]) for i in range(len(sizes))]
    rv, ri = R.topk_desc_stable(score, n)
    gs, gi = F.merge_sorted_desc(score.to(DEV), payload.to(DEV), starts, want_scores=True)
    assert torch.equal(gs.cpu(), rv)
    assert torch.equal(gi.cpu(), torch.gather(payload, 1, ri))


def test_layer_norm_in_place_row_update():
    B, c, N, C = 2, 500, 37, 256
    buf = syn.det_randn("lgx", (B, c, C)).to(DEV)
    r = syn.det_randn("lgr", (B, N, C)).to(DEV)
    norm = torch.nn.LayerNorm(C).to(DEV)
    idx = torch.stack([torch.randperm(c)[:N] for _ in range(B)]).to(DEV)
    want = buf.clone()
    for b in range(B):
        want[b, idx[b]] = torch.nn.functional.layer_norm(buf[b, idx[b]] + r[b], (C,), norm.weight, norm.bias, norm.eps)
    out = F.fused_layer_norm(buf, norm, residual=r, scatter_index=idx, scatter_into=buf, gather_x=True)
    assert out is buf and (buf - want).abs().max().item() <= 1e-5


@pytest.mark.parametrize("B,N", [(1, 1), (2, 31), (2, 300), (2, 320), (2, 900), (1, 1100), (3, 1152)])
def test_attention_heads_matches_fp32_attention(B, N):
    """Dense self-attention kernel (32-channel heads, strided q / k / v slices of one projection output) against the same
    attention in fp32 on the bf16-rounded inputs; the framework's bf16 flash kernel sets the error scale."""
    torch.manual_seed(N)
    H = 8
    qkv = (torch.randn(B, N, 3 * 32 * H) * 1.5).to(torch.bfloat16).to(DEV)
    q, k, v = qkv[..., :256], qkv[..., 256:512], qkv[..., 512:]
    assert F.attention_heads_applies(q, k, v, H)
    got = F.attention_heads(q, k, v, H).float().cpu()
    split = lambda t: t.float().cpu().view(B, N, H, 32).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, N, 256)
    lib = torch.nn.functional.scaled_dot_product_attention(
        q.view(B, N, H, 32).transpose(1, 2), k.view(B, N, H, 32).transpose(1, 2), v.view(B, N, H, 32).transpose(1, 2)
    ).transpose(1, 2).reshape(B, N, 256).float().cpu()
    err, base = (got - ref).abs().max().item(), (lib - ref).abs().max().item()
    assert err <= max(2.0 * base, 0.03), (err, base)
    assert (got - ref).abs().mean().item() <= 4e-3
    with pytest.raises(RuntimeError):
        F.attention_heads(q.float(), k.float(), v.float(), H)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_encoder_prepare_sorted_equals_its_four_parts(dtype):
    """One launch = gather_rows(tokens), gather_rows(pos), score.gather, encoder_reference_points(index) -- bit for bit."""
    B, C = 2, 256
    shapes = [(13, 17), (7, 9), (4, 5), (2, 3)]
    t_shapes = torch.tensor(shapes, dtype=torch.int64)
    sizes = t_shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    S = int(sizes.sum())
    g = torch.Generator().manual_seed(5)
    tokens = torch.randn(B, S, C, generator=g).to(dtype).to(DEV)
    pos = torch.randn(B, S, C, generator=g).to(dtype).to(DEV)
    score = torch.randn(B, S, generator=g).to(DEV)
    n = 200
    wide = torch.stack([torch.randperm(S, generator=g)[:n + 11] for _ in range(B)]).to(DEV)
    index = wide[:, :n]                                    # a column prefix: batch stride > n
    vr = (torch.rand(B, 4, 2, generator=g) * 0.4 + 0.6).to(DEV)
    q, ps, fg, ref = F.encoder_prepare_sorted(tokens, pos, score, index, vr, t_shapes.to(DEV), lsi.to(DEV))
    assert torch.equal(q, F.gather_rows(tokens, index.contiguous()))
    assert torch.equal(ps, F.gather_rows(pos, index.contiguous()))
    assert torch.equal(fg, torch.gather(score, 1, index))
    assert torch.equal(ref, F.encoder_reference_points(vr, t_shapes.to(DEV), lsi.to(DEV), n, index=index))
    # round 6: the masked fill of foreground_score (salience_transformer.py:164-168) applied in the gather instead of as a
    # launch of its own over all tokens -- the same numbers on the gathered rows, and materialize() is the old tensor
    mask = (torch.rand(B, S, generator=g) < 0.3).to(DEV)
    mins = torch.tensor([0.5, -2.25, 1.0, -0.75], device=DEV)
    lazy = F.LazyForegroundScore(score, mask, mins)
    q2, ps2, fg2, ref2 = F.encoder_prepare_sorted(tokens, pos, lazy, index, vr, t_shapes.to(DEV), lsi.to(DEV))
    full = torch.where(mask, mins.min(), score)
    assert torch.equal(lazy.materialize(), full)
    assert torch.equal(fg2, torch.gather(full, 1, index)) and mask.gather(1, index).any()
    assert torch.equal(q2, q) and torch.equal(ps2, ps) and torch.equal(ref2, ref)


@pytest.mark.parametrize("dtype,n", [(torch.bfloat16, 200), (torch.bfloat16, 3001), (torch.float16, 777), (torch.bfloat16, 1),
                                     (torch.bfloat16, 33)])
def test_encoder_prepare_sorted_hands_on_the_first_layers_class_score(dtype, n):
    """The entry gather with a class head: the four results bit for bit, plus class_head(rows).max(-1) * foreground rows
    (salience_transformer.py:462, 366) -- against the fp32 statement and against the class head's own launch."""
    B, C = 2, 256
    shapes = [(100, 60), (50, 30), (25, 15), (13, 8)]
    t_shapes = torch.tensor(shapes, dtype=torch.int64)
    sizes = t_shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    S = int(sizes.sum())
    g = torch.Generator().manual_seed(n)
    tokens = torch.randn(B, S, C, generator=g).to(dtype).to(DEV)
    pos = torch.randn(B, S, C, generator=g).to(dtype).to(DEV)
    score = torch.randn(B, S, generator=g).to(DEV)
    index = torch.stack([torch.randperm(S, generator=g)[:n] for _ in range(B)]).to(DEV)
    vr = (torch.rand(B, 4, 2, generator=g) * 0.4 + 0.6).to(DEV)
    head = torch.nn.Linear(C, 91).to(DEV).to(dtype)
    with torch.no_grad():
        head.bias.copy_((torch.randn(91, generator=g) - 2.0).to(DEV))
    mask = (torch.rand(B, S, generator=g) < 0.3).to(DEV)
    mins = torch.tensor([0.5, -2.25, 1.0, -0.75], device=DEV)
    for sc in (score, F.LazyForegroundScore(score, mask, mins)):
        assert F.prepare_class_score_applies(tokens, sc, head)
        q, ps, fg, ref = F.encoder_prepare_sorted(tokens, pos, sc, index, vr, t_shapes.to(DEV), lsi.to(DEV))
        q2, ps2, fg2, ref2, cls = F.encoder_prepare_sorted(tokens, pos, sc, index, vr, t_shapes.to(DEV), lsi.to(DEV),
                                                           class_head=head)
        assert torch.equal(q, q2) and torch.equal(ps, ps2) and torch.equal(fg, fg2) and torch.equal(ref, ref2)
        with torch.no_grad():
            want = torch.nn.functional.linear(q.float(), head.weight.float(), head.bias.float()).max(-1)[0] * fg
            own = F.class_head_max_times(q, head, fg)
        bar = 2e-4 * (want.abs().max().item() + 1)
        assert cls.shape == (B, n) and (cls - want).abs().max().item() <= bar and (cls - own).abs().max().item() <= bar


@pytest.mark.parametrize("rows,next_rows,splits", [(300, 200, 4), (1200, 1200, 2), (1500, 0, 8), (700, 300, 1)])
def test_fused_ffn_advance_equals_ffn_then_advance_rows(rows, next_rows, splits):
    """The end-of-layer operator (FFN + row bookkeeping in the reduce pass) against the two operators it replaces:
    identical bits in sorted_result and the next layer's queries, rows past an image's focus count untouched / taken
    from the original tokens (salience_transformer.py:474-485)."""
    B, S, n0, C, Fh = 2, 2000, 1600, 256, 512
    torch.manual_seed(rows)
    l1, l2, norm = torch.nn.Linear(C, Fh), torch.nn.Linear(Fh, C), torch.nn.LayerNorm(C)
    l1, l2, norm = (m.to(DEV).to(torch.bfloat16) for m in (l1, l2, norm))
    x = (syn.det_randn(f"ffa{rows}", (B, rows, C)) * 0.7).to(DEV).to(torch.bfloat16)
    tokens = syn.det_randn("ffa_tok", (B, S, C)).to(DEV).to(torch.bfloat16)
    idx = torch.stack([torch.randperm(S)[:n0] for _ in range(B)]).to(DEV)
    count = torch.tensor([rows - 37, max(rows // 3, 1)], dtype=torch.int64, device=DEV)
    res_a = torch.full((B, n0, C), -3.0, dtype=torch.bfloat16, device=DEV)
    res_b = res_a.clone()
    with torch.no_grad():
        want_next = F.advance_rows(F.fused_ffn(x, l1, l2, norm, hidden_splits=splits), res_a, next_rows, tokens, idx, count)
        got_next = F.fused_ffn_advance(x, l1, l2, norm, res_b, next_rows, tokens, idx, count, hidden_splits=splits)
    assert torch.equal(res_a, res_b)
    assert (res_b[0, rows - 37:] == -3).all() and (res_b[1, max(rows // 3, 1):] == -3).all()
    if next_rows == 0:
        assert want_next is None and got_next is None
    else:
        assert torch.equal(want_next, got_next)


@pytest.mark.parametrize("rows,next_rows,splits,hidden", [(300, 200, 4, 512), (1200, 1200, 2, 512), (1500, 0, 8, 512),
                                                          (700, 300, 1, 512), (1111, 900, 1, 2048), (2272, 0, 5, 2048),
                                                          (640, 640, 16, 512)])
def test_attn_tail_ffn_advance_equals_the_two_launches(rows, next_rows, splits, hidden):
    """The layer-end operator (csrc/ffn.hip, TAIL form: output_proj + residual + norm1 in front of the feed-forward in
    ONE launch) against the launches it replaces -- token_linear_ln, then fused_ffn_advance -- and against an fp32
    evaluation of the same block (salience_transformer.py:390-394, 347-351).  Both bf16 paths round x = norm1(...) to bf16
    before the feed-forward; they differ in accumulation order only."""
    B, S, n0, C = 2, 3000, 2400, 256
    torch.manual_seed(rows + splits)
    mk = lambda m: m.to(DEV).to(torch.bfloat16)
    wo, n1 = mk(torch.nn.Linear(C, C)), mk(torch.nn.LayerNorm(C))
    l1, l2, n2 = mk(torch.nn.Linear(C, hidden)), mk(torch.nn.Linear(hidden, C)), mk(torch.nn.LayerNorm(C))
    with torch.no_grad():
        for m, tag in ((n1, "g1"), (n2, "g2")):
            m.weight.copy_((1.0 + 0.2 * syn.det_randn("atf." + tag, (C,))).to(DEV))
            m.bias.copy_((0.2 * syn.det_randn("atf.b" + tag, (C,))).to(DEV))
        wo.bias.copy_((0.3 * syn.det_randn("atf.bo", (C,))).to(DEV))
    sampled = (syn.det_randn(f"atf.s{rows}", (B, rows, C)) * 0.8).to(DEV).to(torch.bfloat16)
    query = (syn.det_randn(f"atf.q{rows}", (B, rows, C)) * 0.9).to(DEV).to(torch.bfloat16)
    tokens = syn.det_randn("atf.tok", (B, S, C)).to(DEV).to(torch.bfloat16)
    idx = torch.stack([torch.randperm(S)[:n0] for _ in range(B)]).to(DEV)
    count = torch.tensor([rows - 37, max(rows // 3, 1)], dtype=torch.int64, device=DEV)
    res_a = torch.full((B, n0, C), -3.0, dtype=torch.bfloat16, device=DEV)
    res_b = res_a.clone()
    act = torch.nn.ReLU()
    assert F.attn_tail_ffn_applies(sampled, query, wo, n1, l1, l2, n2, act) == (B * rows >= 1000)
    with torch.no_grad():
        x = F.token_linear_ln(sampled, wo, n1, residual=query)
        want_next = F.fused_ffn_advance(x, l1, l2, n2, res_a, next_rows, tokens, idx, count, hidden_splits=splits)
        got_next = F.attn_tail_ffn_advance(sampled, query, wo, n1, l1, l2, n2, res_b, next_rows, tokens, idx, count,
                                           hidden_splits=splits)
        # fp32 statement with x rounded to bf16 where both kernels round it
        x32 = torch.nn.functional.layer_norm(query.float() + torch.nn.functional.linear(sampled.float(), wo.weight.float(), wo.bias.float()),
                                             (C,), n1.weight.float(), n1.bias.float(), n1.eps)
        xb = x32.to(torch.bfloat16).float()
        y32 = torch.nn.functional.layer_norm(xb + torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(
            xb, l1.weight.float(), l1.bias.float())), l2.weight.float(), l2.bias.float()), (C,), n2.weight.float(), n2.bias.float(), n2.eps)
    live0, live1 = rows - 37, max(rows // 3, 1)
    # untouched rows stay untouched, the rest agree with the two-launch path to bf16 round-off of the outputs
    assert (res_b[0, live0:] == -3).all() and (res_b[1, live1:] == -3).all()
    for b, live in ((0, live0), (1, live1)):
        got, ref2, ref32 = res_b[b, :live].float(), res_a[b, :live].float(), y32[b, :live]
        assert (got - ref32).abs().max().item() <= 0.06, (got - ref32).abs().max().item()
        assert (got - ref32).abs().mean().item() <= 4e-3
        # as close to the fp32 statement as the two-launch path is, and close to that path itself
        assert (got - ref32).abs().mean().item() <= 1.5 * (ref2 - ref32).abs().mean().item() + 1e-4
        assert (got - ref2).abs().max().item() <= 0.06
    if next_rows == 0:
        assert want_next is None and got_next is None
    else:
        for b, live in ((0, live0), (1, live1)):
            n_live = min(live, next_rows)
            assert (got_next[b, :n_live].float() - want_next[b, :n_live].float()).abs().max().item() <= 0.06
            assert torch.equal(got_next[b, n_live:], want_next[b, n_live:])     # original tokens, copied


@pytest.mark.parametrize("rows,next_rows,hidden", [(700, 300, 512), (1111, 900, 2048), (640, 640, 512), (1300, 1, 512),
                                                   (1025, 33, 512)])
def test_layer_end_hands_on_the_next_layers_class_score(rows, next_rows, hidden):
    """The layer-end launch (one hidden piece: its epilogue; more: the second pass) also does the row bookkeeping and returns
    the NEXT layer's selection score of the rows it hands on (salience_transformer.py:462, 366) -- against the separate launches:
    identical rows, score = class_head_max_times(next rows) up to fp32 accumulation order."""
    B, S, n0, C = 2, 3000, 2400, 256
    torch.manual_seed(rows)
    mk = lambda m: m.to(DEV).to(torch.bfloat16)
    wo, n1 = mk(torch.nn.Linear(C, C)), mk(torch.nn.LayerNorm(C))
    l1, l2, n2 = mk(torch.nn.Linear(C, hidden)), mk(torch.nn.Linear(hidden, C)), mk(torch.nn.LayerNorm(C))
    head = mk(torch.nn.Linear(C, 91))
    with torch.no_grad():
        head.bias.copy_((syn.det_randn("nx.hb", (91,)) - 2.0).to(DEV))
    sampled = (syn.det_randn(f"nx.s{rows}", (B, rows, C)) * 0.8).to(DEV).to(torch.bfloat16)
    query = (syn.det_randn(f"nx.q{rows}", (B, rows, C)) * 0.9).to(DEV).to(torch.bfloat16)
    tokens = syn.det_randn("nx.tok", (B, S, C)).to(DEV).to(torch.bfloat16)
    idx = torch.stack([torch.randperm(S)[:n0] for _ in range(B)]).to(DEV)
    count = torch.tensor([rows - 37, max(rows // 3, 1)], dtype=torch.int64, device=DEV)
    fg_long = syn.det_randn(f"nx.fg{rows}", (B, n0)).to(DEV)
    res_a = torch.full((B, n0, C), -3.0, dtype=torch.bfloat16, device=DEV)
    res_b = res_a.clone()
    with torch.no_grad():
        want_next = F.attn_tail_ffn_advance(sampled, query, wo, n1, l1, l2, n2, res_a, next_rows, tokens, idx, count,
                                            hidden_splits=1)
        got_next, score = F.attn_tail_ffn_advance(sampled, query, wo, n1, l1, l2, n2, res_b, next_rows, tokens, idx, count,
                                                  hidden_splits=1, next_class_head=head, foreground=fg_long)
        assert score is not None and score.shape == (B, next_rows)
        assert torch.equal(res_a, res_b) and torch.equal(want_next, got_next)
        want_score = F.class_head_max_times(got_next, head, fg_long[:, :next_rows]) if B * next_rows >= 32 else None
        logits = torch.nn.functional.linear(got_next.float(), head.weight.float(), head.bias.float())
        ref = logits.max(-1)[0] * fg_long[:, :next_rows]
        assert (score - ref).abs().max().item() <= 2e-4 * (ref.abs().max().item() + 1)
        if want_score is not None:
            assert (score - want_score).abs().max().item() <= 2e-4 * (ref.abs().max().item() + 1)
        # more hidden pieces: the second pass (partial sums + LayerNorm + bookkeeping) hands the score on -- the rows are
        # those of the same split without the class head, bit for bit
        for splits in (2, 3, 5):
            res_c = torch.full((B, n0, C), -3.0, dtype=torch.bfloat16, device=DEV)
            res_d = res_c.clone()
            want3 = F.attn_tail_ffn_advance(sampled, query, wo, n1, l1, l2, n2, res_c, next_rows, tokens, idx, count,
                                            hidden_splits=splits)
            nxt3, score3 = F.attn_tail_ffn_advance(sampled, query, wo, n1, l1, l2, n2, res_d, next_rows, tokens, idx, count,
                                                   hidden_splits=splits, next_class_head=head, foreground=fg_long)
            assert score3 is not None and score3.shape == (B, next_rows)
            assert torch.equal(res_c, res_d) and torch.equal(want3, nxt3)
            logits3 = torch.nn.functional.linear(nxt3.float(), head.weight.float(), head.bias.float())
            ref3 = logits3.max(-1)[0] * fg_long[:, :next_rows]
            assert (score3 - ref3).abs().max().item() <= 2e-4 * (ref3.abs().max().item() + 1)
            if B * next_rows >= 32:
                assert torch.equal(score3, F.class_head_max_times(nxt3, head, fg_long[:, :next_rows])) or \
                    (score3 - F.class_head_max_times(nxt3, head, fg_long[:, :next_rows])).abs().max().item() <= \
                    2e-4 * (ref3.abs().max().item() + 1)


@pytest.mark.parametrize("n,hw,mode", [(273, (13, 21), "enc"), (1050, (25, 42), "enc+coarse")])
def test_salience_head_carrying_a_value_projection_job(n, hw, mode):
    """fused_head_value.hip: stage 1 of a coarse level and a slice of the encoder's value projection in ONE launch
    give the same bits as the two launches on their own (scores, enc_output memory, head-major value maps)."""
    from salience_detr_amd.salience_filtering import MaskPredictor
    torch.manual_seed(7)
    B, C, Nv, heads, groups = 2, 256, 1500, 8, 6
    pred, enc, enc_norm = MaskPredictor(C, C).to(DEV), torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)
    x = (syn.det_randn(f"vx{n}", (B, n, C)) * 1.2).to(DEV)
    coarse = syn.det_randn(f"vc{n}", (B, 1, (hw[0] + 1) // 2, (hw[1] + 1) // 2)).to(DEV)
    alpha = torch.tensor([0.2], device=DEV)
    tokens = syn.det_randn("vtok", (B, Nv, C)).to(DEV).to(torch.bfloat16)
    w = (syn.det_randn("vw", (groups * heads * 32, C)) * 0.05).to(DEV).to(torch.bfloat16)
    bias = (syn.det_randn("vb", (groups * heads * 32,)) * 0.1).to(DEV).to(torch.bfloat16)
    pad = (syn.det_rand("vpad", (B, Nv)) > 0.9).to(DEV)
    kw = dict(enc_output=enc, enc_output_norm=enc_norm)
    if "coarse" in mode:
        kw.update(coarse_score=coarse, level_hw=hw, alpha=alpha)
    with torch.no_grad():
        mem_a, mem_b = torch.zeros(B, n, C, device=DEV), torch.zeros(B, n, C, device=DEV)
        want_score = F.salience_head(x, pred, memory_out=mem_a, **kw)
        want_maps = F.value_proj_head_major(tokens, w, bias, pad, heads, groups, torch.float16)
        maps, jobs = F.plan_value_projection(tokens, w, bias, pad, heads, groups, torch.float16, parts=(2, 1, 3))
        maps.fill_(-5.0)
        got_score = F.salience_head(x, pred, memory_out=mem_b, value_job=jobs[0], value_job2=jobs[1], **kw)
        assert jobs[0].done and jobs[1].done and not jobs[2].done      # stage 1 carried two layers, stage 2 one
        assert torch.equal(maps[:3], want_maps[:3]) and (maps[3:] == -5).all()
        jobs[2].run()
        jobs[2].run()   # idempotent
    assert torch.equal(got_score, want_score) and torch.equal(mem_a, mem_b)
    assert torch.equal(maps, want_maps)


def test_salience_head_carrying_a_rank_job():
    """fused_head_value.hip: stage 1 of a level also carries the (deferred) top-k of the next coarser level -- with and
    without a value-projection job in the same launch -- and gives the bits of the separate launches."""
    from salience_detr_amd.salience_filtering import MaskPredictor
    torch.manual_seed(11)
    B, C, n = 2, 256, 1050
    pred = MaskPredictor(C, C).to(DEV)
    x = (syn.det_randn("rjx", (B, n, C)) * 1.1).to(DEV)
    score = syn.det_randn("rjs", (B, 4200)).to(DEV)
    score[0, 100:140] = score[0, 7]                      # ties: resolved by position
    mask = torch.zeros(B, 6000, dtype=torch.bool, device=DEV)[:, 900:5100]
    mask[:, 4000:] = True
    fill = score.min().reshape(1)
    k = 3360
    want_s, want_i = F.masked_topk_desc(score, k, mask=mask, fill_with_global_min=True, index_offset=21, fill_value=fill)
    tokens = syn.det_randn("rjt", (B, 900, C)).to(DEV).to(torch.bfloat16)
    w = (syn.det_randn("rjw", (2 * 8 * 32, C)) * 0.05).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        want_head = F.salience_head(x, pred)
        want_maps = F.value_proj_head_major(tokens, w, None, None, 8, 2, torch.float16)
        for with_value in (False, True):
            out = (torch.full((B, k + 5), -1.0, device=DEV), torch.full((B, k + 5), -1, dtype=torch.int64, device=DEV))
            job = F.plan_masked_topk(score, k, mask, fill, 21, (out[0][:, 2:2 + k], out[1][:, 2:2 + k]))
            assert job is not None and not job.done
            vjob = None
            if with_value:
                maps, (vjob,) = F.plan_value_projection(tokens, w, None, None, 8, 2, torch.float16, parts=1)
            got_head = F.salience_head(x, pred, rank_job=job, value_job=vjob)
            assert job.done and torch.equal(got_head, want_head)
            assert torch.equal(out[0][:, 2:2 + k], want_s) and torch.equal(out[1][:, 2:2 + k], want_i)
            assert (out[1][:, :2] == -1).all() and (out[1][:, 2 + k:] == -1).all()
            if with_value:
                assert vjob.done and torch.equal(maps, want_maps)
    # a shape that needs the prefilter is run at once instead of being planned
    big = syn.det_randn("rjb", (B, 16800)).to(DEV)
    o = (torch.empty(B, 300, device=DEV), torch.empty(B, 300, dtype=torch.int64, device=DEV))
    assert F.plan_masked_topk(big, 300, None, fill, 0, o) is None
    assert torch.equal(o[1], F.masked_topk_desc(big, 300)[1])


def test_salience_head_carrying_the_finalize_pass():
    """The token-space pass of the encoder output carried by a stage-1 launch + the sorted-rows pass alone give the
    bits of ``encoder_finalize``'s two launches."""
    from salience_detr_amd.salience_filtering import MaskPredictor
    torch.manual_seed(13)
    B, S, C, n0 = 2, 3000, 256, 1700
    pred = MaskPredictor(C, C).to(DEV)
    x = syn.det_randn("fjx", (B, 4200, C)).to(DEV)
    tokens = syn.det_randn("fjt", (B, S, C)).to(DEV).to(torch.bfloat16)
    bg = syn.det_randn("fjb", (S, C)).to(DEV).to(torch.bfloat16)
    pad = (syn.det_rand("fjp", (B, S)) > 0.8).to(DEV)
    result = syn.det_randn("fjr", (B, n0, C)).to(DEV).to(torch.bfloat16)
    idx = torch.stack([torch.randperm(S)[:n0] for _ in range(B)]).to(DEV)
    count = torch.tensor([n0 - 100, 900], dtype=torch.int64, device=DEV)
    with torch.no_grad():
        want = F.encoder_finalize(tokens, result, idx, count, bg, pad, 400)
        want_head = F.salience_head(x, pred)
        job = F.FinalizeJob(tokens, bg, pad)
        got_head = F.salience_head(x, pred, finalize_job=job)
        assert job.done and torch.equal(got_head, want_head)
        got = F.encoder_finalize(tokens, result, idx, count, bg, pad, 400, finalize_job=job)
    assert torch.equal(got, want)


def _head_fp64(pred, x, enc, norm, coarse, hw, alpha, row_scale=None):
    """The head of one level in fp64 (salience_transformer.py:16-47, 139-143; base_transformer.py:104-111)."""
    B, n, C = x.shape
    xd = x.double()
    if enc is not None:
        xd = torch.nn.functional.layer_norm(torch.nn.functional.linear(xd, enc.weight.double(), enc.bias.double()), (C,),
                                            norm.weight.double(), norm.bias.double(), norm.eps)
    if coarse is not None:
        up = torch.nn.functional.interpolate(coarse.double(), size=hw, mode="bilinear", align_corners=True).view(B, n, 1)
        xd = xd + xd * up * alpha.double()
    if row_scale is not None:
        xd = xd + xd * row_scale.double().unsqueeze(-1) * alpha.double()
    l1n, l1 = pred.layer1[0], pred.layer1[1]
    z = torch.nn.functional.gelu(torch.nn.functional.linear(torch.nn.functional.layer_norm(
        xd, (C,), l1n.weight.double(), l1n.bias.double(), l1n.eps), l1.weight.double(), l1.bias.double()))
    z = torch.cat([z[..., :128], z[..., 128:].mean(1, keepdim=True).expand(-1, n, -1)], -1)
    for i, m in enumerate(pred.layer2):
        z = torch.nn.functional.linear(z, m.weight.double(), m.bias.double()) if i % 2 == 0 else torch.nn.functional.gelu(z)
    return z.squeeze(-1)


def _perturbed_predictor(seed):
    from salience_detr_amd.salience_filtering import MaskPredictor
    torch.manual_seed(seed)
    pred = MaskPredictor(256, 256).to(DEV)
    with torch.no_grad():   # (the default init has unit LayerNorm weights and zero biases: c0 would be zero)
        pred.layer1[0].weight.add_((0.2 * syn.det_randn("hh.g", (256,))).to(DEV))
        pred.layer1[0].bias.add_((0.3 * syn.det_randn("hh.b", (256,))).to(DEV))
        pred.layer1[1].bias.add_((0.2 * syn.det_randn("hh.lb", (256,))).to(DEV))
    return pred


@pytest.mark.parametrize("hw", [(13, 21), (50, 84), (100, 167)])
@pytest.mark.parametrize("with_enc", [True, False])
@pytest.mark.parametrize("mod", ["coarse", "row_scale", "none"])
def test_hoisted_salience_head_against_the_per_level_form_and_fp64(hw, with_enc, mod):
    """include/salience_hip.h, sdetr_salience_head_hoist_x3 + sdetr_salience_head_modulate: layer1.Linear(LN(s x)) =
    k G + c0 with both 256 x 256 products taken before s exists.  The scores are as close to an fp64 evaluation of the
    reference's expressions as the per-level kernels' are (both ~1e-6 at scores of order 1), the two forms agree to 3e-6,
    enc_output_norm's output is the same bits, and the hoisted form is bit-reproducible."""
    h, w = hw
    n, B, C = h * w, 2, 256
    pred = _perturbed_predictor(3)
    enc, norm = (torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)) if with_enc else (None, None)
    x = (syn.det_randn(f"hh.x{n}", (B, n, C)) * 1.3).to(DEV)
    coarse = syn.det_randn(f"hh.c{n}", (B, 1, (h + 1) // 2, (w + 1) // 2)).to(DEV) if mod == "coarse" else None
    rs = syn.det_randn(f"hh.r{n}", (B, n)).to(DEV) if mod == "row_scale" else None
    alpha = torch.tensor([-0.7 if mod == "row_scale" else 0.3], device=DEV)   # (row_scale * alpha reaches s < 0)
    kw = dict(enc_output=enc, enc_output_norm=norm) if with_enc else {}
    km = dict(coarse_score=coarse, level_hw=hw, alpha=alpha) if coarse is not None else (
        dict(row_scale=rs, alpha=alpha) if rs is not None else {})
    with torch.no_grad():
        mo_a = torch.zeros_like(x) if with_enc else None
        mo_b = torch.zeros_like(x) if with_enc else None
        direct = F.salience_head(x, pred, memory_out=mo_a, **kw, **km)
        hh = F.salience_head_hoist(x, pred, memory_out=mo_b, **kw)
        hoisted = F.salience_head(x, pred, hoisted=hh.level(0, n), **km)
        ref = _head_fp64(pred, x, enc, norm, coarse, hw, alpha, rs)
        for _ in range(5):
            hh2 = F.salience_head_hoist(x, pred, **kw)
            assert torch.equal(hh2.g, hh.g) and torch.equal(hh2.sigma, hh.sigma)
            assert torch.equal(F.salience_head(x, pred, hoisted=hh2.level(0, n), **km), hoisted)
    scale = ref.abs().max().item() + 1.0
    e_direct, e_hoisted = (direct.double() - ref).abs().max().item(), (hoisted.double() - ref).abs().max().item()
    assert e_hoisted <= 2e-6 * scale, (e_hoisted, scale)
    assert e_hoisted <= 2.0 * e_direct + 2e-7 * scale, (e_hoisted, e_direct)
    assert (direct - hoisted).abs().max().item() <= 3e-6 * scale
    if with_enc:
        assert torch.equal(mo_a, mo_b)


def test_hoisted_salience_head_on_constant_rows_and_zero_scale():
    """Rows without variance (sigma = 0: G = 0, z = GELU(c0)) and a modulation factor s = 0 (k = 0) are the limits of the
    factorisation; the per-level form normalises such a row to beta, i.e. to the same c0."""
    B, n, C = 2, 96, 256
    pred = _perturbed_predictor(5)
    x = (syn.det_randn("hz.x", (B, n, C)) * 0.8).to(DEV)
    x[0, 5] = 0.37
    x[1, 64:70] = -2.0
    rs = syn.det_randn("hz.r", (B, n)).to(DEV)
    rs[0, 9] = -2.0                                  # s = 1 + (-2) * 0.5 = 0
    alpha = torch.tensor([0.5], device=DEV)
    with torch.no_grad():
        hh = F.salience_head_hoist(x, pred)
        assert hh.sigma[0, 5].item() == 0.0 and (hh.sigma[1, 64:70] == 0).all() and (hh.g[0, 5] == 0).all()
        assert torch.isfinite(hh.g).all() and torch.isfinite(hh.sigma).all()
        direct = F.salience_head(x, pred, row_scale=rs, alpha=alpha)
        hoisted = F.salience_head(x, pred, row_scale=rs, alpha=alpha, hoisted=hh)
    assert torch.isfinite(hoisted).all()
    assert (direct - hoisted).abs().max().item() <= 3e-6 * (direct.abs().max().item() + 1.0)


@pytest.mark.parametrize("fin_level", [None, 1, "merge"])
def test_level_filtering_with_the_hoisted_head_carries_the_same_jobs(fin_level):
    """salience_filtering.level_filtering, HOIST_HEAD: scores within 3e-6 of the per-level form on a four-level pyramid,
    the same top-k sets except where the per-level scores themselves tie within 1e-5 at a level's cut, and the jobs its
    launches carry (value projection on the stage-2 launches, the finalize pass on the hoisted or a modulation launch, the
    deferred ranks) give the bits of their stand-alone launches."""
    from salience_detr_amd import salience_filtering as SF
    B, C, heads, groups = 2, 256, 8, 6
    shapes = [(40, 56), (20, 28), (10, 14), (5, 7)]
    starts = [0]
    for hh_, ww_ in shapes[:-1]:
        starts.append(starts[-1] + hh_ * ww_)
    S = sum(a * b for a, b in shapes)
    pred = _perturbed_predictor(9)
    enc, norm = torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)
    alpha = torch.tensor([0.3, 0.2, -0.4, 0.1], device=DEV)
    x = (syn.det_randn("lf.x", (B, S, C)) * 1.1).to(DEV)
    mask = (syn.det_rand("lf.m", (B, S)) > 0.93).to(DEV)
    tokens = syn.det_randn("lf.t", (B, S, C)).to(DEV).to(torch.bfloat16)
    wv = (syn.det_randn("lf.w", (groups * heads * 32, C)) * 0.05).to(DEV).to(torch.bfloat16)
    background = syn.det_randn("lf.bg", (S, C)).to(DEV).to(torch.bfloat16)
    ks = [int(0.4 * shapes[0][0] * shapes[0][1]), int(0.8 * shapes[1][0] * shapes[1][1]), shapes[2][0] * shapes[2][1],
          shapes[3][0] * shapes[3][1]]

    def run(hoist):
        maps, jobs = F.plan_value_projection(tokens, wv, None, mask, heads, groups, torch.float16,
                                             parts=(3, 3) if hoist else (2, 1, 2, 1))
        fin = F.FinalizeJob(tokens, background, mask)
        mem = torch.zeros_like(x)
        flat = torch.zeros(B, S, device=DEV)
        extras = {}
        old = SF.HOIST_HEAD, SF.FINALIZE_LEVEL
        SF.HOIST_HEAD, SF.FINALIZE_LEVEL = hoist, fin_level
        try:
            with torch.no_grad():
                sc, inds, lsc = SF.level_filtering(x, mask, shapes, starts, ks, pred, alpha, enc_output=enc, enc_output_norm=norm,
                                                   memory_out=mem, score_flat=flat, extras=extras, value_jobs=jobs,
                                                   finalize_job=fin)
                for j in jobs:
                    j.run()
                fin.run()
        finally:
            SF.HOIST_HEAD, SF.FINALIZE_LEVEL = old
        return flat, inds, mem, maps, fin.out, extras

    flat0, inds0, mem0, maps0, fin0, ex0 = run(False)
    flat1, inds1, mem1, maps1, fin1, ex1 = run(True)
    assert (flat0 - flat1).abs().max().item() <= 3e-6 * (flat0.abs().max().item() + 1.0)
    assert torch.equal(mem0, mem1) and torch.equal(maps0, maps1) and torch.equal(fin0, fin1)
    with torch.no_grad():
        want_maps = F.value_proj_head_major(tokens, wv, None, mask, heads, groups, torch.float16)
    assert torch.equal(maps1, want_maps)
    assert (ex0["level_min"] - ex1["level_min"]).abs().max().item() <= 3e-6
    for l in range(4):
        for b in range(B):
            a, c = set(inds0[l][b].tolist()), set(inds1[l][b].tolist())
            for t in a ^ c:   # only scores that tie at the level's cut may change sides
                lvl_scores = torch.where(mask[b, starts[l]:starts[l] + shapes[l][0] * shapes[l][1]], ex0["level_min"][l],
                                         flat0[b, starts[l]:starts[l] + shapes[l][0] * shapes[l][1]])
                cut = torch.sort(lvl_scores, descending=True)[0][ks[l] - 1].item()
                s_t = ex0["level_min"][l].item() if mask[b, t] else flat0[b, t].item()
                assert abs(s_t - cut) <= 1e-5, (l, b, t)


def test_sliced_topk_merge_carries_a_rank_job_and_the_finalize_pass():
    """csrc/topk.hip, merge_sorted_lds_rank_kernel: the merge launch of the finest level's sliced top-k takes a pending
    rank job (the deferred top-k of the level before) and the token-space pass of the encoder's output along -- the bits
    of the three launches on their own."""
    torch.manual_seed(21)
    B, N, k, S, C = 2, 16700, 6680, 3000, 256
    score = syn.det_randn("mj.s", (B, N)).to(DEV)
    score[1, 500:560] = score[1, 3]
    mask = torch.zeros(B, N, dtype=torch.bool, device=DEV)
    mask[:, 16000:] = True
    fill = score.min().reshape(1)
    s1 = syn.det_randn("mj.s1", (B, 4200)).to(DEV)
    m1 = torch.zeros(B, 4200, dtype=torch.bool, device=DEV)
    m1[0, 4100:] = True
    f1 = s1.min().reshape(1)
    tokens = syn.det_randn("mj.t", (B, S, C)).to(DEV).to(torch.bfloat16)
    background = syn.det_randn("mj.bg", (S, C)).to(DEV).to(torch.bfloat16)
    pad = (syn.det_rand("mj.p", (B, S)) > 0.8).to(DEV)
    want = F.masked_topk_desc(score, k, mask=mask, fill_with_global_min=True, index_offset=7, fill_value=fill)
    want1 = F.masked_topk_desc(s1, 3360, mask=m1, fill_with_global_min=True, index_offset=16800, fill_value=f1)
    want_fin = tokens + torch.where(pad[..., None], torch.zeros_like(background[None]), background[None])
    out1 = (torch.full((B, 3400), -1.0, device=DEV), torch.full((B, 3400), -1, dtype=torch.int64, device=DEV))
    job = F.plan_masked_topk(s1, 3360, m1, f1, 16800, (out1[0][:, 5:3365], out1[1][:, 5:3365]))
    fin = F.FinalizeJob(tokens, background, pad)
    assert job is not None
    got = F.masked_topk_desc(score, k, mask=mask, fill_with_global_min=True, index_offset=7, fill_value=fill,
                             carry_rank=job, carry_finalize=fin)
    assert job.done and fin.done
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert torch.equal(out1[0][:, 5:3365], want1[0]) and torch.equal(out1[1][:, 5:3365], want1[1])
    assert (out1[1][:, :5] == -1).all() and (out1[1][:, 3365:] == -1).all()
    assert torch.equal(fin.out, want_fin)


@pytest.mark.parametrize("n,hw", [(273, (13, 21)), (1050, (25, 42)), (4200, (50, 84))])
def test_stage2_with_the_constant_in_its_blocks(n, hw):
    """csrc/salience_head_core.h, stage2_const_in_block: stage 2 of a hoisted level sums the partial sums and takes the
    128 x 128 product in every block instead of reading the const launch's result -- the same scores to fp32 round-off
    (another, equally fixed, order of additions), the same minimum bookkeeping, bit-reproducible, with and without a
    value-projection job in the launch."""
    B, C = 2, 256
    pred = _perturbed_predictor(13)
    x = (syn.det_randn(f"cb.x{n}", (B, n, C)) * 1.2).to(DEV)
    coarse = syn.det_randn(f"cb.c{n}", (B, 1, (hw[0] + 1) // 2, (hw[1] + 1) // 2)).to(DEV)
    alpha = torch.tensor([0.25], device=DEV)
    tokens = syn.det_randn("cb.tok", (B, 900, C)).to(DEV).to(torch.bfloat16)
    w = (syn.det_randn("cb.w", (2 * 8 * 32, C)) * 0.05).to(DEV).to(torch.bfloat16)
    old = F.CONST_IN_BLOCK, F.CONST_IN_BLOCK_ROWS
    try:
        with torch.no_grad():
            hh = F.salience_head_hoist(x, pred)
            kw = dict(coarse_score=coarse, level_hw=hw, alpha=alpha, hoisted=hh)
            F.CONST_IN_BLOCK = False
            min_a = torch.zeros(1, device=DEV)
            want = F.salience_head(x, pred, score_min=min_a, **kw)
            F.CONST_IN_BLOCK, F.CONST_IN_BLOCK_ROWS = True, 160
            for with_value in (False, True):
                min_b = torch.full((1,), -7.0, device=DEV)
                vjob = None
                if with_value:
                    maps, (vjob,) = F.plan_value_projection(tokens, w, None, None, 8, 2, torch.float16, parts=1)
                got = F.salience_head(x, pred, score_min=min_b, value_job2=vjob, **kw)
                assert (got - want).abs().max().item() <= 2e-6 * (want.abs().max().item() + 1.0)
                assert min_b.item() == got.min().item() and abs(min_b.item() - min_a.item()) <= 2e-6 * (abs(min_a.item()) + 1.0)
                for _ in range(3):
                    assert torch.equal(F.salience_head(x, pred, **kw), got)
                if with_value:
                    assert vjob.done and torch.equal(maps, F.value_proj_head_major(tokens, w, None, None, 8, 2, torch.float16))
    finally:
        F.CONST_IN_BLOCK, F.CONST_IN_BLOCK_ROWS = old
