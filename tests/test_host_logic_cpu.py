"""CPU: host-side logic of the product package that needs no device -- pyramid plumbing, position
embeddings, token budgets (host arithmetic == the reference's float32 truncation), checkpoint keys."""
import os

import numpy as np
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import pyramid
from salience_detr_amd import synthetic as syn

G = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("tag", ["single", "mixed"])
def test_pyramid_plumbing_matches_golden(tag):
    d = np.load(os.path.join(G, f"hotpath_small_{tag}.npz"))
    masks = [_t(d[f"mask{l}"]) for l in range(4)]
    feats = [_t(d[f"feat{l}"]) for l in range(4)]
    pos = [_t(d[f"pos{l}"]) for l in range(4)]
    shapes, lsi, vr = pyramid.multi_level_misc(masks)
    assert torch.equal(shapes, _t(d["spatial_shapes"])) and torch.equal(lsi, _t(d["level_start_index"]))
    assert (vr - _t(d["valid_ratios"])).abs().max() < 1e-7
    assert torch.equal(pyramid.flatten_multi_level(feats), _t(d["feat_flatten"]))
    assert torch.equal(pyramid.flatten_multi_level(masks), _t(d["mask_flatten"]))
    lp = pyramid.get_lvl_pos_embed(_t(d["sd.level_embeds"]), pos)
    assert (lp - _t(d["lvl_pos_embed_flatten"])).abs().max() < 1e-6
    E = int(d["hyper"][0])
    pe = lambda mask: syn.sine_position_embedding(mask, E // 2)
    for l in range(4):
        assert (pe(masks[l]) - pos[l]).abs().max() < 2e-6
    lin = torch.nn.Linear(E, E)
    ln = torch.nn.LayerNorm(E)
    lin.load_state_dict({"weight": _t(d["sd.enc_output.weight"]), "bias": _t(d["sd.enc_output.bias"])})
    ln.load_state_dict({"weight": _t(d["sd.enc_output_norm.weight"]), "bias": _t(d["sd.enc_output_norm.bias"])})
    with torch.no_grad():
        bom = pyramid.encoder_output_memory(lin, ln, _t(d["feat_flatten"]) + _t(d["lvl_pos_embed_flatten"]),
                                            _t(d["mask_flatten"]), pyramid.level_shapes_of(masks))
    assert (bom - _t(d["backbone_output_memory"])).abs().max() < 2e-5


@pytest.mark.parametrize("sizes", [[(800, 1333)], [(800, 1333), (800, 1066)], [(640, 480), (333, 500), (512, 512)],
                                   [(64, 96), (48, 80)], [(1, 1)], [(37, 41), (800, 1201)]])
def test_host_budgets_equal_device_rule(sizes):
    canvas = syn.pad_to_32(max(s[0] for s in sizes), max(s[1] for s in sizes))
    _, masks = syn.make_masks(sizes)
    ratio = torch.tensor([0.4, 0.8, 1.0, 1.0])
    focus_ref, level_ref, valid_ref = R.token_budgets(masks, ratio)
    focus, level, valid = pyramid.host_token_budgets(sizes, canvas, pyramid.level_shapes_of(masks), ratio.tolist())
    assert valid.tolist() == valid_ref.tolist()
    assert focus.tolist() == focus_ref.tolist()
    assert level.tolist() == level_ref.tolist()


def test_benchmark_shape_constants():
    """800x1333 -> levels, Nv, per-level top-k, per-layer query counts."""
    _, masks = syn.make_masks([(800, 1333)])
    shapes = pyramid.level_shapes_of(masks)
    assert shapes == [(100, 168), (50, 84), (25, 42), (13, 21)]
    focus, level, valid = pyramid.host_token_budgets([(800, 1333)], (800, 1344), shapes, (0.4, 0.8, 1.0, 1.0))
    assert valid.tolist() == [[16700, 4200, 1050, 273]] and level.tolist() == [6680, 3360, 1050, 273]
    assert focus.tolist() == [11363]
    assert pyramid.layer_token_counts(11363, (1.0, 0.8, 0.6, 0.6, 0.4, 0.2)) == [11363, 9090, 6817, 6817, 4545, 2272]


def test_reference_points_match_oracle():
    from salience_detr_amd.salience_encoder import SalienceTransformerEncoder
    _, masks = syn.make_masks([(64, 96), (48, 80)])
    shapes, lsi, vr = pyramid.multi_level_misc(masks)
    got = SalienceTransformerEncoder.get_reference_points(shapes, vr, device="cpu")
    assert (got - R.encoder_reference_points(shapes, vr)).abs().max() < 1e-6


def test_learned_embedding_table_and_limits():
    pe = pyramid.PositionEmbeddingLearned(20, 16)
    m = torch.zeros(2, 5, 7, dtype=torch.bool)
    full = pe(m)
    assert full.shape == (2, 32, 5, 7)
    assert torch.equal(full[0].flatten(1).t(), pe.flat([(5, 7)]))
    with pytest.raises(IndexError):
        pe.flat([(21, 3)])


def test_nms_neighbourhood_follows_the_fp32_iou_comparisons():
    """Row N1: which grid neighbours the reference's 2x2 unit boxes suppress is decided by two fp32 comparisons."""
    from salience_detr_amd.filter_ops import nms_neighbourhood
    assert nms_neighbourhood(0.3) == 4                      # the reference's threshold: edge neighbours (IoU 1/3)
    assert nms_neighbourhood(0.1) == 8                      # below 1/7: corner neighbours as well
    assert nms_neighbourhood(0.5) == 0                      # above 1/3: nothing but exact duplicates
    import numpy as np
    third = float(np.float32(2) / np.float32(6))
    assert nms_neighbourhood(third) == 0 and nms_neighbourhood(float(np.nextafter(np.float32(third), np.float32(0)))) == 4
    seventh = float(np.float32(1) / np.float32(7))
    assert nms_neighbourhood(seventh) == 4 and nms_neighbourhood(float(np.nextafter(np.float32(seventh), np.float32(0)))) == 8


def test_product_modules_keep_the_reference_state_dict_names():
    """Rows N1 / N2 / N4: checkpoints of the reference load by name (parameter names are part of the drop-in surface)."""
    try:
        from salience_detr_amd.salience_transformer import build_salience_transformer
        tr = build_salience_transformer(num_encoder_layers=1, num_decoder_layers=1, d_ffn=32, num_classes=3,
                                        two_stage_num_proposals=5, layer_filter_ratio=(1.0,))
    except RuntimeError as e:           # the HIP library is built by __graft_entry__.build(); without it construction fails loudly
        pytest.skip(str(e))
    keys = set(tr.state_dict())
    for k in ("level_embeds", "alpha", "tgt_embed.weight", "enc_output.weight", "enc_output_norm.bias",
              "encoder_class_head.bias", "encoder_bbox_head.layers.2.weight", "enc_mask_predictor.layer1.0.weight",
              "encoder.layers.0.self_attn.sampling_offsets.weight", "encoder.layers.0.pre_attention.in_proj_weight",
              "encoder.enhance_mcsp.weight", "encoder.background_embedding.row_embed.weight",
              "decoder.layers.0.cross_attn.value_proj.weight", "decoder.layers.0.self_attn.out_proj.bias",
              "decoder.ref_point_head.layers.1.bias", "decoder.class_head.0.weight", "decoder.bbox_head.0.layers.0.weight",
              "decoder.norm.weight", "level_filter_ratio", "layer_filter_ratio"):
        assert k in keys, k


def test_static_tensor_cache_is_lru_not_clear_all():
    """a full clear() would free tensors whose addresses a captured hipGraph still holds."""
    saved, limit = dict(pyramid._STATIC), pyramid._STATIC_LIMIT
    try:
        pyramid._STATIC.clear()
        pyramid._STATIC_LIMIT = 4
        a = pyramid.static_tensor("a", lambda: torch.zeros(1))
        for k in "bcd":
            pyramid.static_tensor(k, lambda: torch.zeros(1))
        assert pyramid.static_tensor("a", lambda: torch.ones(1)) is a        # hit, becomes most recent
        pyramid.static_tensor("e", lambda: torch.zeros(1))                  # evicts ONE entry: the oldest ("b")
        assert set(pyramid._STATIC) == {"a", "c", "d", "e"}
        assert pyramid.static_tensor("a", lambda: torch.ones(1)) is a
    finally:
        pyramid._STATIC.clear()
        pyramid._STATIC.update(saved)
        pyramid._STATIC_LIMIT = limit


def test_init_weights_bumps_versions_and_invalidate_caches():
    """the derived-weight caches are keyed on parameter versions; init_weights must not write through
    `.data` (which leaves the version unchanged), and `invalidate_caches` clears what a `.data` write would leave stale."""
    from salience_detr_amd.ms_deform_attn import MultiScaleDeformableAttention, invalidate_caches
    m = MultiScaleDeformableAttention(32, 4, 4, 4)
    before = {n: p._version for n, p in m.named_parameters()}
    m.init_weights()
    for n, p in m.named_parameters():
        if n != "sampling_offsets.bias":          # (re-created as a fresh Parameter by init_weights)
            assert p._version > before[n], n
    w1, _ = m._fused_query_projection()
    with torch.no_grad():
        m.attention_weights.weight.data.add_(1.0)          # the unsupported kind of write: version unchanged ...
    assert m._fused_query_projection()[0] is w1             # ... so the cache is (knowingly) stale
    invalidate_caches(m)
    w2, _ = m._fused_query_projection()
    assert w2 is not w1 and torch.equal(w2[-m.attention_weights.weight.shape[0]:], m.attention_weights.weight)
    with torch.no_grad():
        m.attention_weights.weight.add_(1.0)                # the supported kind: version bump -> automatic refresh
    assert m._fused_query_projection()[0] is not w2


def test_x3_linear_swap_keeps_parameters_and_falls_back_on_cpu():
    """`use_x3_linear_` switches the class of plain nn.Linear modules only (state-dict keys and parameters untouched,
    the out_proj of nn.MultiheadAttention left alone); without a HIP tensor the forward / backward are F.linear's."""
    from salience_detr_amd import linear_x3 as X
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8),
                                torch.nn.MultiheadAttention(8, 2, batch_first=True))
    ref = [p.detach().clone() for p in model.parameters()]
    keys = list(model.state_dict().keys())
    assert X.use_x3_linear_(model) == 2
    assert isinstance(model[0], X.X3Linear) and isinstance(model[2], X.X3Linear)
    assert type(model[3].out_proj) is not X.X3Linear
    assert list(model.state_dict().keys()) == keys
    assert all(torch.equal(a, b) for a, b in zip(ref, model.parameters()))
    x = torch.randn(5, 16, requires_grad=True)
    y = model[2](model[1](model[0](x)))
    y.sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x2, ref[0], ref[1])), ref[2], ref[3])
    y2.sum().backward()
    assert torch.equal(y, y2) and torch.equal(x.grad, x2.grad)
    # the split of dw's token reduction: never more slices than 256-token pieces, more for fewer output tiles
    assert X._weight_grad_splits(100, 256, 256) == 1
    assert X._weight_grad_splits(22726, 256, 256) > X._weight_grad_splits(22726, 2048, 256) >= 1
    assert X._weight_grad_splits(22726, 256, 256) <= (22726 + 255) // 256


def test_graph_lanes_input_structure_checks_and_device_requirement():
    """Host side of graph_lanes: static-input bookkeeping (structure, shape and dtype must match the example the lanes
    were captured with) and the refusal to run without a HIP device (no CPU fallback)."""
    from salience_detr_amd import graph_lanes as GL
    a = ([torch.zeros(2, 3), torch.zeros(4)], [torch.zeros(2, dtype=torch.bool)])
    clone = GL._map(a, lambda t: t.clone())
    assert isinstance(clone, tuple) and isinstance(clone[0], list) and clone[0][0] is not a[0][0]
    src = ([torch.ones(2, 3), torch.ones(4)], [torch.ones(2, dtype=torch.bool)])
    GL._zip_apply(clone, src, lambda d, s: d.copy_(s))
    assert clone[0][0].sum() == 6 and bool(clone[1][0].all())
    with pytest.raises(ValueError):
        GL._zip_apply(clone, ([torch.ones(2, 3)], [torch.ones(2, dtype=torch.bool)]), lambda d, s: None)   # a tensor short
    with pytest.raises(ValueError):
        GL._zip_apply(clone, ([torch.ones(3, 3), torch.ones(4)], [torch.ones(2, dtype=torch.bool)]), lambda d, s: None)
    with pytest.raises(ValueError):
        GL._zip_apply(clone, ([torch.ones(2, 3), torch.ones(4)], [torch.ones(2)]), lambda d, s: None)      # dtype
    with pytest.raises(TypeError):
        GL._map({"x": torch.zeros(1)}, lambda t: t)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            GL.GraphLanes(lambda x: x, (torch.zeros(1),), lanes=1)
    with pytest.raises(ValueError):
        GL.GraphLanes(lambda x: x, (torch.zeros(1),), lanes=0)


def test_bordered_layout_and_tile_orders_host_logic():
    """The bordered value-map layout (`ms_deform_attn.BorderedLayout`, the host side of `sdetr_bordered_layout`) and the
    tile-major row orders, without a GPU: record count as the library's own host function gives it, pixel records and zero
    records partition the map, `to_bordered` puts every token where the kernel will look for it with zeros all round,
    tile positions are a permutation that keeps a tile's tokens together, `spatial_row_order` sorts rows by them."""
    import ctypes
    import torch
    from salience_detr_amd import _hip, ms_deform_attn as M
    for levels in ([(100, 168), (50, 84), (25, 42), (13, 21)], [(7, 5), (4, 3), (2, 2), (1, 1)],
                   [(200, 336), (100, 168), (50, 84), (25, 42)]):
        lay = M.bordered_layout(levels)
        want = sum((h + 2) * (w + 1) for h, w in levels) + 1
        hw = (ctypes.c_int32 * 8)(*[v for hw_ in levels for v in hw_])
        assert lay.records == want == _hip.lib().sdetr_msda_bordered_records(hw, 4)
        assert lay.tokens == sum(h * w for h, w in levels)
        pix, border = lay.pixel_map_host.long(), lay.border_host.long()
        assert pix.numel() == lay.tokens and pix.unique().numel() == pix.numel()
        both = torch.cat([pix, border])
        assert both.unique().numel() == both.numel() == lay.records      # a partition of the records
        # token (level l, y, x) sits at start_l + (y + 1) (W_l + 1) + x + 1: its left neighbour is the token (y, x - 1) or a
        # zero record, the record above is (y - 1, x) or a zero record -- what lets the kernel read all four corners blindly
        t0 = 0
        for (h, w), start in zip(levels, lay.starts):
            grid = pix[t0:t0 + h * w].view(h, w)
            assert grid[0, 0].item() == start + (w + 1) + 1
            if w > 1:
                assert torch.equal(grid[:, 1:], grid[:, :-1] + 1)
            if h > 1:
                assert torch.equal(grid[1:, :], grid[:-1, :] + (w + 1))
            bset = set(border.tolist())
            assert all(int(grid[y, 0]) - 1 in bset for y in range(h))                 # record -1 of every row
            assert all(int(grid[0, x]) - (w + 1) in bset for x in range(w))           # row -1
            assert all(int(grid[h - 1, x]) + (w + 1) in bset for x in range(w))       # row H
            assert int(grid[h - 1, w - 1]) + 1 in bset                                # (y, W): the next row's record -1
            t0 += h * w
        assert lay.records - 1 in set(border.tolist())                                # the closing record
        v = torch.arange(1, lay.tokens * 2 + 1, dtype=torch.float32).view(1, lay.tokens, 2)
        b = M.to_bordered(v, levels)
        assert b.shape == (1, lay.records, 2) and M.is_bordered(b, levels) and not M.is_bordered(v, levels)
        assert torch.equal(b[0, pix], v[0]) and float(b[0, border].abs().sum()) == 0.0
        with pytest.raises(RuntimeError):
            M.to_bordered(v[:, :-1], levels)
        for tile in (8, 16):
            pos = M.tile_major_positions(levels, tile).long()
            assert pos.numel() == lay.tokens and torch.equal(pos.sort()[0], torch.arange(lay.tokens))
            # the tokens (of all levels) whose centre falls into one cell of the ceil(H0 / tile) x ceil(W0 / tile) grid occupy
            # ONE contiguous range of positions
            h0, w0 = levels[0]
            ty, tx = max(1, -(-h0 // tile)), max(1, -(-w0 // tile))
            cells = []
            for h, w in levels:
                cy = ((torch.arange(h, dtype=torch.float64) + 0.5) / h * ty).floor().clamp(max=ty - 1).long()
                cx = ((torch.arange(w, dtype=torch.float64) + 0.5) / w * tx).floor().clamp(max=tx - 1).long()
                cells.append((cy.view(h, 1) * tx + cx.view(1, w)).reshape(-1))
            cells = torch.cat(cells)
            for c in cells.unique().tolist()[:40]:
                mine = pos[cells == c]
                assert int(mine.max() - mine.min()) + 1 == mine.numel(), (tile, c)
        idx = torch.stack([torch.randperm(lay.tokens, generator=torch.Generator().manual_seed(s))[:min(50, lay.tokens)]
                           for s in (1, 2)])
        order = M.spatial_row_order(idx, levels, 16)
        assert order.dtype == torch.int32 and order.shape == idx.shape
        p16 = M.tile_major_positions(levels, 16).long()
        for r in range(2):
            seq = p16[idx[r][order[r].long()]]
            assert torch.equal(seq, seq.sort()[0]) and torch.equal(order[r].long().sort()[0], torch.arange(idx.shape[1]))


def test_cut_tie_canonicalisation_only_touches_ties():
    """tests/cut_ties.py (used by the full-size GPU digests): a cut neighbourhood that differs from the reference's by an
    exchange of scores within the tolerance is rewritten in the reference's order; a real difference, a token from outside
    the window, or a window inside an image's padded tail is left alone."""
    import cut_ties
    n, W = 40, 4
    ref = torch.arange(100, 100 + n).view(1, n).repeat(3, 1)                    # the reference's sorted list, 3 images
    score = torch.linspace(1.0, 0.0, n).view(1, n).repeat(3, 1).clone()
    n_k = 20
    score[:, n_k - 1] = score[:, n_k] + 3e-7                                      # a tie at the cut
    rec = [(ref[:, n_k - W:n_k + W].numpy(), score[:, n_k - W:n_k + W].numpy(), (n_k - W, n_k))]
    build = ref.clone()
    build[0, n_k - 1], build[0, n_k] = ref[0, n_k], ref[0, n_k - 1]              # image 0: the tie changed sides
    build[1, n_k - 2], build[1, n_k + 1] = ref[1, n_k + 1], ref[1, n_k - 2]      # image 1: a real exchange (scores 0.08 apart)
    build[2, n_k - 1] = 999                                                       # image 2: a token from elsewhere
    views = [build, build[:, :n_k]]
    out, changed = cut_ties.canonical_foreground_inds(views, [n, n, n], rec)
    assert changed == [(0, 0, [int(ref[0, n_k - 1]), int(ref[0, n_k])])]
    assert torch.equal(out[0][0], ref[0]) and torch.equal(out[0][1], build[1]) and torch.equal(out[0][2], build[2])
    assert out[1].data_ptr() == out[0].data_ptr() and out[1].shape == (3, n_k)   # still prefixes of one list
    # a window that reaches beyond the image's valid count: untouched
    out2, changed2 = cut_ties.canonical_foreground_inds(views, [n_k + 1, n, n], rec)
    assert not changed2 and torch.equal(out2[0], build)


def test_round6_launch_fusions_are_gated_by_shape_and_fail_loudly_without_a_gpu():
    """filter_ops.topk_select_inproj / encoder_prepare_sorted(class_head=...) / the second pass's class score: the gates
    decline shapes the kernels do not cover (the callers then take the separate launches), and the operators themselves
    raise on CPU tensors -- no fallback."""
    import pytest
    from salience_detr_amd import filter_ops as F
    mha = torch.nn.MultiheadAttention(256, 8, batch_first=True).to(torch.bfloat16)
    norm = torch.nn.LayerNorm(256).to(torch.bfloat16)
    head = torch.nn.Linear(256, 91).to(torch.bfloat16)
    q = torch.zeros(2, 4000, 256, dtype=torch.bfloat16)
    score = torch.zeros(2, 4000)
    # CPU tensors: never
    assert not F.topk_select_inproj_applies(score, 300, q, q, mha, norm)
    with pytest.raises(RuntimeError):
        F.topk_select_inproj(score, 300, q, q, mha)
    with pytest.raises(RuntimeError):
        F.encoder_prepare_sorted(q, q, score, torch.zeros(2, 10, dtype=torch.int64), torch.ones(2, 4, 2),
                                 torch.ones(4, 2, dtype=torch.int64), torch.zeros(4, dtype=torch.int64), class_head=head)
    # the class-score gate of the entry gather looks at dtypes and shapes only
    assert F.prepare_class_score_applies(q, score, head)
    assert not F.prepare_class_score_applies(q.float(), score, head)                     # fp32 tokens: the class head's own launch
    assert not F.prepare_class_score_applies(q, None, head)                              # no foreground score
    assert not F.prepare_class_score_applies(q, score, torch.nn.Linear(256, 120).to(torch.bfloat16))   # > 96 classes
    assert not F.prepare_class_score_applies(q, score, torch.nn.Linear(256, 91))         # head in another dtype
    old = F.PREPARE_WITH_CLASS_SCORE
    try:
        F.PREPARE_WITH_CLASS_SCORE = False
        assert not F.prepare_class_score_applies(q, score, head)
    finally:
        F.PREPARE_WITH_CLASS_SCORE = old
    # the three switches exist and default to the fused forms
    assert F.SPLIT_PASS_CLASS_SCORE and F.SELECT_WITH_INPROJECTION and F.PREPARE_WITH_CLASS_SCORE
