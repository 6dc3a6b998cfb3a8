"""CPU: rows N1 + N2 -- the oracle's whole-transformer restatement against tests/golden/transformer_small.npz.

Everything up to the NMS inputs (memory, class logits of all tokens, top-k scores, unit boxes, category ids) is the
imported reference's own computation.  torchvision is absent from this image, so the fixture's NMS was served by the
oracle's restated greedy NMS: the post-NMS vectors pin the reference's *surrounding* code, not torchvision itself
("NMS parity unpinned").  The grid-structure test below checks the restated NMS independently.
"""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import synthetic as syn

G = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "transformer_small.npz"))


def reference_keys_state_dict(d, template):
    """Name-seeded weights for ``template`` (any state_dict with the reference's key names), CRC-checked."""
    heads = int(d["hyper"][1])
    sd = syn.det_state_dict(template, num_heads=heads, num_levels=4, num_points=4)
    assert sorted(sd) == d["sd_keys"].tolist()
    assert [zlib.crc32(sd[k].numpy().tobytes()) for k in sorted(sd)] == d["sd_crc"].tolist()
    return sd


def build_product_transformer(d, neck_fixture=None):
    """The product transformer with the fixture's hyper-parameters and name-seeded weights; ``neck_fixture``
    (transformer_small_neck.npz) adds the RepVGGPluX neck and checks the weights against ITS key list."""
    from salience_detr_amd.salience_transformer import build_salience_transformer
    E, heads, d_ffn, enc_layers, dec_layers, classes, topk_sa, max_emb, proposals = d["hyper"].tolist()
    tr = build_salience_transformer(embed_dim=E, num_heads=heads, d_ffn=d_ffn, num_encoder_layers=enc_layers,
                                    num_decoder_layers=dec_layers, num_classes=classes, topk_sa=topk_sa,
                                    max_num_embedding=max_emb, two_stage_num_proposals=proposals,
                                    level_filter_ratio=tuple(d["level_ratio"].tolist()),
                                    layer_filter_ratio=tuple(d["layer_ratio"].tolist()),
                                    with_neck=neck_fixture is not None)
    if neck_fixture is not None:
        keyed = dict(hyper=d["hyper"], sd_keys=neck_fixture["sd_keys"], sd_crc=neck_fixture["sd_crc"])
        sd = reference_keys_state_dict(keyed, tr.state_dict())
    else:
        sd = reference_keys_state_dict(d, tr.state_dict())
    tr.load_state_dict(sd)
    return tr.eval(), sd


def inputs(d):
    return ([_t(d[f"feat{l}"]) for l in range(4)], [_t(d[f"mask{l}"]) for l in range(4)],
            [_t(d[f"pos{l}"]) for l in range(4)])


def test_oracle_transformer_matches_reference(gold):
    d = gold
    _, sd = build_product_transformer(d)
    E, heads, d_ffn, enc_layers, dec_layers, classes, topk_sa, max_emb, proposals = d["hyper"].tolist()
    feats, masks, pos = inputs(d)
    out = R.transformer(sd, feats, masks, pos, proposals, heads=heads, topk_sa=topk_sa, enc_layers=enc_layers,
                        dec_layers=dec_layers)
    assert (out["memory"] - _t(d["memory"])).abs().max() < 2e-4
    assert (out["class_all"] - _t(d["class_all"])).abs().max() < 2e-4
    # NMS inputs: reference-computed
    boxes, idxs, _ = R.nms_inputs(out["topk_index"], _t(d["spatial_shapes"]), _t(d["level_start_index"]))
    ref_scores = _t(d["nms_scores"])
    assert (out["topk_scores"].reshape(-1) - ref_scores).abs().max() < 2e-4
    # tokens that are padding / outside (0.01, 0.99) all carry the same constant score: the order inside such a tie
    # group is torch.topk's (unspecified) in the fixture and lower-index-first in the oracle; everything else is exact
    untied = torch.ones_like(ref_scores, dtype=torch.bool)
    untied[1:] &= ref_scores[1:] != ref_scores[:-1]
    untied[:-1] &= ref_scores[:-1] != ref_scores[1:]
    assert untied.sum() >= 60
    assert torch.equal(boxes[untied], _t(d["nms_boxes"])[untied]) and torch.equal(idxs[untied], _t(d["nms_idxs"])[untied])
    assert float(d["nms_thr"]) == 0.3
    # after the (restated) NMS
    assert (out["enc_outputs_class"] - _t(d["enc_outputs_class"])).abs().max() < 2e-4
    assert (out["enc_outputs_coord"] - _t(d["enc_outputs_coord"])).abs().max() < 2e-5
    assert (out["enc_outputs_coord"] - _t(d["reference_points"])).abs().max() < 2e-5
    assert (out["outputs_classes"] - _t(d["outputs_classes"])).abs().max() < 5e-4
    assert (out["outputs_coords"] - _t(d["outputs_coords"])).abs().max() < 5e-5
    for l in range(4):
        assert (out["salience_score"][l] - _t(d[f"salience{l}"])).abs().max() < 2e-4


def grid_greedy(order_tokens, shapes, lsi, neighbourhood=4):
    """Independent statement of what the NMS does on the proposal boxes: cells are 2x2 boxes on the integer grid, so
    IoU is 1/3 for edge neighbours, 1/7 for diagonal ones, 0 beyond -- at threshold 0.3 a token is dropped iff an
    already kept token is its edge neighbour on the same level."""
    kept, kept_set = [], set()
    for t in order_tokens:
        lvl = int((t >= lsi).sum()) - 1
        h, w = shapes[lvl]
        y, x = divmod(t - int(lsi[lvl]), int(w))
        nb = [(x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)]
        if neighbourhood == 8:
            nb += [(x - 1, y - 1), (x + 1, y - 1), (x - 1, y + 1), (x + 1, y + 1)]
        if any((lvl, a, b) in kept_set for a, b in nb):
            continue
        kept.append(t)
        kept_set.add((lvl, x, y))
    return kept


@pytest.mark.parametrize("thr,nb", [(0.3, 4), (0.1, 8), (0.5, 0)])
def test_restated_nms_equals_grid_neighbour_suppression(thr, nb):
    shapes = torch.tensor([(13, 17), (7, 9), (4, 5), (2, 3)])
    sizes = shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    S = int(sizes.sum())
    g = torch.Generator().manual_seed(7)
    score = torch.randn(2, S, generator=g)
    score[0, 5:40] = 0.5                      # ties: list order decides
    k = 200
    ts, ti = R.topk_desc_stable(score, k)
    got = R.nms_on_topk_index(ts, ti, shapes, lsi, num_proposals=k, iou_threshold=thr)
    per_image = [grid_greedy(ti[b].tolist(), shapes.tolist(), lsi, nb) if nb else ti[b].tolist() for b in range(2)]
    n = min(len(p) for p in per_image)
    assert got.shape == (2, n)
    for b in range(2):
        assert got[b].tolist() == per_image[b][:n]


def test_oracle_transformer_with_neck_matches_reference(gold):
    """Row N3 in place: the reference transformer WITH its RepVGGPluX neck (transformer_small_neck.npz)."""
    d, dn = gold, np.load(os.path.join(G, "transformer_small_neck.npz"))
    _, sd = build_product_transformer(d, dn)
    assert any(k.startswith("neck.") for k in sd)
    E, heads, d_ffn, enc_layers, dec_layers, classes, topk_sa, max_emb, proposals = d["hyper"].tolist()
    feats, masks, pos = inputs(d)
    out = R.transformer(sd, feats, masks, pos, proposals, heads=heads, topk_sa=topk_sa, enc_layers=enc_layers,
                        dec_layers=dec_layers)
    assert (out["memory"] - _t(dn["memory_neck"])).abs().max() < 2e-4
    assert (out["class_all"] - _t(dn["class_all"])).abs().max() < 5e-4
    assert (out["enc_outputs_class"] - _t(dn["enc_outputs_class"])).abs().max() < 5e-4
    assert (out["enc_outputs_coord"] - _t(dn["enc_outputs_coord"])).abs().max() < 2e-5
    assert (out["outputs_classes"] - _t(dn["outputs_classes"])).abs().max() < 2e-3
    assert (out["outputs_coords"] - _t(dn["outputs_coords"])).abs().max() < 2e-4


# ---- the restated torchvision NMS beyond the grid special case: random NON-grid boxes, several
# thresholds, against an independent brute-force statement -- full pairwise IoU matrix in numpy, then the greedy sweep
# "keep the best remaining box, drop everything it overlaps by more than thr".  (torchvision itself is not in this image:
# `nms_greedy` stays labelled parity-unpinned; this pins the ALGORITHM it restates.)
def _brute_force_nms(boxes, scores, thr):
    import numpy as np
    b = boxes.numpy().astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(b[:, None, :2], b[None, :, :2])
    rb = np.minimum(b[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    iou = inter / (area[:, None] + area[None, :] - inter)
    order = sorted(range(len(b)), key=lambda i: (-float(scores[i]), i))     # stable descending
    alive = np.ones(len(b), dtype=bool)
    keep = []
    for i in order:
        if not alive[i]:
            continue
        keep.append(i)
        alive &= ~(iou[i] > np.float32(thr))
        alive[i] = False
    return keep


@pytest.mark.parametrize("thr", [0.1, 0.3, 0.5, 0.7])
@pytest.mark.parametrize("n,seed", [(1, 0), (17, 1), (300, 2), (900, 3)])
def test_restated_nms_equals_brute_force_on_random_boxes(thr, n, seed):
    g = torch.Generator().manual_seed(seed)
    centre = torch.rand(n, 2, generator=g) * 100
    size = torch.rand(n, 2, generator=g) ** 2 * 40 + 1          # from tiny to large, heavy overlap
    boxes = torch.cat([centre - size / 2, centre + size / 2], -1)
    scores = torch.rand(n, generator=g)
    if n > 20:
        scores[5] = scores[9]                                   # a tie: the earlier index wins
    got = R.nms_greedy(boxes, scores, thr).tolist()
    assert got == _brute_force_nms(boxes, scores, thr)


@pytest.mark.parametrize("thr", [0.3, 0.6])
def test_restated_batched_nms_is_per_category_nms(thr):
    g = torch.Generator().manual_seed(7)
    n = 400
    centre = torch.rand(n, 2, generator=g) * 60
    size = torch.rand(n, 2, generator=g) * 25 + 2
    boxes = torch.cat([centre - size / 2, centre + size / 2], -1)
    scores = torch.rand(n, generator=g)
    idxs = torch.randint(0, 5, (n,), generator=g)
    got = R.batched_nms(boxes, scores, idxs, thr).tolist()
    expect = []
    for c in range(5):
        members = (idxs == c).nonzero().flatten()
        expect += [int(members[i]) for i in _brute_force_nms(boxes[members], scores[members], thr)]
    expect.sort(key=lambda i: (-float(scores[i]), i))
    assert got == expect
