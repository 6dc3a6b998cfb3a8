"""GPU: row N1 (two-stage proposal selection: proposal geometry, top-k, NMS, heads on the survivors) and the whole
``SalienceTransformer.forward`` (rows N1 + N2 on top of the encoder hot path).

Index work is compared bit-exactly with the oracle; floating point at 1e-3 (logits) / 1e-4 (boxes) against the
reference's fixture.  The NMS itself is pinned to the oracle's restated torchvision algorithm (torchvision is absent
here; see tests/test_transformer_cpu.py).
"""
import os

import numpy as np
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import synthetic as syn
from salience_detr_amd.filter_ops import (encoder_output_proposals, grid_nms_topk, nms_neighbourhood, proposal_refine)
from test_transformer_cpu import _t, build_product_transformer, inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
FULL_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "transformer_small.npz"))


def _levels(shapes):
    t = torch.tensor(shapes, dtype=torch.int64)
    sizes = t.prod(1)
    return t, torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]]), int(sizes.sum())


@pytest.mark.parametrize("image_sizes", [[(800, 1333)], [(800, 1333), (800, 1066), (401, 640)], [(64, 96), (48, 80)]])
def test_proposal_geometry_matches_oracle(image_sizes):
    _, masks = syn.make_masks(image_sizes)
    shapes = [tuple(m.shape[-2:]) for m in masks]
    mask_flat = R.flatten_levels(masks)
    E = 8
    sd = {"enc_output.weight": torch.eye(E), "enc_output.bias": torch.zeros(E),
          "enc_output_norm.weight": torch.ones(E), "enc_output_norm.bias": torch.zeros(E)}
    mem = torch.ones(len(image_sizes), mask_flat.shape[1], E)
    _, want = R.encoder_output_proposals(sd, mem, mask_flat, torch.tensor(shapes))
    keep, logit = encoder_output_proposals(mask_flat.cuda(), shapes)
    want_keep = torch.isfinite(want).all(-1)
    assert torch.equal(keep.cpu(), want_keep)
    assert torch.equal(torch.isinf(logit.cpu()), torch.isinf(want))
    fin = want_keep[..., None].expand_as(want)
    assert (logit.cpu()[fin] - want[fin]).abs().max() < 2e-6


def _score_maps(kind, B, S, shapes, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "random":
        return torch.randn(B, S, generator=g)
    if kind == "ties":
        s = torch.randn(B, S, generator=g)
        return (s * 4).round() / 4                    # heavy ties: list order decides
    # "ramp": strictly increasing along every row of every level -- the longest possible suppression chains
    return torch.arange(S, dtype=torch.float32).repeat(B, 1) + torch.rand(B, 1, generator=g)


@pytest.mark.parametrize("kind", ["random", "ties", "ramp"])
@pytest.mark.parametrize("thr", [0.3, 0.1, 0.5])
def test_grid_nms_matches_restated_batched_nms(kind, thr):
    shapes = [(40, 67), (20, 34), (10, 17), (5, 9)]
    t_shapes, lsi, S = _levels(shapes)
    B, K, keep_n = 2, 900, 300
    ts, ti = R.topk_desc_stable(_score_maps(kind, B, S, shapes, 3), K)
    want = R.nms_on_topk_index(ts, ti, t_shapes, lsi, num_proposals=keep_n, iou_threshold=thr)
    kept, count = grid_nms_topk(ti.cuda(), shapes, S, thr, keep_n)
    n = min(int(count.min()), keep_n)
    assert n == want.shape[1]
    assert torch.equal(kept[:, :n].cpu(), want)
    assert nms_neighbourhood(thr) == {0.3: 4, 0.1: 8, 0.5: 0}[thr]


def test_grid_nms_full_size_counts_and_order():
    """3600 candidates per image on the 800x1333 pyramid: identical to the restated NMS, kept tokens in score order,
    and the unclamped counts equal the oracle's."""
    t_shapes, lsi, S = _levels(FULL_SHAPES)
    B, K = 2, 3600
    score = _score_maps("random", B, S, FULL_SHAPES, 11)
    score[:, :16700] += 1.0                                  # most candidates on the fine level, like real maps
    ts, ti = R.topk_desc_stable(score, K)
    boxes, idxs, image = R.nms_inputs(ti, t_shapes, lsi)
    kept_all = R.batched_nms(boxes, ts.reshape(-1), idxs, 0.3)
    per_image = [ti.reshape(-1)[kept_all[image[kept_all] == b]] for b in range(B)]
    kept, count = grid_nms_topk(ti.cuda(), FULL_SHAPES, S, 0.3, 900)
    assert count.cpu().tolist() == [int(p.shape[0]) for p in per_image]
    for b in range(B):
        assert torch.equal(kept[b].cpu(), per_image[b][:900])


@pytest.mark.parametrize("kind", ["random", "ramp"])
def test_grid_nms_pyramid_too_large_for_the_neighbour_cache(kind):
    """A 49 877-token pyramid: the u16 rank map (100 KB) and the states fit the LDS, the 16-byte neighbour records of the
    3600 candidates (round 5) no longer do -- the kernel form that re-derives the neighbourhoods every round must give
    the same answer (``ramp``: the longest suppression chains)."""
    shapes = [(150, 250), (75, 125), (38, 63), (19, 32)]
    t_shapes, lsi, S = _levels(shapes)
    assert 2 * S + 3600 <= 150 * 1024 < 2 * S + 17 * 3600
    B, K = 2, 3600
    score = _score_maps(kind, B, S, shapes, 17)
    if kind == "random":
        score[:, :37500] += 1.0
    ts, ti = R.topk_desc_stable(score, K)
    want = R.nms_on_topk_index(ts, ti, t_shapes, lsi, num_proposals=900, iou_threshold=0.3)
    kept, count = grid_nms_topk(ti.cuda(), shapes, S, 0.3, 900)
    n = min(int(count.min()), 900)
    assert n == want.shape[1]
    assert torch.equal(kept[:, :n].cpu(), want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_proposal_refine_matches_gather_sigmoid(dtype):
    B, S, n = 2, 500, 77
    logit = syn.det_randn("pr.l", (B, S, 4))
    logit[0, 3] = float("inf")
    index = torch.stack([torch.randperm(S, generator=torch.Generator().manual_seed(b))[:n] for b in range(B)])
    index[0, 0] = 3
    delta = syn.det_randn("pr.d", (B, n, 4)).to(dtype)
    got = proposal_refine(delta.cuda(), logit.cuda(), index.cuda())
    want = (delta.float() + logit.gather(1, index[..., None].expand(-1, -1, 4))).sigmoid()
    assert got.dtype == torch.float32 and (got.cpu() - want).abs().max() < 1e-6
    assert got[0, 0].cpu().tolist() == [1.0, 1.0, 1.0, 1.0]


def test_transformer_small_matches_reference_fixture(gold):
    d = gold
    tr, sd = build_product_transformer(d)
    tr = tr.cuda()
    feats, masks, pos = inputs(d)
    with torch.no_grad():
        out_cls, out_box, enc_cls, enc_box, sal = tr([f.cuda() for f in feats], [m.cuda() for m in masks],
                                                     [p.cuda() for p in pos], None, None, None)
    assert (enc_cls.cpu() - _t(d["enc_outputs_class"])).abs().max() < 1e-3
    assert (enc_box.cpu() - _t(d["enc_outputs_coord"])).abs().max() < 1e-4
    assert (out_cls.cpu() - _t(d["outputs_classes"])).abs().max() < 1e-3
    assert (out_box.cpu() - _t(d["outputs_coords"])).abs().max() < 1e-4
    for l in range(4):
        assert (sal[l].cpu() - _t(d[f"salience{l}"])).abs().max() < 1e-3


def test_transformer_with_neck_matches_reference_fixture(gold):
    """The whole forward with the RepVGGPluX neck between encoder and proposals, as in every reference config."""
    d, dn = gold, np.load(os.path.join(G, "transformer_small_neck.npz"))
    tr, sd = build_product_transformer(d, dn)
    tr = tr.cuda()
    feats, masks, pos = inputs(d)
    with torch.no_grad():
        out_cls, out_box, enc_cls, enc_box, sal = tr([f.cuda() for f in feats], [m.cuda() for m in masks],
                                                     [p.cuda() for p in pos], None, None, None)
    assert (enc_cls.cpu() - _t(dn["enc_outputs_class"])).abs().max() < 1e-3
    assert (enc_box.cpu() - _t(dn["enc_outputs_coord"])).abs().max() < 1e-4
    assert (out_cls.cpu() - _t(dn["outputs_classes"])).abs().max() < 2e-3
    assert (out_box.cpu() - _t(dn["outputs_coords"])).abs().max() < 2e-4


def test_proposal_stage_on_reference_memory(gold):
    """Row N1 alone, fed the reference's own ``memory``: the selected tokens (after top-k + NMS) are the oracle's,
    bit for bit, and their class / box outputs the fixture's."""
    d = gold
    tr, sd = build_product_transformer(d)
    tr = tr.cuda()
    proposals = int(d["hyper"][8])
    shapes = [tuple(r) for r in d["level_shapes"].tolist()]
    memory, mask = _t(d["memory"]), _t(d["mask_flatten"])
    want = R.two_stage_proposals(sd, memory, mask, _t(d["spatial_shapes"]), _t(d["level_start_index"]), proposals)
    with torch.no_grad():
        om, logit = tr.gen_encoder_output_proposals(memory.cuda(), mask.cuda(), shapes)
        best = tr.encoder_class_head(om).max(-1)[0]
        from salience_detr_amd.filter_ops import masked_topk_desc
        ts, ti = masked_topk_desc(best, min(4 * proposals, best.shape[1]))
        index = tr.nms_on_topk_index(ts, ti, shapes, None, 0.3)
        enc_cls, enc_box = tr.select_proposals(memory.cuda(), mask.cuda(), shapes)
    assert torch.equal(ti.cpu(), want["topk_index"])
    assert torch.equal(index.cpu(), want["index"])
    assert (enc_cls.cpu() - _t(d["enc_outputs_class"])).abs().max() < 1e-3
    assert (enc_box.cpu() - _t(d["enc_outputs_coord"])).abs().max() < 1e-4


def test_transformer_full_size_bf16_runs_and_agrees_with_fp32_selection():
    """Full configuration (800x1333 + 800x1066, 900 proposals, 6+6 layers): fp32 and bf16 runs pick largely the same
    proposals (the selection is a discontinuous function of the scores, so this is a statistical statement: >= 80% of
    the 900 tokens per image in common), outputs are finite and shaped like the reference's."""
    from salience_detr_amd.salience_transformer import build_salience_transformer
    image_sizes = [(800, 1333), (800, 1066)]
    tr = build_salience_transformer()
    tr.load_state_dict(syn.det_state_dict(tr.state_dict()))
    tr = tr.eval().cuda()
    _, masks = syn.make_masks(image_sizes)
    shapes = [tuple(m.shape[-2:]) for m in masks]
    feats = [f.cuda() for f in syn.make_feats(2, shapes, 256, 0)]
    pe = lambda mask: syn.sine_position_embedding(mask, 128)
    masks = [m.cuda() for m in masks]
    pos = [pe(m) for m in masks]
    with torch.no_grad():
        o32 = tr(feats, masks, pos)
        i32 = tr.last_proposal_index.cpu()
        tr.set_dtype(torch.bfloat16, torch.float16)
        o16 = tr(feats, masks, pos)
        i16 = tr.last_proposal_index.cpu()
    for o in (o32, o16):
        assert o[0].shape == (6, 2, 900, 91) and o[1].shape == (6, 2, 900, 4)
        assert o[2].shape == (2, 900, 91) and o[3].shape == (2, 900, 4)
        assert all(torch.isfinite(t.float()).all() for t in o[:4])
        assert (o[3] >= 0).all() and (o[3] <= 1).all()
    for b in range(2):
        common = len(set(i32[b].tolist()) & set(i16[b].tolist()))
        print("image", b, "proposals in common", common)
        assert common >= 0.8 * 900, common


def test_fp16_request_runs_config5_shape():
    """BASELINE.json configs[4]: "fp16, 900 queries" (the reference's --mixed-precision fp16, main.py:24-56).  Since
    round 5 a torch.float16 request runs IEEE-half activations (libsalience_hip_f16.so: the token-resident kernels built
    with -DSDETR_ACT_F16; fp32 accumulators, LayerNorm, softmax and class scores; saturating stores) + fp16 value maps --
    rounds 2-4 served it as bf16 activations.  This test pins the dtypes that result, that the fp16 library is the one
    that ran, and the whole transformer (encoder + proposals + six decoder layers at 900 queries) in that mode; the
    numerical bars are tests/test_encoder_timed_mode_gpu.py (against the reference's own fp16 autocast) and
    tests/test_decoder_gpu.py::test_fp16_mode_against_fp16_operand_arithmetic."""
    from salience_detr_amd.salience_transformer import build_salience_transformer
    image_sizes = [(800, 1333), (800, 1066)]
    tr = build_salience_transformer()
    tr.load_state_dict(syn.det_state_dict(tr.state_dict()))
    tr = tr.eval().cuda()
    tr.set_dtype(torch.float16)
    assert tr.encoder_dtype == torch.float16
    assert tr.decoder.layers[0].linear1.weight.dtype == torch.float16
    assert all(l.self_attn.value_dtype == torch.float16 for l in tr.encoder.layers)
    assert all(l.cross_attn.value_dtype == torch.float16 for l in tr.decoder.layers)
    assert tr.enc_mask_predictor.layer1[1].weight.dtype == torch.float32      # the filtering stage stays fp32
    _, masks = syn.make_masks(image_sizes)
    shapes = [tuple(m.shape[-2:]) for m in masks]
    feats = [f.cuda() for f in syn.make_feats(2, shapes, 256, 0)]
    masks = [m.cuda() for m in masks]
    pos = [syn.sine_position_embedding(m, 128) for m in masks]
    from salience_detr_amd import _hip, ms_deform_attn as M
    with torch.no_grad():
        out = tr(feats, masks, pos)
    assert _hip._lib_f16 is not None and M.last_forward_kernel(torch.float16) != 0    # the fp16-activation library ran the MSDA
    assert out[0].dtype == torch.float16
    assert out[0].shape == (6, 2, 900, 91) and out[1].shape == (6, 2, 900, 4)
    assert all(torch.isfinite(t.float()).all() for t in out[:4])
    assert (out[1] >= 0).all() and (out[1] <= 1).all()


def test_config5_step_is_a_graph_of_at_most_240_launches():
    """BASELINE configs[4] with the neck under hipGraph capture: 234 nodes since round 5 (313 before the decoder's GEMM
    chains, in-projections, attention tails and layer heads became row-tile launches: ten per decoder layer), none of
    them a memset, and the replay reproduces the eager outputs bit for bit.  A silent fall-back of one of those stages to
    its module-by-module form (a cache key, a dtype or a contiguity rule that no longer holds) adds 6-36 nodes."""
    from salience_detr_amd import graph_guard
    from salience_detr_amd.salience_transformer import build_salience_transformer
    image_sizes = [(800, 1333), (800, 1066)]
    tr = build_salience_transformer(with_neck=True)
    tr.load_state_dict(syn.det_state_dict(tr.state_dict()))
    tr = tr.eval().cuda()
    tr.set_dtype(torch.float16)
    tr.static_proposals = True
    img_mask, masks = syn.make_masks(image_sizes)
    canvas = tuple(img_mask.shape[-2:])
    shapes = [tuple(m.shape[-2:]) for m in masks]
    feats = [f.cuda() for f in syn.make_feats(2, shapes, 256, 0)]
    masks = [m.cuda() for m in masks]
    pos = [syn.sine_position_embedding(m, 128).cuda() for m in masks]

    def step():
        with torch.no_grad():
            return tr(feats, masks, pos, image_sizes=image_sizes, canvas=canvas)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            eager = [t.clone() for t in step()[:4]]
    torch.cuda.current_stream().wait_stream(side)
    g = graph_guard.new_graph()
    with torch.cuda.graph(g):
        out = step()
    types = graph_guard.node_types(g)
    if not types:
        pytest.skip("this torch build exposes no graph handle")
    assert graph_guard.memset_nodes(g) == 0
    assert len(types) <= 240, len(types)
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(out[:4], eager):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tag", ["plain", "neck"])
def test_whole_transformer_training_step_matches_reference_gradients(gold, tag):
    """``SalienceTransformer.forward`` under autograd with denoising queries (salience_transformer.py:195-233; train mode:
    the neck's norms take batch statistics): the five outputs, the loss of fixed random weights on them and the gradients
    of a spread of parameters -- filtering head, encoder layers, proposal heads, neck, decoder, embeddings -- and of the
    finest input level against what the imported reference produced (tests/golden/make_golden.py transformer_train)."""
    t = np.load(os.path.join(G, "transformer_train_small.npz"))
    neck_fixture = {"sd_keys": t["neck.sd_keys"], "sd_crc": t["neck.sd_crc"]} if tag == "neck" else None
    tr, sd = build_product_transformer(gold, neck_fixture)
    assert sorted(sd) == t[f"{tag}.sd_keys"].tolist()
    tr = tr.cuda().train()
    E, proposals = int(gold["hyper"][0]), int(gold["hyper"][8])
    feats, masks, pos = [[x.cuda() for x in part] for part in inputs(gold)]
    feats[0].requires_grad_(True)
    label_q, box_q, attn_mask = [x.cuda() for x in syn.denoising_inputs(feats[0].shape[0], E, proposals, int(t["dn"]))]
    out_cls, out_box, enc_cls, enc_box, sal = tr(feats, masks, pos, label_q, box_q, attn_mask)
    assert (enc_cls.cpu() - _t(t[f"{tag}.enc_outputs_class"])).abs().max() < 1e-3
    assert (enc_box.cpu() - _t(t[f"{tag}.enc_outputs_coord"])).abs().max() < 1e-4
    assert (out_cls.cpu() - _t(t[f"{tag}.outputs_classes"])).abs().max() < 2e-3
    assert (out_box.cpu() - _t(t[f"{tag}.outputs_coords"])).abs().max() < 2e-4
    outs = [out_cls, out_box, enc_cls, enc_box] + list(sal)
    ws = syn.train_loss_weights([o.shape for o in outs])
    loss = sum((o * w.cuda()).sum() for o, w in zip(outs, ws))
    loss.backward()
    want_loss = float(t[f"{tag}.loss"])
    assert abs(loss.item() - want_loss) < 2e-3 * max(1.0, abs(want_loss)), (loss.item(), want_loss)
    params = dict(tr.named_parameters(remove_duplicate=False))
    names = t["grad_names"].tolist() + (t["neck_grad_names"].tolist() if tag == "neck" else [])
    worst = {}
    for n in names:
        g = params[n].grad
        assert g is not None, n
        g = g.detach().cpu()
        if g.numel() > 4096 and g.dim() >= 2:
            g = g[::4, ::4]
        ref = _t(t[f"{tag}.grad.{n}"])
        assert g.shape == ref.shape, n
        worst[n] = ((g - ref).abs().max() / max(1.0, ref.abs().max().item())).item()
    gf = feats[0].grad.cpu()[:, ::8]
    ref = _t(t[f"{tag}.grad.feat0"])
    worst["feat0"] = ((gf - ref).abs().max() / max(1.0, ref.abs().max().item())).item()
    bad = {n: e for n, e in worst.items() if not e < 2e-3}
    assert not bad, bad
    if tag == "neck":   # the training-mode forward also moved the norms' running statistics, as nn.BatchNorm2d does
        bn = tr.neck.lateral_convs[0][1]
        assert (bn.running_mean.cpu() - _t(t["neck.running_mean.lateral0"])).abs().max() < 1e-4
        assert (bn.running_var.cpu() - _t(t["neck.running_var.lateral0"])).abs().max() < 1e-4


def test_whole_transformer_eval_mode_is_differentiable_too(gold):
    """eval() + grad mode (fine-tuning with frozen norm statistics): the neck takes its torch-op form with running
    statistics; the forward equals the no-grad kernels' and a gradient reaches the input."""
    neck = np.load(os.path.join(G, "transformer_small_neck.npz"))
    tr, _ = build_product_transformer(gold, neck)
    tr = tr.cuda().eval()
    feats, masks, pos = [[x.cuda() for x in part] for part in inputs(gold)]
    with torch.no_grad():
        want = tr(feats, masks, pos)
    feats[1].requires_grad_(True)
    got = tr(feats, masks, pos)
    assert (got[0] - want[0]).abs().max() < 2e-3 and (got[1] - want[1]).abs().max() < 2e-4
    (got[0].sum() + got[2].sum()).backward()
    assert feats[1].grad is not None and torch.isfinite(feats[1].grad).all() and feats[1].grad.abs().max() > 0

