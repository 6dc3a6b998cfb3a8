"""GPU: the training step `bench.py` times (`train_step`), at the benchmark's full size, against autograd through the
oracle -- one 800x1333 image, E = 256, six layers, fp32, Linear products on the x3 kernels, forward + backward both eager
and as a replayed hipGraph (reference: `util/engine.py:44-64` around `SalienceTransformer.forward`,
`models/bricks/salience_transformer.py:97-183`).

Checks: the loss, and the gradients of a handful of parameters from every stage of the path (position
embedding, salience head, the coarse-to-fine `alpha`, deformable attention projections, feed-forward, the 300-row
attention, LayerNorm), within 1e-2 of each gradient's own scale (8e-2 for a layer whose top-300 set differs from the oracle's by a token); and that the replayed graph reproduces the eager step's
gradients (the round-4 finding: memset nodes are not replayed on this stack -- `CHANGELOG.md`)."""
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import synthetic as syn
from salience_detr_amd.hot_path import build_hot_path
from salience_detr_amd.linear_x3 import use_x3_linear_
from salience_detr_amd.salience_filtering import replay_safe_mean

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CHECKED = (
    "level_embeds",
    "alpha",
    "enc_mask_predictor.layer1.1.weight",
    "enc_mask_predictor.layer2.4.weight",
    "enc_output.weight",
    "encoder.layers.0.self_attn.sampling_offsets.weight",
    "encoder.layers.0.self_attn.attention_weights.bias",
    "encoder.layers.0.self_attn.value_proj.weight",
    "encoder.layers.2.self_attn.output_proj.weight",
    "encoder.layers.0.linear1.weight",
    "encoder.layers.0.linear1.bias",
    "encoder.layers.3.linear2.weight",
    "encoder.layers.5.linear1.weight",
    "encoder.layers.5.linear1.bias",
    "encoder.layers.5.linear2.weight",
    "encoder.layers.5.norm1.weight",
    "encoder.layers.5.norm2.weight",
    "encoder.layers.5.self_attn.output_proj.weight",
    "encoder.layers.5.self_attn.value_proj.weight",
    "encoder.layers.4.linear1.weight",
    "encoder.layers.2.linear1.weight",
    "encoder.layers.1.pre_attention.in_proj_weight",
    "encoder.layers.4.norm2.weight",
)


def _compare(got, ref, name):
    """(worst error / the gradient's scale, rows left out).  The ReLU's gate is discontinuous: a hidden unit whose
    pre-activation is within rounding of zero for some token (expected: ~1e-6 of the T x 2048 of them, a handful per layer)
    is on in one run and off in the other -- GPU against host, or two GPU runs whose split reductions add in another order
    -- and that token's whole contribution dh[t, j] * x[t] appears in / vanishes from row j of linear1's gradient: a few
    percent of a row that sums ~1000 active tokens.  Rows of linear1's gradients are therefore compared one by one and the
    few beyond 1e-2 counted instead of bounded."""
    scale = ref.abs().max().item()
    assert scale > 0.0, name
    err = (got - ref).abs() / scale
    off = 0
    if ".linear1." in name:
        per_row = err.reshape(err.shape[0], -1).max(1)[0]
        off = int((per_row > 1e-2).sum())
        err = per_row[per_row <= 1e-2] if off else per_row
    return err.max().item(), off


def _loss(memory, score_maps, w, mean):
    return mean(memory * w) * 100.0 + sum((s * s).mean() for s in score_maps)


def test_full_size_training_step_matches_oracle_autograd():
    m = build_hot_path(max_num_embedding=200)
    sd0 = syn.det_state_dict(m.state_dict())
    m.load_state_dict(sd0)
    sizes = [(800, 1333)]
    _, masks = syn.make_masks(sizes)
    canvas = syn.pad_to_32(*sizes[0])
    shapes = [tuple(x.shape[-2:]) for x in masks]
    feats = syn.make_feats(1, shapes, 256, seed=0)
    pos = [syn.sine_position_embedding(x, 128) for x in masks]

    # ---- oracle: the same loss through the differentiable closed form on the host
    torch.set_num_threads(max(8, torch.get_num_threads()))
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    out = R.hot_path(sd, feats, masks, pos, core=R.msda_core_torch)
    w = syn.det_randn("train_full.w", tuple(out["memory"].shape))
    rloss = _loss(out["memory"], out["score_maps"], w, lambda t: t.mean())
    rloss.backward()

    # ---- the product path: train mode, x3 Linear products, as bench.py's train_record sets it up
    m = m.to(DEV).train()
    use_x3_linear_(m)
    params = dict(m.named_parameters())
    missing = [n for n in CHECKED if n not in params or n not in sd]
    assert not missing, missing
    f = [t.to(DEV) for t in feats]
    k = [t.to(DEV) for t in masks]
    p = [t.to(DEV) for t in pos]
    wd = w.to(DEV)

    def forward_backward():
        m.zero_grad(set_to_none=True)
        memory, score_maps = m(f, k, p, image_sizes=sizes, canvas=canvas)
        loss = _loss(memory, score_maps, wd, replay_safe_mean)
        loss.backward()
        return loss.detach()

    # instrumentation for the comparison below: every layer's top-300 positions and the sorted index list they refer to
    from salience_detr_amd import salience_encoder as SE
    picked, lists = [], []
    real_topk = SE.masked_topk_desc

    def recording_topk(score, k, *a, **kw):
        r = real_topk(score, k, *a, **kw)
        if k == 300:
            picked.append(r[1].detach().cpu())
        return r

    hook = m.encoder.register_forward_pre_hook(lambda mod, a, kw: lists.append(kw["foreground_inds"][0].detach().cpu()),
                                               with_kwargs=True)
    SE.masked_topk_desc = recording_topk
    try:
        loss = forward_backward()
    finally:
        SE.masked_topk_desc = real_topk
        hook.remove()
    torch.cuda.synchronize()
    # top-300 sets of each layer, both sides as TOKEN ids (each side's positions through its own sorted list): a near-tie
    # of the class score may fall the other way (fp32 on both sides, different summation orders), and a layer with a flipped
    # token has ~1/300 of its 300-row attention -- and of what depends on it -- on another row
    assert len(picked) == 6 and len(lists) == 1
    flips = []
    for kk in range(6):
        a = set(out["foreground_inds"][0][0][out["layer_sel"][kk][0]].tolist())
        g_ = set(lists[0][0][picked[kk][0]].tolist())
        flips.append(len(a ^ g_))
    print("top-300 selection: tokens in exactly one of (GPU, oracle) sets per layer:", flips)
    assert sum(flips) <= 12, flips
    assert abs(loss.item() - rloss.item()) < 2e-3 * max(1.0, abs(rloss.item())), (loss.item(), rloss.item())
    eager = {n: params[n].grad.detach().clone() for n in CHECKED}
    worst, off_rows = {}, {}
    for n in CHECKED:
        worst[n], off_rows[n] = _compare(eager[n].cpu(), sd[n].grad, n)
    print("full-size training step: worst gradient error / scale per parameter:", {n: round(v, 6) for n, v in worst.items()})
    # (fp32 on both sides, but sums over up to 22 323 tokens in different orders, exact-split matrix-core products against
    # the host's fp32 GEMMs, and fixed-point accumulation in the MSDA backward: measured 1e-6 .. 6e-3 of a gradient's scale;
    # the parameters of a layer whose top-300 set differs by a token: a few percent)
    print("rows of linear1 gradients off by more than 1e-2 of the gradient's scale (ReLU gate flips):", off_rows)
    for n, v in worst.items():
        layer = int(n.split(".")[2]) if n.startswith("encoder.layers.") else None
        bar = 8e-2 if layer is not None and flips[layer] else 1e-2
        assert v < bar, (n, v, flips)
        assert off_rows.get(n, 0) <= 20, (n, off_rows)   # of 2048 rows

    # ---- the same step as a replayed hipGraph (the form bench.py times): gradients land in the captured tensors
    forward_backward()   # (a second eager step: the pool's allocations settle before the capture)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        forward_backward()
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_static = forward_backward()
    captured = {n: params[n].grad for n in CHECKED}
    for _ in range(3):   # replays on freshly poisoned gradient buffers: whatever the graph leaves there is its own work
        for t in captured.values():
            t.fill_(float("nan"))
        g.replay()
    torch.cuda.synchronize()
    assert abs(loss_static.item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    for n in CHECKED:
        # Two runs of the same step are not bit-identical: fp32 atomics (MSDA backward flush, split reductions of the Linear
        # products) add in a different order, a pre-activation within that noise of zero flips its ReLU gate, and the ONE
        # token whose dh[t, j] appears or vanishes moves every gradient upstream of it by up to ~1/T of its scale times that
        # token's weight -- measured 3e-4 .. 2.1e-3 between runs (T = 2272 rows in the last layer).  What this check is
        # for -- a replay that drops work (CHANGELOG round 4: memset nodes) -- shows as NaNs or order-one errors.
        v, off = _compare(captured[n], eager[n], n)
        assert v <= (1e-2 if ".linear1." in n else 5e-3) and off <= 20, (n, v, off)
