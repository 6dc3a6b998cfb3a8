"""The top-k attention launch that also carries the MSDA offset | weight projection (csrc/fused_head_value.hip:
fused_attn_proj_kernel + topk_proj_scatter_kernel) against the two operators one after the other."""
import pytest
import torch

from salience_detr_amd import filter_ops as F
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("rows,N", [(2272, 300), (11363, 300), (700, 289), (3000, 320)])
def test_attention_carrying_the_projection_equals_the_two_operators(rows, N):
    B = 2
    torch.manual_seed(rows)
    mha = torch.nn.MultiheadAttention(256, 8, batch_first=True).to(DEV).to(torch.bfloat16)
    norm = torch.nn.LayerNorm(256).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * syn.det_randn("tpg", (256,)).to(DEV))
        norm.bias.copy_(0.1 * syn.det_randn("tpb", (256,)).to(DEV))
    q0 = (syn.det_randn(f"tpq{rows}", (B, rows, 256)) * 0.8).to(DEV).to(torch.bfloat16)
    pos = (syn.det_randn(f"tpp{rows}", (B, rows + 50, 256)) * 0.5).to(DEV).to(torch.bfloat16)
    sel = torch.stack([torch.randperm(rows)[:N] for _ in range(B)]).to(DEV)
    w = (syn.det_randn("tpw", (384, 256)) * 0.06).to(DEV).to(torch.bfloat16)
    b = (syn.det_randn("tpbb", (384,)) * 0.2).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        qa, qb = q0.clone(), q0.clone()
        F.topk_self_attention_(qa, pos, sel, mha, norm)
        want = F.token_linear(qa, w, b, x_add=pos[:, :rows], group_features=48)
        got = F.topk_self_attention_(qb, pos, sel, mha, norm, projection=(w, b))
    assert got is not None and got.shape == (B, 8, rows, 48)
    assert torch.equal(qa, qb)                              # the attention itself: same bits
    untouched = torch.ones(B, rows, dtype=torch.bool, device=DEV)
    untouched[torch.arange(B, device=DEV)[:, None], sel] = False
    # rows the attention did not update come from the same kernel body: same bits
    assert torch.equal(got.permute(0, 2, 1, 3)[untouched], want.permute(0, 2, 1, 3)[untouched])
    # the 300 updated rows are projected inside the attention kernel (16x16x32 MFMA, another summation order): bf16 ulps
    d = (got.float() - want.float()).abs()
    scale = want.float().abs().max().item()
    assert d.max().item() <= 2 ** -7 * scale
    assert (d > 0).float().mean().item() < 0.05


@pytest.mark.parametrize("rows,N,ties", [(11363, 300, False), (9090, 300, True), (2272, 300, False), (1500, 289, False),
                                        (4545, 320, True),
                                        # rows beyond one workgroup's sort (the 5scale pyramid's layers): slices first
                                        (45330, 300, False), (36264, 300, True), (27198, 300, False), (18132, 300, True),
                                        (17409, 320, True)])
def test_selection_launch_with_the_in_projection_equals_the_two_launches(rows, N, ties):
    """csrc/topk.hip topk_hsort_inproj_kernel: the layer's top-k selection and the in-projection of the selected rows in ONE
    launch (every workgroup of an image repeats the selection) -- the same selection as masked_topk_desc (ties by position
    included) and, through the attention launch that follows, the same bits in the queries and the projection slab."""
    B = 2
    torch.manual_seed(rows + N)
    mha = torch.nn.MultiheadAttention(256, 8, batch_first=True).to(DEV).to(torch.bfloat16)
    norm = torch.nn.LayerNorm(256).to(DEV).to(torch.bfloat16)
    q0 = (syn.det_randn(f"siq{rows}", (B, rows, 256)) * 0.8).to(DEV).to(torch.bfloat16)
    pos = (syn.det_randn(f"sip{rows}", (B, rows + 50, 256)) * 0.5).to(DEV).to(torch.bfloat16)
    score = syn.det_randn(f"sis{rows}", (B, rows)).to(DEV)
    if ties:
        score = (score * 8).round() / 8          # many exact ties, also across the cut
        score[1, : rows // 2] = score[1, 0]      # a crowded bin
    w = (syn.det_randn("siw", (384, 256)) * 0.06).to(DEV).to(torch.bfloat16)
    b = (syn.det_randn("sibb", (384,)) * 0.2).to(DEV).to(torch.bfloat16)
    assert F.topk_select_inproj_applies(score, N, q0, pos, mha, norm)
    with torch.no_grad():
        qa, qb = q0.clone(), q0.clone()
        sel = F.masked_topk_desc(score, N, want_scores=False)[1]
        want = F.topk_self_attention_(qa, pos, sel, mha, norm, projection=(w, b))
        both = F.topk_select_inproj(score, N, qb, pos, mha)
        got = F.topk_self_attention_(qb, pos, both.selected, mha, norm, projection=(w, b), inprojection=both)
    assert torch.equal(both.selected, sel)
    assert torch.equal(qa, qb) and torch.equal(got, want)
    # the selection against the framework's (descending scores; equal scores by position)
    ref = torch.sort(score, dim=1, descending=True, stable=True)[1][:, :N]
    assert torch.equal(sel, ref)


def test_selection_launch_with_the_in_projection_carries_the_row_orders():
    """... and the encoder's row-order job, as the selection's own launch does at layer 0."""
    B, rows, N = 2, 11363, 300
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21)]
    S = sum(h * w for h, w in shapes)
    torch.manual_seed(5)
    mha = torch.nn.MultiheadAttention(256, 8, batch_first=True).to(DEV).to(torch.bfloat16)
    norm = torch.nn.LayerNorm(256).to(DEV).to(torch.bfloat16)
    q0 = (syn.det_randn("soq", (B, rows, 256)) * 0.8).to(DEV).to(torch.bfloat16)
    pos = (syn.det_randn("sop", (B, rows, 256)) * 0.5).to(DEV).to(torch.bfloat16)
    score = syn.det_randn("sos", (B, rows)).to(DEV)
    sorted_index = torch.stack([torch.randperm(S)[:rows] for _ in range(B)]).to(DEV)
    counts = [rows, 9090, 6817, 6817, 4545, 2272]
    want_orders = F.layer_row_orders(sorted_index, counts, shapes)
    job = F.layer_row_orders(sorted_index, counts, shapes, as_job=True)
    with torch.no_grad():
        both = F.topk_select_inproj(score, N, q0, pos, mha, orders_job=job)
    assert job.done
    for a, c in zip(job.orders, want_orders):
        assert torch.equal(a, c)
    assert torch.equal(both.selected, F.masked_topk_desc(score, N, want_scores=False)[1])


def test_selection_launch_random_row_lengths_against_a_stable_sort():
    """Row lengths across both forms (one launch up to 17 408 keys, slices beyond), every admissible k, batch 1-3: wherever
    the gate accepts the shape the selection equals torch's stable descending sort (ties by position); what it declines is
    exactly what the library's own query declines."""
    g = torch.Generator().manual_seed(20260)
    mha = torch.nn.MultiheadAttention(256, 8, batch_first=True).to(DEV).to(torch.bfloat16)
    norm = torch.nn.LayerNorm(256).to(DEV).to(torch.bfloat16)
    took = declined = 0
    lengths = [1024, 1500, 17408, 17409, 17500, 20000, 24577, 32768, 40001, 65536, 90000] + \
        [int(v) for v in torch.randint(1024, 70000, (14,), generator=g)]
    for n in lengths:
        B = int(torch.randint(1, 4, (1,), generator=g))
        k = int(torch.randint(289, 321, (1,), generator=g))
        score = torch.randn(B, n, generator=g)
        score = ((score * 64).round() / 64).to(DEV)              # plenty of exact ties
        q = torch.zeros(B, n, 256, dtype=torch.bfloat16, device=DEV)
        if not F.topk_select_inproj_applies(score, k, q, q, mha, norm):
            declined += 1
            assert n < 1024 or 5 * k > 2 * n or (n > 17408 and F._hip.lib().sdetr_topk_select_candidate_bytes(B, n, k) == 0)
            continue
        took += 1
        with torch.no_grad():
            both = F.topk_select_inproj(score, k, q, q, mha)
        ref = torch.sort(score, dim=1, descending=True, stable=True)[1][:, :k]
        assert torch.equal(both.selected, ref), (n, k, B)
    assert took >= 20
