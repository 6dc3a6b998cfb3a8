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
