"""Two-launch top-k self-attention (csrc/topk_attention.hip) against an fp32 restatement of
salience_transformer.py:366-379 -- gather select_tgt / select_pos, nn.MultiheadAttention(q = k = x + pos, v = x),
residual + pre_norm, scatter -- fed the SAME bf16-rounded operands and parameters.

Bar: the result is stored as bf16 (|y| <= ~4 after LayerNorm: half an ulp is up to 1.6e-2); operands of the three
matrix products are rounded to bf16 once each (q/k/v rows, softmax weights, concatenated heads).
max |err| <= 6e-2, mean |err| <= 6e-3; rows that were not selected must be bit-identical.
"""
import math

import pytest
import torch
from torch import nn

from salience_detr_amd import synthetic as syn
from salience_detr_amd.filter_ops import topk_self_attention_, topk_self_attention_applies

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _modules(seed):
    mha = nn.MultiheadAttention(256, 8, 0.0, batch_first=True)
    norm = nn.LayerNorm(256)
    sd = syn.det_state_dict({**{"mha." + k: v for k, v in mha.state_dict().items()},
                             **{"norm." + k: v for k, v in norm.state_dict().items()}}, salt=seed)
    mha.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("mha.")})
    norm.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("norm.")})
    return mha.to(DEV).to(torch.bfloat16), norm.to(DEV).to(torch.bfloat16)


def _reference(x, pos, sel, mha, norm):
    """fp32 arithmetic on the bf16-rounded tensors."""
    B, _, E = x.shape
    xf, pf = x.float(), pos.float()
    idx = sel.unsqueeze(-1).expand(-1, -1, E)
    tgt, tp = torch.gather(xf, 1, idx), torch.gather(pf[:, :x.shape[1]], 1, idx)
    w, b = mha.in_proj_weight.float(), mha.in_proj_bias.float()
    H, hd = 8, 32
    N = sel.shape[1]
    qk = tgt + tp
    q = torch.nn.functional.linear(qk, w[:E], b[:E]).view(B, N, H, hd).transpose(1, 2)
    k = torch.nn.functional.linear(qk, w[E:2 * E], b[E:2 * E]).view(B, N, H, hd).transpose(1, 2)
    v = torch.nn.functional.linear(tgt, w[2 * E:], b[2 * E:]).view(B, N, H, hd).transpose(1, 2)
    att = ((q / math.sqrt(hd)) @ k.transpose(-1, -2)).softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, E)
    o = torch.nn.functional.linear(o, mha.out_proj.weight.float(), mha.out_proj.bias.float())
    y = torch.nn.functional.layer_norm(tgt + o, (E,), norm.weight.float(), norm.bias.float(), norm.eps)
    return xf.scatter(1, idx, y)


@pytest.mark.parametrize("B,rows,N,n0", [(2, 11363, 300, 11363), (2, 2272, 300, 11363), (1, 500, 37, 700),
                                         (3, 640, 320, 640), (2, 40, 32, 40)])
def test_topk_self_attention_vs_fp32_restatement(B, rows, N, n0):
    mha, norm = _modules(seed=N)
    x = syn.det_randn("tk.x", (B, rows, 256), salt=rows).to(DEV).to(torch.bfloat16)
    pos = syn.det_randn("tk.pos", (B, n0, 256), salt=rows).to(DEV).to(torch.bfloat16)
    g = torch.Generator().manual_seed(N)
    sel = torch.stack([torch.randperm(rows, generator=g)[:N] for _ in range(B)]).to(DEV)
    assert topk_self_attention_applies(x, pos, mha, norm, N)
    expect = _reference(x, pos, sel, mha, norm)
    got = topk_self_attention_(x.clone(), pos, sel, mha, norm)
    err = (got.float() - expect).abs()
    picked = torch.zeros(B, rows, dtype=torch.bool, device=DEV).scatter_(1, sel, True)
    assert torch.equal(got[~picked], x[~picked])                     # untouched rows: bit-identical
    e = err[picked]
    assert e.max().item() <= 6e-2, e.max().item()
    assert e.mean().item() <= 6e-3, e.mean().item()


def test_topk_self_attention_rejects_other_shapes():
    mha, norm = _modules(seed=1)
    x = torch.zeros(1, 64, 256, dtype=torch.bfloat16, device=DEV)
    sel = torch.arange(8, device=DEV)[None]
    assert not topk_self_attention_applies(x.float(), x, mha, norm, 8)
    with pytest.raises(RuntimeError):
        topk_self_attention_(x.float(), x, sel, mha, norm)
    with pytest.raises(RuntimeError):
        topk_self_attention_(x, x, sel.int(), mha, norm)
