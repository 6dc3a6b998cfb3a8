"""The training step's dense attention core (csrc/attention_train.hip) against float64 autograd of the textbook form."""
import math

import pytest
import torch

from salience_detr_amd import attention_train as A
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _err(got, want64):
    return ((got.double().cpu() - want64).abs().max() / max(1e-30, want64.abs().max().item())).item()


def _textbook(qk, v, H):
    B, N, E = v.shape
    hd = E // H
    q = qk[..., :E].view(B, N, H, hd).transpose(1, 2)
    k = qk[..., E:].view(B, N, H, hd).transpose(1, 2)
    vv = v.view(B, N, H, hd).transpose(1, 2)
    att = (q @ k.transpose(-1, -2) / math.sqrt(hd)).softmax(-1)
    return (att @ vv).transpose(1, 2).reshape(B, N, E)


@pytest.mark.parametrize("B,N,H", [(2, 300, 8), (1, 1, 8), (3, 5, 2), (2, 64, 8), (1, 65, 4), (2, 512, 8), (1, 3, 1)])
def test_attention_core_forward_backward_match_float64(B, N, H):
    E = 32 * H
    qk = syn.det_randn(f"at.qk{N}{H}", (B, N, 2 * E)) * 1.5
    v = syn.det_randn(f"at.v{N}{H}", (B, N, E))
    go = syn.det_randn(f"at.g{N}{H}", (B, N, E))

    def run(dtype, device, native):
        a = qk.to(device=device, dtype=dtype).requires_grad_(True)
        b = v.to(device=device, dtype=dtype).requires_grad_(True)
        o = A.attention_qk_v(a, b, H) if native else _textbook(a, b, H)
        o.backward(go.to(device=device, dtype=dtype))
        return o.detach(), a.grad, b.grad

    want = run(torch.float64, "cpu", False)
    ref = run(torch.float32, DEV, False)
    assert A.applies(qk.to(DEV), v.to(DEV), H)
    got = run(torch.float32, DEV, True)
    for g, f, w in zip(got, ref, want):
        assert _err(g, w) <= max(3.0 * _err(f, w), 3e-6)


def test_attention_core_extreme_logits_and_support():
    """Logits of +-80 (a saturated softmax) stay finite; shapes the kernel does not take are refused by `applies`."""
    B, N, H = 1, 40, 8
    E = 32 * H
    qk = syn.det_randn("at.big", (B, N, 2 * E)).to(DEV) * 9
    v = syn.det_randn("at.bigv", (B, N, E)).to(DEV)
    a, b = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    o = A.attention_qk_v(a, b, H)
    o.sum().backward()
    assert torch.isfinite(o).all() and torch.isfinite(a.grad).all() and torch.isfinite(b.grad).all()
    want = _textbook(qk.double().cpu(), v.double().cpu(), H)
    assert _err(o, want) < 1e-5
    assert not A.applies(qk.bfloat16(), v.bfloat16(), H)
    assert not A.applies(torch.zeros(1, 513, 512, device=DEV), torch.zeros(1, 513, 256, device=DEV), 8)
    assert not A.applies(torch.zeros(1, 10, 128, device=DEV), torch.zeros(1, 10, 64, device=DEV), 4)   # 16-channel heads
    assert not A.applies(torch.zeros(1, 10, 512), torch.zeros(1, 10, 256), 8)                            # CPU tensors


def test_encoder_layer_pre_attention_equals_multihead_attention_module():
    """The layer's autograd path (two in-projection GEMMs + the native core + out-projection) against
    nn.MultiheadAttention with the same parameters: output and parameter gradients."""
    from salience_detr_amd.salience_encoder import SalienceTransformerEncoderLayer
    layer = SalienceTransformerEncoderLayer(embed_dim=256, d_ffn=64, dropout=0.0, n_heads=8, topk_sa=300).to(DEV)
    qk = syn.det_randn("at.layer.qk", (2, 300, 256)).to(DEV).requires_grad_(True)
    v = syn.det_randn("at.layer.v", (2, 300, 256)).to(DEV).requires_grad_(True)
    w = syn.det_randn("at.layer.w", (2, 300, 256)).to(DEV)
    out = layer._pre_attention(qk, v)
    (out * w).sum().backward()
    got = (out.detach(), qk.grad.clone(), v.grad.clone(), layer.pre_attention.in_proj_weight.grad.clone(),
           layer.pre_attention.in_proj_bias.grad.clone(), layer.pre_attention.out_proj.weight.grad.clone())
    layer.zero_grad()
    qk2, v2 = qk.detach().clone().requires_grad_(True), v.detach().clone().requires_grad_(True)
    ref = layer.pre_attention(qk2, qk2, v2, need_weights=False)[0]
    (ref * w).sum().backward()
    want = (ref.detach(), qk2.grad, v2.grad, layer.pre_attention.in_proj_weight.grad, layer.pre_attention.in_proj_bias.grad,
            layer.pre_attention.out_proj.weight.grad)
    for g, r in zip(got, want):
        assert (g - r).abs().max() <= 2e-5 * max(1.0, r.abs().max().item())
