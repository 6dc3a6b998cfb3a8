"""``norm(x + residual)`` of the training step (csrc/layer_norm_train.hip) against float64 autograd: the error bar is what
torch's own fp32 LayerNorm meets on the same operands."""
import pytest
import torch

from salience_detr_amd import layer_norm_train as L
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _err(got, want64):
    return ((got.double().cpu() - want64).abs().max() / max(1e-30, want64.abs().max().item())).item()


@pytest.mark.parametrize("shape", [(1, 1, 256), (2, 300, 256), (2, 11363, 256), (3, 1001, 64), (5, 77, 128), (2, 130, 512)])
@pytest.mark.parametrize("with_residual", [True, False])
def test_add_layer_norm_forward_backward_match_float64(shape, with_residual):
    C = shape[-1]
    x = syn.det_randn(f"ln.x{shape}", shape) * 2 + 0.5
    r = syn.det_randn(f"ln.r{shape}", shape) if with_residual else None
    gy = syn.det_randn(f"ln.g{shape}", shape)
    norm = torch.nn.LayerNorm(C)
    with torch.no_grad():
        norm.weight.copy_(syn.det_randn(f"ln.w{C}", (C,)) * 0.3 + 1)
        norm.bias.copy_(syn.det_randn(f"ln.b{C}", (C,)))

    def run(dtype, device, fused):
        n = torch.nn.LayerNorm(C).to(device=device, dtype=dtype)
        n.load_state_dict({k: v.to(dtype) for k, v in norm.state_dict().items()})
        xx = x.to(device=device, dtype=dtype).requires_grad_(True)
        rr = None if r is None else r.to(device=device, dtype=dtype).requires_grad_(True)
        y = L.add_layer_norm(xx, n, rr) if fused else n(xx if rr is None else xx + rr)
        y.backward(gy.to(device=device, dtype=dtype))
        return y.detach(), xx.grad, None if rr is None else rr.grad, n.weight.grad, n.bias.grad

    want = run(torch.float64, "cpu", False)
    ref = run(torch.float32, DEV, False)
    assert L.applies(x.to(DEV), norm.to(DEV), None if r is None else r.to(DEV))
    got = run(torch.float32, DEV, True)
    for g, f, w in zip(got, ref, want):
        if w is None:
            assert g is None
            continue
        assert _err(g, w) <= max(3.0 * _err(f, w), 2e-6)


def test_add_layer_norm_falls_back_where_the_kernel_does_not_apply():
    norm = torch.nn.LayerNorm(96).to(DEV)       # 96 channels: not a supported width
    x = syn.det_randn("ln.fb", (4, 96)).to(DEV)
    assert not L.applies(x, norm)
    assert torch.equal(L.add_layer_norm(x, norm, x), norm(x + x))
    cpu = torch.nn.LayerNorm(256)
    assert torch.equal(L.add_layer_norm(torch.ones(2, 256), cpu), cpu(torch.ones(2, 256)))
    bf = torch.nn.LayerNorm(256).to(DEV)
    assert not L.applies(x.new_zeros(2, 256).bfloat16(), bf)


def test_strided_inputs_and_gradient_accumulation_into_shared_parameters():
    """Views (a prefix slice of a wider buffer) as inputs, and one LayerNorm applied twice: the parameter gradients add up."""
    norm = torch.nn.LayerNorm(256).to(DEV)
    base = syn.det_randn("ln.base", (2, 500, 256)).to(DEV).requires_grad_(True)
    ref = base.detach().clone().requires_grad_(True)
    x, xr = base[:, :321], ref[:, :321]
    y = L.add_layer_norm(L.add_layer_norm(x, norm, x * 0.5), norm)
    y.square().sum().backward()
    g = (norm.weight.grad.clone(), norm.bias.grad.clone(), base.grad.clone())
    norm.zero_grad()
    yr = norm(norm(xr + xr * 0.5))
    yr.square().sum().backward()
    assert (y - yr).abs().max() < 1e-5
    for a, b in zip(g, (norm.weight.grad, norm.bias.grad, ref.grad)):
        assert (a - b).abs().max() <= 1e-4 * max(1.0, b.abs().max().item())
