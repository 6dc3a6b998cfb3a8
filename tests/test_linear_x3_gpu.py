"""fp32 GEMM / Linear on the bf16 matrix cores at fp32 accuracy (csrc/gemm_x3.hip) against float64 products: the
error bar is the one torch's own fp32 matmul meets on the same operands."""
import pytest
import torch

from salience_detr_amd import linear_x3 as X
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _err(got, want64):
    return ((got.double().cpu() - want64).abs().max() / want64.abs().max()).item()


@pytest.fixture(params=["128x128 tiles", "256x128 tiles"])
def generation(request):
    """Every GEMM test runs on both kernel generations (the library picks by shape; ``pinned_generation`` pins one)."""
    with X.pinned_generation(1 if request.param.startswith("128") else 2):
        yield request.param


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 260, 256), (22726, 256, 2048), (4545, 384, 256), (4, 4, 4),
                                   (1000, 2048, 256), (516, 132, 36), (256, 128, 96), (260, 4, 100)])
@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, False), (False, True)])
def test_gemm_x3_matches_float64(M, N, K, ak, bk, generation):
    a = syn.det_randn(f"ga{M}{K}", (M, K)) * 1.3
    b = syn.det_randn(f"gb{N}{K}", (N, K)) * 0.7
    want = a.double() @ b.double().t()
    ad = (a if ak else a.t().contiguous()).to(DEV)
    bd = (b if bk else b.t().contiguous()).to(DEV)
    if (not ak and M % 4) or (not bk and N % 4):
        pytest.skip("row counts of a row-major-reduction operand must be multiples of 4")
    got = X.gemm_x3(ad, ak, bd, bk, M, N, K)
    ref32 = (a.to(DEV) @ b.to(DEV).t())
    assert _err(got, want) <= max(2.0 * _err(ref32, want), 2e-6)
    if K >= 256:   # split reduction (atomics into a zeroed C)
        got2 = X.gemm_x3(ad, ak, bd, bk, M, N, K, reduction_splits=4)
        assert _err(got2, want) <= max(2.0 * _err(ref32, want), 2e-6)


def test_gemm_x3_row_sums_of_a_reduction_major_operand(generation):
    """dw = dy^T x with db = sum_t dy from the same launch (with and without a split reduction)."""
    T, N, K = 3001 * 4, 260, 256
    dy, x = syn.det_randn("rs_dy", (T, N)).to(DEV), syn.det_randn("rs_x", (T, K)).to(DEV)
    want = dy.double().sum(0).cpu()
    want_dw = dy.double().cpu().t() @ x.double().cpu()
    bar = max(2.0 * _err(dy.t() @ x, want_dw), 3e-6)      # what torch's own fp32 product loses over 12 004 terms
    for splits in (1, 7):
        db = torch.zeros(N, device=DEV)
        dw = X.gemm_x3(dy, False, x, False, N, K, T, reduction_splits=splits, a_row_sum=db)
        assert _err(db, want) < 3e-6
        assert _err(dw, want_dw) <= bar
    with pytest.raises(RuntimeError):
        X.gemm_x3(x, True, x, True, T, T, K, a_row_sum=torch.zeros(T, device=DEV))


@pytest.mark.parametrize("M,N,K", [(300, 260, 256), (22726, 256, 2048), (4545, 384, 256), (130, 8, 8)])
@pytest.mark.parametrize("ak", [True, False])
@pytest.mark.parametrize("transposed", [False, True])
def test_gemm_x3_with_a_presplit_operand(M, N, K, ak, transposed, generation):
    """B given as three bf16 planes made once per call (from the matrix or from its transpose)."""
    if not ak and M % 4:
        pytest.skip("row counts of a reduction-major operand must be multiples of 4")
    a = syn.det_randn(f"pa{M}{K}", (M, K)) * 1.3
    b = syn.det_randn(f"pb{N}{K}", (N, K)) * 0.7
    want = a.double() @ b.double().t()
    ad = (a if ak else a.t().contiguous()).to(DEV)
    planes = X.presplit(b.t().contiguous().to(DEV), transpose=True) if transposed else X.presplit(b.to(DEV))
    assert planes.shape == (3, N, K)
    assert torch.equal(planes.float().sum(0).cpu(), b)          # the split is exact
    bias = syn.det_randn(f"pbias{N}", (N,)).to(DEV)
    got = X.gemm_x3_presplit_b(ad, ak, planes, M, N, K, bias=bias)
    ref32 = a.to(DEV) @ b.to(DEV).t() + bias
    want = want + bias.double().cpu()
    assert _err(got, want) <= max(2.0 * _err(ref32, want), 2e-6)


def test_gemm_x3_bias_and_argument_checks(generation):
    a, b = syn.det_randn("gba", (70, 64)).to(DEV), syn.det_randn("gbb", (36, 64)).to(DEV)
    bias = syn.det_randn("gbias", (36,)).to(DEV)
    got = X.gemm_x3(a, True, b, True, 70, 36, 64, bias=bias)
    want = a.double().cpu() @ b.double().cpu().t() + bias.double().cpu()
    assert _err(got, want) < 2e-6
    with pytest.raises(RuntimeError):
        X.gemm_x3(a.cpu(), True, b, True, 70, 36, 64)
    with pytest.raises(RuntimeError):
        X.gemm_x3(a, True, b, True, 70, 36, 60)
    with pytest.raises(RuntimeError):   # K-major operand with K % 4 != 0
        X.gemm_x3(a[:, :62].contiguous(), True, b[:, :62].contiguous(), True, 70, 36, 62)


@pytest.mark.parametrize("shape,N", [((2, 1137, 256), 2048), ((3000, 2048), 256), ((2, 300, 256), 384)])
def test_x3_linear_forward_backward_match_float64(shape, N, monkeypatch, generation):
    for flag in ("X3_FORWARD", "X3_DX", "X3_DW"):   # all three products through the kernel under test
        monkeypatch.setattr(X, flag, True)
    monkeypatch.setattr(X, "X3_WIDE_OUT_ROWS", 1)
    monkeypatch.setattr(X, "X3_WIDE_FEATURES", 1)
    K = shape[-1]
    lin = torch.nn.Linear(K, N)
    x = syn.det_randn(f"lx{N}", shape)
    gy = syn.det_randn(f"lg{N}", shape[:-1] + (N,))
    x64 = x.double().requires_grad_(True)
    l64 = torch.nn.Linear(K, N).double()
    l64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    l64(x64).backward(gy.double())
    xd = x.to(DEV).requires_grad_(True)
    ld = torch.nn.Linear(K, N).to(DEV)
    ld.load_state_dict(lin.state_dict())
    assert X.use_x3_linear_(ld) == 1 and isinstance(ld, X.X3Linear)
    y = ld(xd)
    y.backward(gy.to(DEV))
    xr = x.to(DEV).requires_grad_(True)
    lr = torch.nn.Linear(K, N).to(DEV)
    lr.load_state_dict(lin.state_dict())
    lr(xr).backward(gy.to(DEV))
    for got, ref, want in ((y, lr(xr), l64(x64)), (xd.grad, xr.grad, x64.grad), (ld.weight.grad, lr.weight.grad, l64.weight.grad),
                           (ld.bias.grad, lr.bias.grad, l64.bias.grad)):
        assert _err(got.detach(), want.detach()) <= max(3.0 * _err(ref.detach(), want.detach()), 3e-6)


def test_x3_linear_falls_back_where_the_kernel_does_not_apply():
    lin = X.X3Linear(256, 91).to(DEV)       # 91 classes: out features not a multiple of 4
    x = syn.det_randn("lfb", (5, 256)).to(DEV)
    assert torch.equal(lin(x), torch.nn.functional.linear(x, lin.weight, lin.bias))
    cpu = X.X3Linear(8, 8)
    assert torch.equal(cpu(torch.ones(2, 8)), torch.nn.functional.linear(torch.ones(2, 8), cpu.weight, cpu.bias))


@pytest.mark.parametrize("M,N,K", [(300, 260, 256), (4545, 2048, 256), (516, 132, 36)])
def test_gemm_x3_relu_and_gate_epilogues(M, N, K, generation):
    """C = relu(acc + bias) and C = gate <= 0 ? 0 : acc (the feed-forward's ReLU inside the products around it):
    bit-identical to the plain product followed by torch's elementwise pass; NaNs as torch's."""
    a = syn.det_randn(f"ea{M}{K}", (M, K)).to(DEV)
    b = syn.det_randn(f"eb{N}{K}", (N, K)).to(DEV)
    bias = syn.det_randn(f"ebias{N}", (N,)).to(DEV)
    plain = X.gemm_x3(a, True, b, True, M, N, K, bias=bias)
    assert torch.equal(X.gemm_x3(a, True, b, True, M, N, K, bias=bias, epilogue=X.EPI_RELU), plain.clamp_min(0.0))
    gate = syn.det_randn(f"eg{M}{N}", (M, N)).to(DEV)
    gate[1, 2] = float("nan")
    gate[3, 1] = 0.0
    gated = X.gemm_x3(a, True, b, True, M, N, K, bias=bias, epilogue=X.EPI_GATE, gate=gate)
    assert torch.equal(gated, torch.ops.aten.threshold_backward(plain, gate, 0.0))
    an = a.clone()
    an[0, 0] = float("nan")
    r = X.gemm_x3(an, True, b, True, M, N, K, epilogue=X.EPI_RELU)
    assert torch.isnan(r[0]).all() and not torch.isnan(r[1:]).any()
    with pytest.raises(RuntimeError):   # an atomic partial sum cannot be clamped
        X.gemm_x3(a, True, b, True, M, N, K, reduction_splits=2, epilogue=X.EPI_RELU,
                  out=torch.zeros(M, N, device=DEV))
    with pytest.raises(RuntimeError):
        X.gemm_x3(a, True, b, True, M, N, K, epilogue=X.EPI_GATE)


@pytest.mark.parametrize("rows,wide", [((2, 1137), True), ((2, 1137), False), ((2, 11363), None)])
def test_x3_ffn_matches_the_unfused_modules(rows, wide, monkeypatch):
    """linear2(relu(linear1(x))) as one autograd node (ReLU in the products' epilogues) against the same layers as
    separate X3Linear modules + nn.ReLU, and both against float64.  wide = True / False forces every product through
    the x3 kernel / the library's GEMM; None = the shipped shape routing at the benchmark's layer-0 size."""
    if wide is not None:
        monkeypatch.setattr(X, "X3_WIDE_OUT_ROWS", 1 if wide else 10 ** 9)
        monkeypatch.setattr(X, "X3_LONG_REDUCTION_ROWS", 1 if wide else 10 ** 9)
    l1, l2 = torch.nn.Linear(256, 2048), torch.nn.Linear(2048, 256)
    x = syn.det_randn("ffn_x", rows + (256,))
    gy = syn.det_randn("ffn_gy", rows + (256,))
    x64 = x.double().requires_grad_(True)
    a64, b64 = torch.nn.Linear(256, 2048).double(), torch.nn.Linear(2048, 256).double()
    a64.load_state_dict({k: v.double() for k, v in l1.state_dict().items()})
    b64.load_state_dict({k: v.double() for k, v in l2.state_dict().items()})
    y64 = b64(torch.relu(a64(x64)))
    y64.backward(gy.double())

    def run(fused):
        m1, m2 = torch.nn.Linear(256, 2048).to(DEV), torch.nn.Linear(2048, 256).to(DEV)
        m1.load_state_dict(l1.state_dict())
        m2.load_state_dict(l2.state_dict())
        seq = torch.nn.Sequential(m1, m2)
        assert X.use_x3_linear_(seq) == 2
        xd = x.to(DEV).requires_grad_(True)
        if fused:
            assert X.x3_ffn_applies(xd, m1, m2)
            y = X.x3_ffn(xd, m1, m2)
        else:
            y = m2(torch.relu(m1(xd)))
        y.backward(gy.to(DEV))
        return y.detach(), xd.grad, m1.weight.grad, m1.bias.grad, m2.weight.grad, m2.bias.grad

    fused, plain = run(True), run(False)
    want = (y64.detach(), x64.grad, a64.weight.grad, a64.bias.grad, b64.weight.grad, b64.bias.grad)
    for got, ref, w in zip(fused, plain, want):
        assert _err(got, w) <= max(3.0 * _err(ref, w), 3e-6)
