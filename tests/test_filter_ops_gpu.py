"""GPU: bit-exact parity of the masked top-k / sort and row movers against the oracle's rule
(stable descending sort = ties to the lower index), incl. ties, masks, ragged and maximum sizes."""
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import filter_ops as F
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_topk(score, k, mask=None):
    if mask is not None:
        score = score.masked_fill(mask, score.min())
    return R.topk_desc_stable(score, k)


@pytest.mark.parametrize("B,N,k", [(1, 1, 1), (2, 7, 3), (2, 273, 273), (2, 1050, 1050), (2, 4200, 3360),
                                   (2, 16800, 6680), (2, 11363, 300), (2, 11363, 11363), (3, 5000, 1),
                                   (2, 22323, 3600), (1, 16384, 16384), (1, 40000, 20000), (1, 67200, 26880)])
def test_topk_exact(B, N, k):
    score = syn.det_randn(f"score{N}", (B, N))
    v, i = F.masked_topk_desc(score.to(DEV), k)
    rv, ri = R.topk_desc_stable(score, k)
    assert torch.equal(i.cpu(), ri)
    assert torch.equal(v.cpu(), rv)


@pytest.mark.parametrize("B,N,k", [(2, 273, 273), (2, 4200, 3360), (2, 16800, 6680)])
def test_masked_topk_with_ties(B, N, k):
    score = syn.det_randn("ms", (B, N))
    score[:, ::5] = score[:, 1::5][:, : score[:, ::5].shape[1]]  # plant exact duplicates
    mask = torch.zeros(B, N, dtype=torch.bool)
    mask[1, N // 2:] = True  # second image heavily padded: fill value ties everywhere
    mask[0, ::97] = True
    v, i = F.masked_topk_desc(score.to(DEV), k, mask=mask.to(DEV), fill_with_global_min=True, index_offset=1000)
    rv, ri = _oracle_topk(score, k, mask)
    assert torch.equal(i.cpu(), ri + 1000)
    assert torch.equal(v.cpu(), rv)


def test_sort_with_payload_and_special_values():
    B, N = 2, 11363
    score = syn.det_randn("sp", (B, N))
    score[0, :4] = torch.tensor([0.0, -0.0, float("inf"), -float("inf")])
    payload = torch.stack([torch.randperm(50000, generator=torch.Generator().manual_seed(b))[:N] for b in range(B)])
    _, got = F.masked_topk_desc(score.to(DEV), N, payload=payload.to(DEV), want_scores=False)
    order = torch.sort(score, dim=1, descending=True, stable=True)[1]
    assert torch.equal(got.cpu(), payload.gather(1, order))


def test_topk_errors():
    s = torch.zeros(2, 10, device=DEV)
    with pytest.raises(RuntimeError):
        F.masked_topk_desc(s, 11)
    with pytest.raises(RuntimeError):
        F.masked_topk_desc(s.cpu(), 3)
    with pytest.raises(RuntimeError):
        F.masked_topk_desc(s, 3, mask=torch.zeros(2, 10, dtype=torch.bool, device=DEV))
    v, i = F.masked_topk_desc(s, 0)
    assert i.shape == (2, 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.int64])
@pytest.mark.parametrize("C", [256, 8, 6])
def test_gather_scatter_rows(dtype, C):
    if dtype == torch.bfloat16 and C % 2:
        pytest.skip("rows must be a multiple of 4 bytes")
    B, S, n = 2, 1000, 613
    src = (syn.det_randn("rows", (B, S, C)) * 100).to(dtype)
    idx = torch.stack([torch.randperm(S, generator=torch.Generator().manual_seed(b))[:n] for b in range(B)])
    got = F.gather_rows(src.to(DEV), idx.to(DEV))
    assert torch.equal(got.cpu(), torch.gather(src, 1, idx[..., None].expand(-1, -1, C)))
    dst = torch.zeros(B, S, C, dtype=dtype)
    new = (syn.det_randn("new", (B, n, C)) * 100).to(dtype)
    count = torch.tensor([n, 100])
    expect = dst.clone()
    for b in range(B):
        expect[b, idx[b, :count[b]]] = new[b, :count[b]]
    d = dst.to(DEV)
    F.scatter_rows_(d, idx.to(DEV), new.to(DEV), count.to(DEV))
    assert torch.equal(d.cpu(), expect)
    d2 = dst.to(DEV)
    F.scatter_rows_(d2, idx.to(DEV), new.to(DEV))
    expect2 = dst.clone().scatter(1, idx[..., None].expand(-1, -1, C), new)
    assert torch.equal(d2.cpu(), expect2)
