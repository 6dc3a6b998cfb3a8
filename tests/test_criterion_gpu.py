"""GPU: row N4 -- SalienceCriterion (targets, focal loss, gradient) against the reference's vectors and the oracle.
Bars: targets 1e-6, loss 1e-5 relative, gradients 1e-6 (fp32)."""
import os

import numpy as np
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd.salience_criterion import SalienceCriterion
from test_criterion_cpu import TAGS, _t, case_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def cases():
    return np.load(os.path.join(G, "criterion_cases.npz"))


@pytest.mark.parametrize("tag", TAGS)
def test_criterion_matches_reference(cases, tag):
    d = cases
    masks, boxes, strides, image_sizes = case_inputs(d, tag, device="cuda", requires_grad=True)
    crit = SalienceCriterion()
    shapes = [tuple(m.shape[-2:]) for m in masks]
    target = crit.mask_targets([{"boxes": b} for b in boxes], shapes, strides, image_sizes, "cuda")
    assert (target.cpu() - _t(d[f"{tag}.mask_targets"])).abs().max() < 1e-6
    loss = crit(masks, [{"boxes": b} for b in boxes], strides, image_sizes)["loss_salience"]
    ref = float(d[f"{tag}.loss"])
    assert abs(float(loss.detach()) - ref) <= 1e-5 * abs(ref)
    (loss * 1.5).backward()
    for l, m in enumerate(masks):
        assert (m.grad.cpu() - 1.5 * _t(d[f"{tag}.grad{l}"])).abs().max() < 2e-6


def test_criterion_other_hyperparameters_and_noise_match_oracle():
    """gamma != 2 (the powf path), another alpha, three levels, noise mixing; many boxes (several LDS chunks)."""
    torch.manual_seed(0)
    shapes, strides = [(20, 30), (10, 15), (5, 8)], [(8.0, 8.0), (16.0, 16.0), (32.0, 30.0)]
    image_sizes = [(160, 240), (150, 200)]
    boxes = [torch.cat([torch.rand(300, 2) * 0.8 + 0.1, torch.rand(300, 2) * 0.5 + 0.02], -1), torch.zeros(0, 4)]
    masks = [torch.randn(2, 1, h, w) for h, w in shapes]
    limit = ((-1, 40), (40, 90), (90, 1e5))
    crit = SalienceCriterion(limit_range=limit, noise_scale=0.0, alpha=0.4, gamma=1.5)
    gm = [m.cuda().requires_grad_(True) for m in masks]
    loss = crit(gm, [{"boxes": b.cuda()} for b in boxes], strides, image_sizes)["loss_salience"]
    loss.backward()
    om = [m.clone().requires_grad_(True) for m in masks]
    oloss, otarget = R.salience_criterion(om, boxes, strides, image_sizes, limit_range=limit, alpha=0.4, gamma=1.5)
    oloss.backward()
    assert abs(float(loss.detach()) - float(oloss.detach())) <= 2e-5 * abs(float(oloss.detach()))
    for a, b in zip(gm, om):
        assert (a.grad.cpu() - b.grad).abs().max() < 2e-6
    # noise: the mix is checked with the noise tensor handed to the kernel directly
    from salience_detr_amd.salience_criterion import salience_targets
    xyxy = [torch.cat([(b[:, :2] - 0.5 * b[:, 2:]), (b[:, :2] + 0.5 * b[:, 2:])], -1)
            * torch.tensor([iw, ih, iw, ih], dtype=torch.float32) for b, (ih, iw) in zip(boxes, image_sizes)]
    noise = torch.rand(2, sum(h * w for h, w in shapes))
    got = salience_targets(torch.cat(xyxy).cuda(), torch.tensor([0, 300, 300], dtype=torch.int32).cuda(), shapes, strides,
                           limit, 0.3, noise.cuda())
    want = R.salience_targets(xyxy, shapes, strides, limit, 0.3, noise)
    assert (got.cpu() - want).abs().max() < 1e-6


def test_criterion_refuses_cpu_tensors():
    crit = SalienceCriterion()
    with pytest.raises(RuntimeError):
        crit([torch.zeros(1, 1, 4, 4)], [{"boxes": torch.zeros(0, 4)}], [(8.0, 8.0)], [(32, 32)])
