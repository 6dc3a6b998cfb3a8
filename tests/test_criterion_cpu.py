"""CPU: row N4 -- the oracle's restatement of SalienceCriterion (targets, focal loss, gradient) against vectors from the
imported reference (tests/golden/criterion_cases.npz).  Bars: targets 1e-6, loss 1e-5 relative, gradients 1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import salience_ref as R

G = os.path.join(os.path.dirname(__file__), "golden")
TAGS = ["small", "empty", "full"]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def cases():
    return np.load(os.path.join(G, "criterion_cases.npz"))


def case_inputs(d, tag, device="cpu", requires_grad=False):
    shapes = [tuple(r) for r in d[f"{tag}.level_shapes"].tolist()]
    masks = [_t(d[f"{tag}.logits{l}"]).to(device).requires_grad_(requires_grad) for l in range(len(shapes))]
    boxes = [_t(d[f"{tag}.boxes{b}"]).to(device) for b in range(len(d[f"{tag}.num_boxes"]))]
    strides = [tuple(r) for r in d[f"{tag}.strides"].tolist()]
    image_sizes = [tuple(r) for r in d[f"{tag}.image_sizes"].tolist()]
    return masks, boxes, strides, image_sizes


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_criterion_matches_reference(cases, tag):
    d = cases
    masks, boxes, strides, image_sizes = case_inputs(d, tag, requires_grad=True)
    loss, target = R.salience_criterion(masks, boxes, strides, image_sizes)
    assert (target - _t(d[f"{tag}.mask_targets"])).abs().max() < 1e-6
    ref = float(d[f"{tag}.loss"])
    assert abs(float(loss.detach()) - ref) <= 1e-5 * abs(ref)
    loss.backward()
    for l, m in enumerate(masks):
        assert (m.grad - _t(d[f"{tag}.grad{l}"])).abs().max() < 1e-6


def test_oracle_noise_mix():
    shapes, strides = [(4, 6), (2, 3)], [(8.0, 8.0), (16.0, 16.0)]
    boxes = [torch.tensor([[5.0, 3.0, 40.0, 30.0]]), torch.zeros(0, 4)]
    base = R.salience_targets(boxes, shapes, strides, ((-1, 64), (64, 128)))
    noise = torch.rand(2, 30, generator=torch.Generator().manual_seed(0))
    mixed = R.salience_targets(boxes, shapes, strides, ((-1, 64), (64, 128)), noise_scale=0.2, noise=noise)
    assert torch.allclose(mixed, 0.8 * base + 0.2 * noise)
    assert (base[1] == 0).all() and (base[0] > 0).any()
