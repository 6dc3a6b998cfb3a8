"""MSDA backward with grad_value accumulated in LDS windows (msda_backward_tiled.hip) against the C oracle and the
direct global-atomic kernel.  Reference arithmetic: models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:76-148,290-392.
The test asserts which kernel ran (``last_backward_kernel``)."""
import numpy as np
import pytest
import torch

from oracle import msda_c
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"
LEVELS_SMALL = [(20, 30), (10, 15), (5, 8), (3, 4)]
LEVELS_FULL = [(100, 168), (50, 84), (25, 42), (13, 21)]
LEVELS_TILED = [(40, 70), (33, 40), (5, 8)]     # two tiled levels (2800 and 1320 pixels) and one held whole


def _smooth_mask(loc, shapes):
    # d(out)/d(loc) jumps where a sample sits on a pixel boundary; 1 ulp in loc*size-0.5 flips floor() there
    px = loc * torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()[None, None, None, :, None, :] - 0.5
    return ((px - px.round()).abs() > 1e-3).all(-1, keepdim=True).expand_as(loc).numpy()


def _run(lds, value, shapes, lsi, loc, aw, go):
    old = M.lds_backward, M.lds_backward_min_queries
    M.lds_backward, M.lds_backward_min_queries = lds, 1    # the kernel under test, whatever the query count
    try:
        out = M.ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in out], M.last_backward_kernel()
    finally:
        M.lds_backward, M.lds_backward_min_queries = old


@pytest.mark.parametrize("B,Nq,levels,M_,spread", [
    (2, 333, LEVELS_SMALL, 8, 4.0),      # every level held whole
    (1, 700, LEVELS_TILED, 3, 4.0),      # tiled levels, 3 heads
    (2, 1500, LEVELS_TILED, 8, 12.0),    # offsets beyond the halo: per-sample fallback to global atomics
    (2, 2272, LEVELS_FULL, 8, 6.0),      # encoder layer 5 at the benchmark shape
])
def test_lds_backward_vs_c_oracle(B, Nq, levels, M_, spread):
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, levels, M_, 32, 4, seed=3, spread_px=spread)
    go = syn.det_randn("gout_lds", (B, Nq, M_ * 32))
    rgv, rgl, rga = msda_c.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), go.numpy())
    dev = [t.to(DEV) for t in (value, shapes, lsi, loc, aw, go)]
    (gv, gl, ga), which = _run(True, *dev)
    assert which == M.KERNEL_BWD_LDS
    smooth = _smooth_mask(loc, shapes)
    assert smooth.mean() > 0.99
    assert np.abs(gv - rgv).max() < 2e-4 * max(1.0, np.abs(rgv).max())
    assert np.abs((gl - rgl) * smooth).max() < 2e-4 * max(1.0, np.abs(rgl).max())
    assert np.abs(ga - rga).max() < 2e-4 * max(1.0, np.abs(rga).max())


@pytest.mark.parametrize("Nq", [257, 11363])
def test_lds_backward_matches_direct_kernel_full_size(Nq):
    """Full benchmark size (B = 2, layer 0): the two kernels agree; the LDS one is the tighter of the two against a
    float64 accumulation of the same contributions (its window sums are exact integers)."""
    B, M_ = 2, 8
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, LEVELS_FULL, M_, 32, 4, seed=5, spread_px=4.0)
    go = syn.det_randn("gout_full", (B, Nq, M_ * 32))
    dev = [t.to(DEV) for t in (value, shapes, lsi, loc, aw, go)]
    (gv1, gl1, ga1), k1 = _run(True, *dev)
    (gv0, gl0, ga0), k0 = _run(False, *dev)
    assert (k1, k0) == (M.KERNEL_BWD_LDS, M.KERNEL_BWD_DIRECT)
    scale = max(1.0, np.abs(gv0).max())
    assert np.abs(gv1 - gv0).max() < 1e-4 * scale
    smooth = _smooth_mask(loc, shapes)
    assert np.abs((gl1 - gl0) * smooth).max() < 1e-4 * max(1.0, np.abs(gl0).max())
    assert np.abs(ga1 - ga0).max() < 1e-4 * max(1.0, np.abs(ga0).max())
    # linearity in grad_output (a size-independent property): backward(2 g) = 2 backward(g) exactly for grad_loc /
    # grad_aw (powers of two), and to fixed-point resolution for grad_value
    dev2 = list(dev)
    dev2[5] = dev[5] * 2
    (gv2, gl2, ga2), _ = _run(True, *dev2)
    assert np.array_equal(gl2, 2 * gl1) and np.array_equal(ga2, 2 * ga1)
    assert np.abs(gv2 - 2 * gv1).max() < 1e-5 * scale


def test_lds_backward_extreme_gradients():
    """Gradient magnitudes far from 1 (loss scaling, vanishing gradients) and all-zero gradients: the per-window
    scale follows the data; out-of-range bounds take the global-atomic path."""
    B, Nq, M_ = 1, 400, 8
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, LEVELS_SMALL, M_, 32, 4, seed=7, spread_px=3.0)
    go = syn.det_randn("gout_x", (B, Nq, M_ * 32))
    base = msda_c.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), go.numpy())[0]
    for factor in (0.0, 2.0 ** -60, 2.0 ** 40, 2.0 ** 100):
        dev = [t.to(DEV) for t in (value, shapes, lsi, loc, aw, go * factor)]
        (gv, _, _), which = _run(True, *dev)
        assert which == M.KERNEL_BWD_LDS
        if factor == 0.0:
            assert not gv.any()
        else:
            assert np.abs(gv / factor - base).max() < 2e-4 * max(1.0, np.abs(base).max())


def test_lds_backward_heavy_tailed_gradients_keep_the_small_ones():
    """the window's fixed-point quantum follows the LARGEST row bound of a (level, image, head).  One
    outlier query (here 2^28 times the others) must not round everybody else's contributions away: rows far below the
    bound take the fp32 path.  Checked on pixels the outlier's samples do not touch, against the C oracle."""
    B, Nq, M_ = 1, 600, 8
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(B, Nq, LEVELS_SMALL, M_, 32, 4, seed=11, spread_px=2.0)
    go = syn.det_randn("gout_tail", (B, Nq, M_ * 32))
    base = msda_c.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), go.numpy())[0]
    go_tail = go.clone()
    go_tail[0, 17] *= 2.0 ** 28
    only = torch.zeros_like(go)
    only[0, 17] = go_tail[0, 17]
    spike = msda_c.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), only.numpy())[0]
    untouched = spike == 0            # grad_value entries that receive nothing from the outlier row
    assert untouched.mean() > 0.5
    dev = [t.to(DEV) for t in (value, shapes, lsi, loc, aw, go_tail)]
    (gv, _, _), which = _run(True, *dev)
    assert which == M.KERNEL_BWD_LDS
    # without query 17 the oracle's sum on those entries is the sum of the ordinary rows (query 17 adds exact zeros)
    rest = go.clone()
    rest[0, 17] = 0
    want = msda_c.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), aw.numpy(), rest.numpy())[0]
    err = np.abs(gv - want)[untouched].max()
    assert err < 2e-4 * max(1.0, np.abs(base).max()), err
    # and the spike itself arrives
    assert np.abs(gv - (want + spike))[~untouched].max() < 1e-5 * np.abs(spike).max()
