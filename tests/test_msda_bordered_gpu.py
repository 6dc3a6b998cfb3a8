"""The round-4 MSDA forward kernel (bordered head-major maps, optional row order) against the plain-C oracle.

Same operands, oracle and bar (2e-4) as tests/test_msda_timed_kernels_gpu.py: fp16 maps, bf16 head-major projection slab,
fp32 output; the maps go through ``to_bordered`` (zero border records around every level, include/salience_hip.h).  The
reference's corner tests (ms_deform_im2col_cuda.cuh:31-66, 258-262) are replaced by clamping the position onto the border:
positions on / beyond every edge, NaN-free wild offsets and both reference-point forms are covered here.
"""
import os

import numpy as np
import pytest
import torch

from salience_detr_amd import ms_deform_attn as M
from tests.test_msda_timed_kernels_gpu import (CASES, D, DEV, HEADS, L, LEVELS_5SCALE, LEVELS_FULL, LEVELS_L3_ONLY,
                                               LEVELS_SMALL, P, TOL, _case, _expected, _head_major_slab)

pytestmark = pytest.mark.gpu


def _maps(value, levels):
    B, Nv = value.shape[:2]
    hm = M.value_to_head_major(value.view(B, Nv, HEADS * D).to(DEV), None, HEADS, torch.float16)
    return hm, M.to_bordered(hm, levels)


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("chunks", [0, 1, 7])
@pytest.mark.parametrize("B,Nq,levels", CASES + [(1, 40, LEVELS_SMALL), (3, 1000, LEVELS_FULL), (2, 777, LEVELS_L3_ONLY),
                                                 (1, 4533, LEVELS_5SCALE)])
def test_bordered_kernel_vs_oracle(B, Nq, levels, chunks, ref_dim):
    if chunks and Nq > 3000:
        pytest.skip("chunk sweeps on the small cases only")
    value, shapes, lsi, proj, ref = _case(B, Nq, levels, ref_dim, seed=Nq + 1)
    expect = _expected(value.to(torch.float16).float(), shapes, lsi, ref, proj.float())
    hm, hb = _maps(value, levels)
    slab = _head_major_slab(proj).to(DEV)
    out = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, out_dtype=torch.float32, chunks=chunks)
    assert M.last_forward_kernel() == M.KERNEL_BORDERED
    assert np.abs(out.cpu().numpy() - expect).max() < TOL
    # the round-3 kernel on the plain maps: same arithmetic up to the order of the weight products and of the position
    # arithmetic (ref * W + 0.5 + offset here, (ref + offset / W) * W - 0.5 there: ~W * 2^-23 pixels apart)
    plain = M.msda_resident_forward(hm, levels, ref.to(DEV), slab, out_dtype=torch.float32)
    assert (out - plain).abs().max().item() < 1e-4
    # any row order gives the same bits
    g = torch.Generator().manual_seed(Nq)
    order = torch.stack([torch.randperm(Nq, generator=g) for _ in range(B)]).to(torch.int32).to(DEV)
    again = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, row_order=order, out_dtype=torch.float32, chunks=chunks)
    assert torch.equal(again, out)
    # bf16 output with the exact fp32 corner products (accumulate = ACC_EXACT) = the fp32 result rounded
    b16 = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, row_order=order, out_dtype=torch.bfloat16, chunks=chunks,
                                  accumulate=M.ACC_EXACT)
    assert torch.equal(b16, out.to(torch.bfloat16))
    # 16-bit outputs since round 5: a sample's four corners combined in packed fp16 (PK = 1; four fp16 roundings of every
    # interpolated sample on top of the maps' own), PK = 2 (the library's default for bf16 outputs): a level's four points too
    for pk, bar_mean, bar_max in ((M.ACC_DEFAULT, 1.5e-4, 4e-3), (M.ACC_PACKED_SAMPLE, 4e-5, 1.5e-3),
                                  (M.ACC_PACKED_LEVEL, 1.5e-4, 4e-3)):
        got = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, row_order=order, out_dtype=torch.bfloat16, chunks=chunks,
                                      accumulate=pk)
        # against the fp32 result, beyond the bf16 rounding of the output itself
        excess = ((got.float() - out).abs() - out.abs() * 2.0 ** -8).clamp_(min=0)
        assert excess.mean().item() < bar_mean and excess.max().item() < bar_max, (pk, excess.mean().item(), excess.max().item())


@pytest.mark.parametrize("scale", [1.0, 60.0, 2000.0])
def test_packed_fp16_accumulation_scales_with_the_maps(scale):
    """The timed 16-bit form sums a level's 16 corner products in packed fp16 (ACC_PACKED_LEVEL; ADVICE r5: bound it against
    the exact form on large-magnitude maps).  The partial sums are convex combinations of the map's values (bilinear
    weights sum to 1, attention weights to <= 1), so they cannot overflow half's range while the map itself fits it, and
    their rounding is relative: the excess over the exact fp32 sums (beyond the bf16 rounding of the output) stays below
    4e-3 of the map's scale at every magnitude -- including maps close to half's maximum (65 504)."""
    B, Nq, levels = 2, 2272, LEVELS_FULL
    value, shapes, lsi, proj, ref = _case(B, Nq, levels, 2, seed=17)
    value = value * scale                                           # |v| up to ~5 * scale: 10 000 at scale 2000
    assert value.abs().max() < 60000
    _, hb = _maps(value, levels)
    slab = _head_major_slab(proj).to(DEV)
    order = M.spatial_row_order(torch.stack([torch.randperm(sum(h * w for h, w in levels))[:Nq] for _ in range(B)]).to(DEV),
                                levels, 16)
    exact = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, row_order=order, out_dtype=torch.float32)
    assert torch.isfinite(exact).all()
    for acc, bar_mean, bar_max in ((M.ACC_PACKED_SAMPLE, 4e-5, 1.5e-3), (M.ACC_PACKED_LEVEL, 1.5e-4, 4e-3)):
        got = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, row_order=order, out_dtype=torch.bfloat16, accumulate=acc)
        assert torch.isfinite(got.float()).all()
        excess = ((got.float() - exact).abs() - exact.abs() * 2.0 ** -8).clamp_(min=0) / scale
        assert excess.mean().item() < bar_mean and excess.max().item() < bar_max, (scale, acc, excess.mean().item(), excess.max().item())
    # and the exact form under a 16-bit output is the fp32 result rounded, at every magnitude
    b16 = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, row_order=order, out_dtype=torch.bfloat16, accumulate=M.ACC_EXACT)
    assert torch.equal(b16, exact.to(torch.bfloat16))


def test_bordered_spatial_row_order_is_a_permutation_that_groups_tiles():
    levels = LEVELS_FULL
    Nv = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(3)
    tok = torch.stack([torch.randperm(Nv, generator=g)[:5000] for _ in range(2)]).to(DEV)
    for tile in (8, 16, 32):
        order = M.spatial_row_order(tok, levels, tile)
        assert order.dtype == torch.int32 and order.shape == tok.shape
        assert torch.equal(order.long().sort(1).values, torch.arange(5000, device=DEV).expand(2, -1))
        pos = M.tile_major_positions(levels, tile).to(DEV)
        walked = pos[torch.gather(tok, 1, order.long())]
        assert (walked[:, 1:] > walked[:, :-1]).all()


@pytest.mark.parametrize("lanes", [1, 2, 3, 5])
@pytest.mark.parametrize("B,Nq,levels", [(5, 300, LEVELS_SMALL), (3, 1000, LEVELS_FULL), (4, 40, LEVELS_L3_ONLY)])
def test_bordered_kernel_image_lanes(B, Nq, levels, lanes):
    value, shapes, lsi, proj, ref = _case(B, Nq, levels, 2, seed=Nq + 7)
    _, hb = _maps(value, levels)
    slab = _head_major_slab(proj).to(DEV)
    want = M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, out_dtype=torch.float32, image_lanes=0)
    order = torch.stack([torch.randperm(Nq) for _ in range(B)]).to(torch.int32).to(DEV)
    for chunks in (0, 1, 5):
        assert torch.equal(M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, out_dtype=torch.float32, chunks=chunks,
                                                   image_lanes=lanes), want)
        assert torch.equal(M.msda_bordered_forward(hb, levels, ref.to(DEV), slab, row_order=order, out_dtype=torch.float32,
                                                   chunks=chunks, image_lanes=lanes), want)
    expect = _expected(value.to(torch.float16).float(), shapes, lsi, ref, proj.float())
    assert np.abs(want.cpu().numpy() - expect).max() < TOL


def test_bordered_kernel_borders_and_wild_locations():
    """Samples on / beyond every border of every level and wild offsets (the reference's early-out,
    ms_deform_im2col_cuda.cuh:258-262): finite output, zero contribution outside."""
    B, Nq, levels = 1, 512, LEVELS_SMALL
    value, shapes, lsi, proj, _ = _case(B, Nq, levels, 2, seed=5)
    t = torch.linspace(-0.2, 1.2, 32)
    gx, gy = torch.meshgrid(t, t[:16], indexing="xy")
    ref = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)[None, :, None, :].expand(B, Nq, L, 2).contiguous()
    proj = proj.float()
    proj[..., :HEADS * L * P * 2] = (proj[..., :HEADS * L * P * 2] * 0.5).round()   # whole-pixel offsets: exact borders
    proj[0, 0, :8] = 1e30
    proj[0, 1, 8:12] = -1e30
    proj[0, 2, 16:20] = float("inf")
    proj[0, 3, 20:24] = float("-inf")
    proj = proj.to(torch.bfloat16)
    expect = _expected(value.to(torch.float16).float(), shapes, lsi, ref, proj.float())
    _, hb = _maps(value, levels)
    out = M.msda_bordered_forward(hb, levels, ref.to(DEV), _head_major_slab(proj).to(DEV), out_dtype=torch.float32)
    assert torch.isfinite(out).all()
    assert np.abs(out.cpu().numpy() - expect).max() < TOL
    # NaN offsets / reference points: the reference's comparisons fail -> no contribution from that sample
    proj2 = proj.float()
    proj2[0, 5, :2] = float("nan")          # head 0, level 0, point 0
    ref2 = ref.clone()
    ref2[0, 7, 1, 0] = float("nan")         # row 7, level 1: its four points in every head
    out2 = M.msda_bordered_forward(hb, levels, ref2.to(DEV), _head_major_slab(proj2.to(torch.bfloat16)).to(DEV),
                                   out_dtype=torch.float32)
    assert torch.isfinite(out2).all()


def test_bordered_rejects_unsupported():
    hm = torch.zeros(1, 8, sum(h * w for h, w in LEVELS_SMALL), 32, dtype=torch.float16, device=DEV)
    ref = torch.zeros(1, 4, 4, 2, device=DEV)
    slab = torch.zeros(1, 8, 4, 48, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        M.msda_bordered_forward(hm, LEVELS_SMALL, ref, slab)                       # plain maps
    hb = M.to_bordered(hm, LEVELS_SMALL)
    with pytest.raises(RuntimeError):
        M.msda_bordered_forward(hb.to(torch.bfloat16), LEVELS_SMALL, ref, slab)    # bf16 maps
    with pytest.raises(RuntimeError):
        M.msda_bordered_forward(hb, LEVELS_SMALL, ref, slab, row_order=torch.zeros(1, 4, dtype=torch.int64, device=DEV))
    assert M.bordered_supported(LEVELS_5SCALE, 4, 4)
    assert not M.bordered_supported([(400, 672), (200, 336), (100, 168), (50, 84)], 4, 4)


def test_value_projection_writes_the_bordered_layout():
    """``value_proj_head_major(..., bordered_levels=...)`` = the plain projection moved record by record into the bordered
    layout, borders zeroed by the launch itself (the destination starts as garbage)."""
    from salience_detr_amd import filter_ops as K
    from salience_detr_amd import synthetic as syn
    for levels, B, groups in ((LEVELS_SMALL, 2, 3), (LEVELS_FULL, 2, 6), (LEVELS_L3_ONLY, 1, 1)):
        Nv = sum(h * w for h, w in levels)
        x = syn.det_randn("bordered.x", (B, Nv, 256), salt=Nv).to(torch.bfloat16).to(DEV)
        w = (syn.det_randn("bordered.w", (groups * 256, 256), salt=Nv) * 0.05).to(torch.bfloat16).to(DEV)
        b = syn.det_randn("bordered.b", (groups * 256,), salt=Nv).to(torch.bfloat16).to(DEV)
        pad = (syn.det_rand("bordered.pad", (B, Nv), salt=Nv) < 0.1).to(DEV)
        plain = K.value_proj_head_major(x, w, b, pad, HEADS, groups, torch.float16)
        # poison the allocator's next block so that unwritten records would show
        junk = torch.full((groups, B, HEADS, M.bordered_layout(levels).records, 32), float("nan"), dtype=torch.float16, device=DEV)
        del junk
        got = K.value_proj_head_major(x, w, b, pad, HEADS, groups, torch.float16, bordered_levels=levels)
        assert got.shape == (groups, B, HEADS, M.bordered_layout(levels).records, 32)
        assert torch.equal(got, M.to_bordered(plain, levels))
        # the same through the job form the hot path uses (slices carried by other launches, run on their own here)
        maps, jobs = K.plan_value_projection(x, w, b, pad, HEADS, groups, torch.float16, parts=min(groups, 4),
                                             bordered_levels=levels)
        for job in jobs:
            job.run()
        assert torch.equal(maps, got)


@pytest.mark.parametrize("tile", [8, 16])
def test_layer_row_orders_match_the_torch_reference(tile):
    from salience_detr_amd import filter_ops as K
    levels = LEVELS_FULL
    Nv = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(tile)
    n0 = 11363
    counts = [11363, 9090, 6817, 6817, 4545, 2272]
    sorted_index = torch.stack([torch.randperm(Nv, generator=g)[:n0] for _ in range(2)]).to(DEV)
    orders = K.layer_row_orders(sorted_index, counts, levels, tile=tile)
    assert len(orders) == 6
    for c, order in zip(counts, orders):
        assert order.shape == (2, c) and order.dtype == torch.int32
        want = M.spatial_row_order(sorted_index[:, :c], levels, tile)
        assert torch.equal(order, want)
    small = K.layer_row_orders(sorted_index[:, :40].contiguous(), [40, 7], LEVELS_FULL, tile=tile)
    assert torch.equal(small[1], M.spatial_row_order(sorted_index[:, :7], levels, tile))


def test_layer_row_orders_large_pyramid_takes_several_passes():
    """Round 5: the reference's 5scale pyramid (89 250 tokens: 178 KB of 16-bit slots, more than a workgroup's LDS) is
    sorted in two passes over the slot array -- BASELINE configs[3]'s first layer has 45 330 rows.  Same torch reference."""
    from salience_detr_amd import filter_ops as K
    levels = LEVELS_5SCALE
    Nv = sum(h * w for h, w in levels)
    assert Nv == 89250
    g = torch.Generator().manual_seed(5)
    counts = [45330, 36264, 27198, 27198, 18132, 9066]
    sorted_index = torch.stack([torch.randperm(Nv, generator=g)[:counts[0]] for _ in range(2)]).to(DEV)
    orders = K.layer_row_orders(sorted_index, counts, levels, tile=16)
    assert orders is not None and len(orders) == 6
    for c, order in zip(counts, orders):
        assert torch.equal(order, M.spatial_row_order(sorted_index[:, :c], levels, 16))
    # the carried form (a job riding in a top-k launch whose LDS limit is lower) gives the same orders
    job = K.layer_row_orders(sorted_index, counts, levels, tile=16, as_job=True)
    score = torch.rand(2, 4000, generator=g).to(DEV)
    K.masked_topk_desc(score, 300, orders_job=job)
    job.run()
    for c, order in zip(counts, job.orders):
        assert torch.equal(order, M.spatial_row_order(sorted_index[:, :c], levels, 16))


def test_layer_row_orders_stay_permutations_on_duplicate_or_stray_tokens():
    """a caller-supplied list with a token twice, or tokens outside the pyramid, cannot be counting-sorted (one
    row per slot) -- but the order must still be a permutation of the rows (the gather writes exactly the rows it lists).
    A row that loses its slot to a duplicate follows its part's run, rows with stray tokens close the order; every other
    row keeps its tile-major place."""
    from salience_detr_amd import filter_ops as K
    levels = LEVELS_FULL
    Nv = sum(h * w for h, w in levels)
    g = torch.Generator().manual_seed(9)
    good = torch.randperm(Nv, generator=g)[:2000]
    dup = good.clone()
    dup[1500] = dup[5]                    # one token twice (rows 5 and 1500: the 900-row prefix stays clean)
    wild = good.clone()
    wild[1234] = Nv + 3                   # one token outside the pyramid
    wild[77] = -1
    sorted_index = torch.stack([good, dup, wild]).to(DEV)
    orders = K.layer_row_orders(sorted_index, [2000, 900], levels, tile=16)
    ident = torch.arange(2000, dtype=torch.int32)
    pos = M.tile_major_positions(levels, 16)
    assert torch.equal(orders[0][0], M.spatial_row_order(sorted_index[:1], levels, 16)[0])
    for b in (1, 2):
        assert torch.equal(orders[0][b].cpu().sort()[0], ident)                      # a permutation
    # duplicate: without rows 5 and 1500 the order is the clean list's without them
    o = orders[0][1].cpu()
    want = M.spatial_row_order(sorted_index[:1], levels, 16)[0].cpu()
    keep = lambda t: t[(t != 5) & (t != 1500)]
    assert torch.equal(keep(o), keep(want))
    # stray tokens: the valid rows in tile order, then the stray rows in row order
    o = orders[0][2].cpu()
    rows = torch.tensor([r for r in range(2000) if r not in (77, 1234)])
    ref = rows[pos[wild[rows]].argsort(stable=True)].to(torch.int32)
    assert torch.equal(o[:-2], ref) and o[-2:].tolist() == [77, 1234]
    assert torch.equal(orders[1][1], M.spatial_row_order(sorted_index[1:2, :900], levels, 16)[0])   # the prefix is clean
