"""The kernels `bench.py` actually times, pinned to the oracle at a tight bar.

The benchmark's encoder layers call the fused MSDA forward with fp16 head-major value maps and the bf16 HEAD-MAJOR
projection slab; above ``resident_min_queries`` that is ``msda_resident_kernel`` (levels 2+3 in LDS), below it
``msda_gather_l4p4_kernel``.  Each case asserts WHICH kernel ran (``sdetr_msda_last_kernel``) and compares its fp32
output with the plain-C oracle (``oracle/msda_oracle.c``: ms_deform_im2col_cuda.cuh:22-73, 226-288) fed the same
fp16- / bf16-rounded operands, the sampling locations / softmax of ms_deform_attn.py:322-355 restated in torch fp32.
Tolerance 2e-4 absolute on outputs of magnitude ~1 (fp32 accumulation order and the reciprocal in 1/W differ).
"""
import numpy as np
import pytest
import torch

from oracle import msda_c
from salience_detr_amd import ms_deform_attn as M
from salience_detr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LEVELS_FULL = [(100, 168), (50, 84), (25, 42), (13, 21)]
LEVELS_SMALL = [(20, 30), (10, 15), (5, 8), (3, 4)]
LEVELS_5SCALE = [(200, 336), (100, 168), (50, 84), (25, 42)]   # levels 2 + 3 exceed the LDS: level 3 alone is resident
LEVELS_L3_ONLY = [(40, 60), (40, 50), (30, 45), (16, 20)]       # a small pyramid that takes the same variant
HEADS, D, L, P = 8, 32, 4, 4
TOL = 2e-4


def _case(B, Nq, levels, ref_dim, seed):
    """value [B,Nv,M,D] fp32, proj rows [B,Nq,384] (bf16-rounded, as fp32), reference points fp32."""
    value, shapes, lsi, _, _ = syn.make_msda_inputs(B, 8, levels, HEADS, D, P, seed=seed)
    ring = syn._ring_bias(HEADS, L, P)                                   # the reference's offset initialisation
    off = ring[None, None] + syn.det_randn("timed.off", (B, Nq, HEADS * L * P * 2), salt=seed) * 1.5
    lgt = syn.det_randn("timed.lgt", (B, Nq, HEADS * L * P), salt=seed)
    proj = torch.cat([off, lgt], -1).to(torch.bfloat16)
    if ref_dim == 2:
        ref = syn.det_rand("timed.ref", (B, Nq, 1, 2), salt=seed).expand(B, Nq, L, 2) * 1.1 - 0.05   # some outside [0,1]
        ref = (ref + 0.01 * syn.det_randn("timed.refj", (B, Nq, L, 2), salt=seed)).contiguous()
    else:
        ref = torch.cat([syn.det_rand("timed.ref", (B, Nq, L, 2), salt=seed),
                         syn.det_rand("timed.wh", (B, Nq, L, 2), salt=seed) * 0.4 + 0.02], -1)
    return value, shapes, lsi, proj, ref


def _expected(value_q, shapes, lsi, ref, proj_f32):
    B, Nq = proj_f32.shape[:2]
    off = proj_f32[..., :HEADS * L * P * 2].view(B, Nq, HEADS, L, P, 2)
    aw = proj_f32[..., HEADS * L * P * 2:].view(B, Nq, HEADS, L * P).softmax(-1).view(B, Nq, HEADS, L, P)
    if ref.shape[-1] == 2:
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    return msda_c.msda_forward(value_q.numpy(), shapes.numpy(), lsi.numpy(), loc.contiguous().numpy(),
                               aw.contiguous().numpy())


def _head_major_slab(proj):
    """[B,Nq,384] token rows -> [B,M,Nq,48]: per head its 32 offsets then its 16 logits."""
    B, Nq, _ = proj.shape
    off = proj[..., :HEADS * L * P * 2].view(B, Nq, HEADS, L * P * 2)
    lgt = proj[..., HEADS * L * P * 2:].view(B, Nq, HEADS, L * P)
    return torch.cat([off, lgt], -1).permute(0, 2, 1, 3).contiguous()


CASES = [(2, 257, LEVELS_SMALL), (2, 2272, LEVELS_FULL), (2, 11363, LEVELS_FULL)]


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("vdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Nq,levels", CASES)
def test_l4p4_kernel_vs_oracle(B, Nq, levels, vdt, ref_dim):
    value, shapes, lsi, proj, ref = _case(B, Nq, levels, ref_dim, seed=Nq)
    Nv = value.shape[1]
    expect = _expected(value.to(vdt).float(), shapes, lsi, ref, proj.float())
    hm = M.value_to_head_major(value.view(B, Nv, HEADS * D).to(DEV), None, HEADS, vdt)
    dsh, dls, dref = shapes.to(DEV), lsi.to(DEV), ref.to(DEV)
    # (i) the head-major bf16 projection slab -- what the benchmark's encoder layers pass
    out = M.msda_fused_forward(hm, dsh, dls, dref, _head_major_slab(proj).to(DEV), L, P, out_dtype=torch.float32,
                               proj_head_major=True)
    assert M.last_forward_kernel() == M.KERNEL_L4P4
    assert np.abs(out.cpu().numpy() - expect).max() < TOL
    # (ii) bf16 token rows, row stride 384 (a multiple of 8: the specialised kernel, not the generic one)
    rows = proj.to(DEV)
    assert rows.stride(1) == 384
    out = M.msda_fused_forward(hm, dsh, dls, dref, rows, L, P, out_dtype=torch.float32)
    assert M.last_forward_kernel() == M.KERNEL_L4P4
    assert np.abs(out.cpu().numpy() - expect).max() < TOL
    # bf16 output (the benchmark's): the same numbers rounded once
    out16 = M.msda_fused_forward(hm, dsh, dls, dref, _head_major_slab(proj).to(DEV), L, P, out_dtype=torch.bfloat16,
                                 proj_head_major=True)
    assert torch.equal(out16.cpu(), torch.from_numpy(expect).to(torch.bfloat16)) or \
        (out16.float().cpu() - torch.from_numpy(expect)).abs().max() < 1e-2


@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("chunks", [0, 1, 7])
@pytest.mark.parametrize("B,Nq,levels", CASES + [(1, 40, LEVELS_SMALL), (3, 1000, LEVELS_FULL), (2, 777, LEVELS_L3_ONLY),
                                                 (1, 4533, LEVELS_5SCALE)])
def test_resident_kernel_vs_oracle(B, Nq, levels, chunks, ref_dim):
    if chunks and Nq > 3000:
        pytest.skip("chunk sweeps on the small cases only")
    value, shapes, lsi, proj, ref = _case(B, Nq, levels, ref_dim, seed=Nq + 1)
    Nv = value.shape[1]
    expect = _expected(value.to(torch.float16).float(), shapes, lsi, ref, proj.float())
    hm = M.value_to_head_major(value.view(B, Nv, HEADS * D).to(DEV), None, HEADS, torch.float16)
    slab = _head_major_slab(proj).to(DEV)
    out = M.msda_resident_forward(hm, levels, ref.to(DEV), slab, out_dtype=torch.float32, chunks=chunks)
    assert M.last_forward_kernel() == M.KERNEL_RESIDENT
    assert np.abs(out.cpu().numpy() - expect).max() < TOL
    # and it agrees with the direct kernel far inside the oracle bar (same arithmetic, different data path)
    direct = M.msda_fused_forward(hm, shapes.to(DEV), lsi.to(DEV), ref.to(DEV), slab, L, P, out_dtype=torch.float32,
                                  proj_head_major=True)
    assert (out - direct).abs().max().item() < 2e-5


@pytest.mark.parametrize("lanes", [1, 2, 3, 5])
@pytest.mark.parametrize("B,Nq,levels", [(5, 300, LEVELS_SMALL), (3, 1000, LEVELS_FULL), (4, 40, LEVELS_L3_ONLY)])
def test_resident_kernel_image_lanes(B, Nq, levels, lanes):
    """Large batches: a workgroup walks the images of its lane (g, g + G, ...) so that only G images' maps are gathered
    from at a time -- forced here on small batches (also with more lanes than images, with idle waves and with a
    partial last round); the same numbers as the one-image-per-workgroup form."""
    value, shapes, lsi, proj, ref = _case(B, Nq, levels, 2, seed=Nq + 7)
    Nv = value.shape[1]
    hm = M.value_to_head_major(value.view(B, Nv, HEADS * D).to(DEV), None, HEADS, torch.float16)
    slab = _head_major_slab(proj).to(DEV)
    want = M.msda_resident_forward(hm, levels, ref.to(DEV), slab, out_dtype=torch.float32, image_lanes=0)
    for chunks in (0, 1, 5):
        got = M.msda_resident_forward(hm, levels, ref.to(DEV), slab, out_dtype=torch.float32, chunks=chunks, image_lanes=lanes)
        assert torch.equal(got, want)
    expect = _expected(value.to(torch.float16).float(), shapes, lsi, ref, proj.float())
    assert np.abs(want.cpu().numpy() - expect).max() < TOL


def test_resident_kernel_batch_16_takes_the_image_lanes():
    """366 MB of value maps (batch 16 of the benchmark pyramid): the launch picks the image-lane form by itself; checked
    against the direct kernel (same arithmetic, different data path)."""
    B, Nq = 16, 1500
    value, shapes, lsi, proj, ref = _case(B, Nq, LEVELS_FULL, 2, seed=99)
    Nv = value.shape[1]
    hm = M.value_to_head_major(value.view(B, Nv, HEADS * D).to(DEV), None, HEADS, torch.float16)
    slab = _head_major_slab(proj).to(DEV)
    out = M.msda_resident_forward(hm, LEVELS_FULL, ref.to(DEV), slab, out_dtype=torch.float32)
    direct = M.msda_fused_forward(hm, shapes.to(DEV), lsi.to(DEV), ref.to(DEV), slab, L, P, out_dtype=torch.float32,
                                  proj_head_major=True)
    assert (out - direct).abs().max().item() < 2e-5


def test_resident_kernel_borders_and_wild_locations():
    """Samples on / beyond every border of every level, NaN-free zero contributions outside (the reference's
    `h_im > -1 && w_im > -1 && h_im < H && w_im < W` early-out, ms_deform_im2col_cuda.cuh:258-262)."""
    B, Nq, levels = 1, 512, LEVELS_SMALL
    value, shapes, lsi, proj, _ = _case(B, Nq, levels, 2, seed=5)
    Nv = value.shape[1]
    # reference points on a lattice that includes exactly 0, 1, pixel centres and far-out values
    t = torch.linspace(-0.2, 1.2, 32)
    gx, gy = torch.meshgrid(t, t[:16], indexing="xy")
    ref = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)[None, :, None, :].expand(B, Nq, L, 2).contiguous()
    proj = proj.float()
    proj[..., :HEADS * L * P * 2] = (proj[..., :HEADS * L * P * 2] * 0.5).round()   # whole-pixel offsets: exact borders
    proj[0, 0, :8] = 1e30   # wild offsets: every weight must be zero, no NaN
    proj[0, 1, 8:12] = -1e30
    proj[0, 2, 16:20] = float("inf")
    proj = proj.to(torch.bfloat16)
    expect = _expected(value.to(torch.float16).float(), shapes, lsi, ref, proj.float())
    hm = M.value_to_head_major(value.view(B, Nv, HEADS * D).to(DEV), None, HEADS, torch.float16)
    out = M.msda_resident_forward(hm, levels, ref.to(DEV), _head_major_slab(proj).to(DEV), out_dtype=torch.float32)
    assert torch.isfinite(out).all()
    assert np.abs(out.cpu().numpy() - expect).max() < TOL


def test_resident_rejects_unsupported():
    hm = torch.zeros(1, 8, 20 * 30 + 10 * 15 + 5 * 8 + 3 * 4, 32, dtype=torch.bfloat16, device=DEV)
    assert not M.resident_supported(hm, LEVELS_SMALL, 4, 4)            # bf16 maps stay on the direct kernel
    assert not M.resident_supported(hm.half(), LEVELS_SMALL[:3], 4, 4)
    # the 5scale pyramid: levels 2 + 3 do not fit together, level 3 does -> the level-3-only variant takes it
    assert M.resident_supported(torch.zeros(1, 1, sum(h * w for h, w in LEVELS_5SCALE), 32, dtype=torch.float16,
                                            device=DEV), LEVELS_5SCALE, 4, 4)
    big = [(400, 672), (200, 336), (100, 168), (50, 84)]                # a coarsest level of 4200 pixels: too large
    assert not M.resident_supported(torch.zeros(1, 1, sum(h * w for h, w in big), 32, dtype=torch.float16, device=DEV),
                                    big, 4, 4)
    with pytest.raises(RuntimeError):
        M.msda_resident_forward(hm, LEVELS_SMALL, torch.zeros(1, 4, 4, 2, device=DEV),
                                torch.zeros(1, 8, 4, 48, dtype=torch.bfloat16, device=DEV))
