"""CPU: the C-ABI library builds, loads and exports every symbol the header declares; the product
path refuses to run without a device (no CPU fallback); host-side argument checks."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from salience_detr_amd.csrc import build
    return build.build()


def test_header_symbols_exported(libpath):
    header = open(os.path.join(ROOT, "include", "salience_hip.h")).read()
    declared = set(re.findall(r"\b(sdetr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 12
    cdll = ctypes.CDLL(libpath)
    for name in declared:
        assert hasattr(cdll, name), name
    from salience_detr_amd import _hip
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    assert _hip.lib().sdetr_abi_version() == 1


def test_fp16_flavour_exports_the_same_abi(libpath):
    """libsalience_hip_f16.so (round 5: the same sources with -DSDETR_ACT_F16, IEEE-half activations for BASELINE configs[4])
    is built next to the bf16 library, exports exactly the same `sdetr_*` symbols, loads beside it without the two seeing
    each other, and `_hip.lib(dtype)` hands out the one that matches the activations."""
    import subprocess
    from salience_detr_amd.csrc import build
    from salience_detr_amd import _hip
    assert os.path.exists(build.F16_LIB) and os.path.dirname(build.F16_LIB) == os.path.dirname(libpath)

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("sdetr_")}
    assert exported(build.F16_LIB) == exported(libpath)
    a, b = _hip.lib(), _hip.lib(torch.float16)
    assert a is not b and a is _hip.lib(torch.bfloat16) and a is _hip.lib(torch.float32) and a is _hip.lib(None)
    assert b.sdetr_abi_version() == 1
    # each library answers with its own state: an error raised in one is not visible in the other
    assert b.sdetr_msda_bordered_forward(None, None, 2, None, None, 2, 0, None, None, 0, 1, -1, 8, 4, None, 1, 0) != 0
    assert b"bad dims" in b.sdetr_last_error() and b"bad dims" not in a.sdetr_last_error()


def test_product_library_carries_no_benchmark_instantiations(libpath):
    """The deliberately crippled MSDA instantiations (wrong results by construction) and the phase-stamp hook exist only in
    the benchmark build (`csrc/build.py --ablations`): the product library does not export the hook, and the only exported
    `sdetr_*` symbols are the ones the header declares."""
    import subprocess
    header = open(os.path.join(ROOT, "include", "salience_hip.h")).read()
    declared = set(re.findall(r"\b(sdetr_[a-z0-9_]+)\s*\(", header))
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("sdetr_")}
    assert "sdetr_msda_debug_stamps" not in exported
    internal = {"sdetr_topk_inproj_launch"}   # (library-internal launch helper shared between translation units; bound by _hip)
    assert exported - declared - internal == set(), exported - declared - internal


def test_argument_rejection_without_gpu(libpath):
    """Invalid arguments are rejected on the host before any launch (safe without a GPU)."""
    from salience_detr_amd import _hip
    lib = _hip.lib()
    assert lib.sdetr_masked_topk_desc_f32(None, None, None, 0, 0, None, None, 2, 10, 11, 0, None, None, 0, None, 0) == _hip.EINVAL
    assert b"out of range" in lib.sdetr_last_error()
    assert lib.sdetr_msda_im2col_f32(None, None, None, None, None, None, 1, 1, 0, 32, 4, 1, 4, None) == _hip.EINVAL
    assert lib.sdetr_msda_fused_forward(None, 1, 0, 1, 1, 1, 3, 0, 1, 0, 384, 0, None, 1, 1, 8, 32, 4, 1, 4, 1, 0) == _hip.EINVAL
    assert b"must be 2 or 4" in lib.sdetr_last_error()
    assert lib.sdetr_topk_workspace_bytes(2, 1050, 1050) == 16  # one float: the masked-fill value
    assert lib.sdetr_topk_workspace_bytes(2, 11363, 300) == 16  # the one-launch histogram sort keeps its lists in LDS
    assert lib.sdetr_topk_workspace_bytes(2, 16800, 6680) == 16
    assert lib.sdetr_topk_workspace_bytes(2, 20000, 6680) == 16 + 2 * 20000 * 8 + 2 * 4 + 16  # + prefilter candidates
    assert lib.sdetr_topk_workspace_bytes(0, 5, 1) == 0
    # neck (13)
    assert lib.sdetr_neck_conv3x3(None, None, 0, 1, 4, 4, 32, None, None, 4, 8, 8, 3, 0, None) == _hip.EINVAL
    assert b"stride" in lib.sdetr_last_error()
    assert lib.sdetr_neck_conv3x3(None, None, 0, 1, 4, 4, 32, None, None, 4, 6, 8, 1, 0, None) == _hip.EINVAL  # 6 % 4
    assert lib.sdetr_neck_conv3x3_mfma_bf16(None, None, 1, 4, 4, 32, None, None, 4, 8, 8, 1, 0, None) == _hip.EINVAL
    assert b"in_per_group" in lib.sdetr_last_error()
    assert lib.sdetr_neck_pack_conv3x3_bf16(None, None, 4, 8, 8, None) == _hip.EINVAL
    assert lib.sdetr_neck_combine(None, None, 6, None, 0, 0, 0, None, 0, 1, 2, 2, 6, 1, None, 6) == _hip.EINVAL
    assert lib.sdetr_neck_gate_shortcut(None, None, 0, 1, 16, 512, None, None, None, 32, None, 512, None, 0, None, 0,
                                        None, None) == _hip.EINVAL
    assert b"at most 256" in lib.sdetr_last_error()
    assert lib.sdetr_neck_conv3x3(None, None, 0, 0, 4, 4, 32, None, None, 4, 8, 8, 1, 0, None) == 0   # empty batch


def test_no_cpu_fallback():
    from salience_detr_amd import filter_ops, ms_deform_attn as M
    from salience_detr_amd import synthetic as syn
    value, shapes, lsi, loc, aw = syn.make_msda_inputs(1, 5, [(4, 4), (2, 2)], 2, 8, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
    with pytest.raises(RuntimeError):
        filter_ops.masked_topk_desc(torch.zeros(1, 4), 2)
    mod = M.MultiScaleDeformableAttention(32, 2, 4, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mod(torch.zeros(1, 5, 32), torch.zeros(1, 5, 2, 2), torch.zeros(1, 20, 32), shapes, lsi, None)


def test_module_boundary_contract():
    """B3: constructor errors, parameter names, default initialisation (reference ms_deform_attn.py:221-284)."""
    from salience_detr_amd.ms_deform_attn import MultiScaleDeformableAttention
    with pytest.raises(ValueError):
        MultiScaleDeformableAttention(embed_dim=30, num_heads=8)
    m = MultiScaleDeformableAttention(256, 4, 8, 4)
    assert sorted(m.state_dict()) == sorted(
        f"{n}.{p}" for n in ("sampling_offsets", "attention_weights", "value_proj", "output_proj")
        for p in ("weight", "bias"))
    assert m.sampling_offsets.weight.abs().max() == 0 and m.attention_weights.bias.abs().max() == 0
    bias = m.sampling_offsets.bias.view(8, 4, 4, 2)
    assert torch.allclose(bias[0, :, :, 0], torch.arange(1.0, 5.0).expand(4, 4))  # head 0 points along +x
    assert torch.allclose(bias[:, :, 3].abs().max(-1)[0], torch.full((8, 4), 4.0))
    assert sum(p.numel() for p in m.parameters()) == 230272


def test_host_side_size_helpers_need_no_gpu():
    """Pure host functions of the C ABI: workspace / packed sizes and the feed-forward split heuristic (which falls back
    to 256 compute units when no device can be queried)."""
    from salience_detr_amd import _hip
    lib = _hip.lib()
    assert lib.sdetr_ffn_packed_bytes(2048) == 64 * 32768
    assert lib.sdetr_ffn_workspace_bytes(1000, 1) == 0 and lib.sdetr_ffn_workspace_bytes(1000, 3) == 3 * 1000 * 256 * 4
    assert lib.sdetr_ffn_auto_splits(22726, 2048) == 1          # 178 token blocks already fill most of the chip
    assert lib.sdetr_ffn_auto_splits(13634, 2048) == 2
    assert 4 <= lib.sdetr_ffn_auto_splits(1800, 2048) <= 16
    assert lib.sdetr_ffn_auto_splits(0, 2048) == 1
    assert lib.sdetr_linear_packed_bytes(91) == 65536 and lib.sdetr_linear_packed_bytes(384) == 3 * 65536
    assert lib.sdetr_focal_loss_workspace_bytes(0) == 0 and lib.sdetr_focal_loss_workspace_bytes(10 ** 9) == 1024 * 8
    assert lib.sdetr_topk_workspace_bytes(2, 20000, 6680) > 2 * 20000 * 8 and lib.sdetr_topk_workspace_bytes(2, 11363, 300) == 16
    # neck (13): pooling partials = ceil(pixels / pixels_per_block) x (channels + 2) floats per image, the block halved from
    # 128 pixels down to 16 until the launch has 512 workgroups (round 5); MFMA weight fragments in bf16
    assert lib.sdetr_neck_gate_workspace_bytes(2, 16800, 256) == 2 * 263 * 258 * 4         # 64-pixel blocks
    assert lib.sdetr_neck_gate_workspace_bytes(64, 16800, 256) == 64 * 132 * 258 * 4       # 128-pixel blocks fill the chip
    assert lib.sdetr_neck_gate_workspace_bytes(2, 273, 256) == 2 * 18 * 258 * 4            # the floor: 16 pixels
    assert lib.sdetr_neck_gate_workspace_bytes(0, 16800, 256) == 0
    assert lib.sdetr_neck_conv3x3_packed_bytes(4, 64, 64) == 4 * 9 * 64 * 64 * 2
    assert lib.sdetr_neck_conv3x3_packed_bytes(1, 256, 256) == 9 * 256 * 256 * 2
    assert lib.sdetr_neck_conv3x3_packed_bytes(4, 8, 8) == 0      # shape the matrix-core kernel does not take


def test_criterion_and_attention_refuse_cpu_tensors():
    import torch
    from salience_detr_amd.filter_ops import attention_heads
    from salience_detr_amd.salience_criterion import SalienceCriterion
    with pytest.raises(RuntimeError):
        SalienceCriterion()([torch.zeros(1, 1, 4, 4)], [{"boxes": torch.zeros(0, 4)}], [(8.0, 8.0)], [(32, 32)])
    x = torch.zeros(1, 8, 256, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        attention_heads(x, x, x, 8)


def test_round2_entry_points_reject_bad_arguments(libpath):
    """Host-side checks of the entries added in round 2 (no launch is reached, safe without a GPU)."""
    from salience_detr_amd import _hip
    lib = _hip.lib()
    L = 2
    ptrs = (ctypes.c_void_p * L)(8, 8)              # non-null dummies: rejected before anything dereferences them
    hs, ws = (ctypes.c_int * L)(4, 2), (ctypes.c_int * L)(4, 2)
    flat = lambda levels, S, p=ptrs, h=hs, w=ws: lib.sdetr_pyramid_flatten(
        None, levels, p, p, p, h, w, 8, 1, 64, S, None, None, 8, 8, None, None, None)
    assert flat(0, 20) == _hip.EINVAL and b"levels" in lib.sdetr_last_error()
    assert flat(9, 20) == _hip.EINVAL
    assert flat(L, 21) == _hip.EINVAL and b"pixel count" in lib.sdetr_last_error()      # 4*4 + 2*2 = 20
    assert flat(L, 20, h=(ctypes.c_int * L)(4, 0)) == _hip.EINVAL and b"bad level" in lib.sdetr_last_error()
    assert flat(L, 20, p=(ctypes.c_void_p * L)(8, None)) == _hip.EINVAL
    assert lib.sdetr_pyramid_flatten(None, L, ptrs, ptrs, ptrs, hs, ws, 8, 0, 64, 20, None, None, 8, 8, None, None,
                                     None) == 0                                            # empty batch: nothing to do
    # fp32-accurate GEMM
    assert lib.sdetr_gemm_x3_f32(None, 8, 4, 1, 8, 4, 1, 8, 4, -1, 4, 4, None, 1, None) == _hip.EINVAL
    assert b"negative" in lib.sdetr_last_error()
    assert lib.sdetr_gemm_x3_f32(None, None, 4, 1, 8, 4, 1, 8, 4, 4, 4, 4, None, 1, None) == _hip.EINVAL
    assert b"null" in lib.sdetr_last_error()
    # LDS-accumulating MSDA backward: shape support and scratch size are host functions
    assert lib.sdetr_msda_col2im_lds_supported(8, 32, 4, 4, 22323) == 1
    assert lib.sdetr_msda_col2im_lds_supported(8, 64, 4, 4, 22323) == 0      # head_dim 64
    assert lib.sdetr_msda_col2im_lds_supported(8, 32, 9, 4, 22323) == 0      # more than 8 levels
    assert lib.sdetr_msda_col2im_lds_workspace_bytes(0, 100, 8, 4) == 256
    big = lib.sdetr_msda_col2im_lds_workspace_bytes(2, 11363, 8, 4)
    assert big > 4 * 2 * 11363 * 6 and big % 256 == 0
    assert lib.sdetr_msda_col2im_lds_f32(None, None, None, None, None, None, None, 2, 100, 8, 64, 4, 10, 4, None, None,
                                         None, None, 0) == _hip.EINVAL
    assert b"D = 32" in lib.sdetr_last_error()
    # which top-k shapes go through the prefilter, how many stage-1 blocks a level takes
    assert lib.sdetr_topk_uses_prefilter(11363, 300) == 1 and lib.sdetr_topk_uses_prefilter(1050, 1050) == 0
    assert lib.sdetr_salience_head_blocks(2, 16800) == 525 and lib.sdetr_salience_head_blocks(2, 0) == 0


def test_layer_norm_train_rejects_bad_arguments(libpath):
    from salience_detr_amd import _hip
    lib = _hip.lib()
    assert lib.sdetr_layer_norm_train_supported(256) == 1 and lib.sdetr_layer_norm_train_supported(96) == 0
    f = lib.sdetr_layer_norm_train_forward_f32
    assert f(None, 8, None, 8, 8, 1e-5, 4, 96, None, 8, 8, 8) == _hip.EINVAL and b"channels" in lib.sdetr_last_error()
    assert f(None, 8, 8, 8, 8, 1e-5, 4, 256, None, 8, 8, 8) == _hip.EINVAL and b"sum_out" in lib.sdetr_last_error()
    assert f(None, None, None, 8, 8, 1e-5, 4, 256, None, 8, 8, 8) == _hip.EINVAL and b"null" in lib.sdetr_last_error()
    assert f(None, 8, None, 8, 8, 1e-5, 0, 256, None, 8, 8, 8) == 0
    g = lib.sdetr_layer_norm_train_backward_f32
    assert g(None, 8, 8, 8, 8, 8, -1, 256, 8, 8, 8) == _hip.EINVAL
    assert g(None, 8, 8, 8, 8, 8, 4, 256, 8, None, 8) == _hip.EINVAL and b"null" in lib.sdetr_last_error()


def test_sampling_prep_rejects_bad_arguments(libpath):
    from salience_detr_amd import _hip
    lib = _hip.lib()
    assert lib.sdetr_sampling_prep_supported(4, 4) == 1 and lib.sdetr_sampling_prep_supported(5, 4) == 0
    f = lib.sdetr_sampling_prep_f32
    assert f(None, 8, 8, 8, 8, 10, 8, 5, 4, 2, 8, 8) == _hip.EINVAL and b"4 levels" in lib.sdetr_last_error()
    assert f(None, 8, 8, 8, 8, 10, 8, 4, 4, 3, 8, 8) == _hip.EINVAL and b"2 or 4" in lib.sdetr_last_error()
    assert f(None, 8, None, 8, 8, 10, 8, 4, 4, 2, 8, 8) == _hip.EINVAL and b"null" in lib.sdetr_last_error()
    assert f(None, 8, 8, 8, 8, 0, 8, 4, 4, 2, 8, 8) == 0
    g = lib.sdetr_sampling_prep_backward_f32
    assert g(None, 8, 8, 8, 8, 8, -1, 8, 4, 4, 2, 8, 8) == _hip.EINVAL
    assert g(None, 8, 8, 8, 8, 8, 10, 8, 4, 4, 2, None, 8) == _hip.EINVAL and b"null" in lib.sdetr_last_error()


def test_attention_train_rejects_bad_arguments(libpath):
    from salience_detr_amd import _hip
    lib = _hip.lib()
    assert lib.sdetr_attention_train_max_rows() >= 300
    f = lib.sdetr_attention_train_forward_f32
    args = lambda N=300, hd=32, rs=512, q=8: (None, q, 300 * 512, rs, 8, 300 * 512, 512, 8, 300 * 256, 256, 2, 8, N, hd, 0.17, 8, 8)
    assert f(*args(hd=64)) == _hip.EINVAL and b"32-channel" in lib.sdetr_last_error()
    assert f(*args(N=100000)) == _hip.EINVAL and b"at most" in lib.sdetr_last_error()
    assert f(*args(rs=130)) == _hip.EINVAL and b"strides" in lib.sdetr_last_error()
    assert f(*args(q=None)) == _hip.EINVAL and b"null" in lib.sdetr_last_error()
    assert f(*args(N=0)) == 0
    g = lib.sdetr_attention_train_backward_f32
    assert g(None, 8, 1, 512, 8, 1, 512, 8, 1, 256, 2, 8, 300, 32, 0.17, 8, 8, 8, 8, None, 8) == _hip.EINVAL

