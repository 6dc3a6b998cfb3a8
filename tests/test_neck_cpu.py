"""Row N3 on the CPU: the oracle's restatement of the RepVGGPluX neck against the vectors captured from the imported
reference (tests/golden/make_golden.py neck)."""
import os
import zlib

import numpy as np
import pytest
import torch

import neck_emulation as EMU
from oracle import salience_ref as R
from salience_detr_amd import filter_ops as FO
from salience_detr_amd import synthetic as syn
from salience_detr_amd.salience_neck import build_neck

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["small", "ragged", "wide"]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def neck_case(tag):
    """(state dict, NCHW inputs, NCHW reference outputs) of one fixture case."""
    d = np.load(os.path.join(G, "neck_cases.npz"))
    C = int(d[f"{tag}.channels"])
    prefix = f"sd{C}."
    sd = {k[len(prefix):]: _t(d[k]) for k in d.files if k.startswith(prefix)}
    if not sd:  # name-seeded weights: regenerate them from the product module's key names / shapes
        sd = syn.det_state_dict(build_neck(C).state_dict())
        crc = zlib.crc32(b"".join(sd[k].numpy().tobytes() for k in sorted(sd)))
        assert crc == int(d[f"{tag}.sd_crc"]), "name-seeded weights differ from the ones the fixture was made with"
    feats = [_t(d[f"{tag}.feat{l}"]) for l in range(4)]
    outs = [_t(d[f"{tag}.out{l}"]) for l in range(4)]
    return sd, feats, outs


@pytest.mark.parametrize("tag", CASES)
def test_oracle_neck_matches_reference(tag):
    sd, feats, outs = neck_case(tag)
    got = R.neck(sd, feats, groups=4)
    for l in range(4):
        assert got[l].shape == outs[l].shape
        assert (got[l] - outs[l]).abs().max() < 2e-5, (tag, l)


def test_oracle_neck_on_memory_is_the_token_major_form():
    sd, feats, outs = neck_case("ragged")
    shapes = torch.tensor([f.shape[-2:] for f in feats])
    memory = torch.cat([f.flatten(2).transpose(1, 2) for f in feats], 1)
    got = R.neck_on_memory(sd, memory, shapes)
    ref = torch.cat([o.flatten(2).transpose(1, 2) for o in outs], 1)
    assert (got - ref).abs().max() < 2e-5


def test_neck_state_dict_keys_are_the_references():
    d = np.load(os.path.join(G, "neck_cases.npz"))
    ref = {k[len("sd32."):]: tuple(d[k].shape) for k in d.files if k.startswith("sd32.")}
    mine = {k: tuple(v.shape) for k, v in build_neck(32).state_dict().items()}
    assert mine == ref


@pytest.mark.parametrize("tag", CASES)
def test_neck_host_logic_against_reference(tag, monkeypatch):
    """salience_neck.py with the three kernels replaced by torch restatements of their ABI contracts: the folding of
    the norms and of the 3x3 + 1x1 pair, the weight layouts and the split / commuted 1x1 convolutions reproduce the
    reference's outputs."""
    monkeypatch.setattr(FO, "neck_conv3x3", EMU.conv3x3)
    monkeypatch.setattr(FO, "neck_combine", EMU.combine)
    monkeypatch.setattr(FO, "neck_gate_shortcut", EMU.gate_shortcut)
    sd, feats, outs = neck_case(tag)
    net = build_neck(int(feats[0].shape[1]))
    net.load_state_dict(sd)
    net.eval()
    with torch.no_grad():
        got = list(net(dict(enumerate(feats))).values())
        shapes = [tuple(f.shape[-2:]) for f in feats]
        mem = net.forward_memory(torch.cat([f.flatten(2).transpose(1, 2) for f in feats], 1), shapes)
    for l in range(4):
        assert (got[l] - outs[l]).abs().max() < 5e-5, (tag, l)
    assert (mem - torch.cat([o.flatten(2).transpose(1, 2) for o in outs], 1)).abs().max() < 5e-5


def test_neck_refuses_cpu_tensors_in_both_modes():
    net = build_neck(32)
    x = dict(enumerate(torch.zeros(1, 32, h, w) for h, w in [(8, 12), (4, 6), (2, 3), (1, 2)]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # training form: device tensors only, too
        net(x)
    net.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(x)


def _patched(monkeypatch):
    monkeypatch.setattr(FO, "neck_conv3x3", EMU.conv3x3)
    monkeypatch.setattr(FO, "neck_combine", EMU.combine)
    monkeypatch.setattr(FO, "neck_gate_shortcut", EMU.gate_shortcut)


def test_neck_folded_weights_follow_parameter_updates(monkeypatch):
    """The folded / packed weights are cached on the module; an in-place update of any parameter or BatchNorm buffer
    (optimizer step, load_state_dict) must invalidate them."""
    _patched(monkeypatch)
    sd, feats, _ = neck_case("small")
    net = build_neck(32)
    net.load_state_dict(sd)
    net.eval()
    x = dict(enumerate(feats))
    with torch.no_grad():
        first = [o.clone() for o in net(x).values()]
        again = list(net(x).values())
        assert all(torch.equal(a, b) for a, b in zip(first, again))
        sd2 = dict(sd)
        for key in ("pan_blocks.1.bottlenecks.2.conv2.1.running_var", "lateral_convs.0.0.weight",
                    "layer_blocks.0.bottlenecks.0.se_module.conv_mask.weight"):
            sd2[key] = sd[key] * 1.5
        net.load_state_dict(sd2)
        got = list(net(x).values())
    want = R.neck(sd2, feats, groups=4)
    assert any((a - b).abs().max() > 1e-3 for a, b in zip(first, got))
    for l in range(4):
        assert (got[l] - want[l]).abs().max() < 5e-5


def test_neck_extra_block_and_constructor_checks(monkeypatch):
    from torch import nn
    from salience_detr_amd.salience_neck import RepVGGPluXNetwork
    _patched(monkeypatch)
    sd, feats, outs = neck_case("ragged")
    net = RepVGGPluXNetwork([32] * 4, [32] * 4, groups=4, extra_block=True)
    net.load_state_dict(sd)
    net.eval()
    with torch.no_grad():
        out = net({"a": feats[0], "b": feats[1], "c": feats[2], "d": feats[3]})
    assert list(out.keys()) == ["a", "b", "c", "d", "pool"]
    # F.max_pool2d(x, kernel 1, stride 2, padding 0) of the coarsest output (models/necks/repnet.py:242-243)
    assert torch.equal(out["pool"], torch.nn.functional.max_pool2d(out["d"], 1, 2, 0))
    assert (out["d"] - outs[3]).abs().max() < 5e-5
    with pytest.raises(ValueError):
        RepVGGPluXNetwork([32, 64, 64, 64], [32, 64, 64, 64])          # one channel count on all levels
    with pytest.raises(ValueError):
        RepVGGPluXNetwork([32] * 4, [32] * 4, activation=nn.ReLU)      # the kernels implement SiLU
    with pytest.raises(ValueError):
        RepVGGPluXNetwork([0, 32, 32, 32], [32] * 4)                   # as the reference (repnet.py:147-149)


def test_oracle_neck_training_mode_matches_reference():
    """Batch-statistics form (what the reference trains with; SyncBatchNorm over all ranks' pixels equals this on one
    process): outputs, updated running statistics and gradients of a probe loss against the imported reference.  The
    HIP neck is eval-only this round; this pins the checker the training form will be held to."""
    d = np.load(os.path.join(G, "neck_cases.npz"))
    sd, feats, _ = neck_case("small")
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    feats = [f.clone().requires_grad_(True) for f in feats]
    stats = {}
    outs = R.neck(sd, feats, groups=4, new_stats=stats)
    probes = [syn.det_randn(f"neck.train.probe{l}", o.shape) for l, o in enumerate(outs)]
    loss = sum((o * p).sum() for o, p in zip(outs, probes))
    loss.backward()
    assert abs(float(loss.detach()) - float(d["train.loss"])) < 2e-3
    for l in range(4):
        assert (outs[l].detach() - _t(d[f"train.out{l}"])).abs().max() < 5e-5, l
        assert (feats[l].grad - _t(d[f"train.grad_feat{l}"])).abs().max() < 2e-4, l
    stat_keys = [k[len("train.stat."):] for k in d.files if k.startswith("train.stat.")]
    grad_keys = [k[len("train.grad."):] for k in d.files if k.startswith("train.grad.")]
    assert len(stat_keys) == 5 and len(grad_keys) == 4
    for k in stat_keys:
        assert (stats[k] - _t(d["train.stat." + k])).abs().max() < 1e-5, k
    for k in grad_keys:
        ref = _t(d["train.grad." + k])
        assert (sd[k].grad - ref).abs().max() < 2e-4 * max(1.0, float(ref.abs().max())), k
    # every norm of the neck reported new statistics: 3 lateral + 3 down + 6 CSP layers x (2 + 3 blocks x 2)
    assert len(stats) == 2 * (3 + 3 + 6 * 8)
