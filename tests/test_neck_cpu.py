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


def test_neck_refuses_training_mode_and_cpu_tensors():
    net = build_neck(32)
    x = dict(enumerate(torch.zeros(1, 32, h, w) for h, w in [(8, 12), (4, 6), (2, 3), (1, 2)]))
    with pytest.raises(RuntimeError, match="eval"):
        net(x)
    net.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(x)
