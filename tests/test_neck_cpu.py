"""Row N3 on the CPU: the oracle's restatement of the RepVGGPluX neck against the vectors captured from the imported
reference (tests/golden/make_golden.py neck)."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import synthetic as syn

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
    if not sd:  # name-seeded weights: regenerate them from the key names / shapes of the 32-channel set
        small = {k[len("sd32."):]: d[k] for k in d.files if k.startswith("sd32.")}
        scale = C // 32
        shapes = {}
        for k, v in small.items():
            shp = list(v.shape)
            if k.endswith("se_module.se_module.0.weight"):
                shp[0], shp[1] = shp[0] * scale, shp[1] * scale
            elif k.endswith("conv_mask.weight"):
                shp[1] *= scale
            elif k.endswith("conv_mask.bias") or k.endswith("num_batches_tracked"):
                pass
            else:
                shp = [x * scale if i < 2 else x for i, x in enumerate(shp)]
            shapes[k] = torch.zeros(shp, dtype=_t(v).dtype)
        sd = syn.det_state_dict(shapes)
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
