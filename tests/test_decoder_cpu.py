"""CPU: row N2 (decoder) -- the oracle's restatement and the host-side helper functions against the vectors the
imported reference produced (tests/golden/decoder_cases.npz, made by make_golden.py decoder).

Weights are the name-seeded ``synthetic.det_state_dict`` values; the fixture pins them with per-tensor CRCs.
Bars: helpers 1e-6, decoder outputs 2e-4 (fp32 accumulation order only).
"""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import salience_ref as R
from salience_detr_amd import synthetic as syn
from salience_detr_amd.salience_decoder import (MLP, SalienceTransformerDecoder, SalienceTransformerDecoderLayer,
                                                get_sine_pos_embed, inverse_sigmoid)

G = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def cases():
    return np.load(os.path.join(G, "decoder_cases.npz"))


def build_decoder(d, tag):
    """Our decoder with the fixture's name-seeded weights (checked against the stored CRCs)."""
    E, heads, d_ffn, layers, classes, B, Nq = d[f"{tag}.hyper"].tolist()
    levels = d[f"{tag}.shapes"].shape[0]
    layer = SalienceTransformerDecoderLayer(embed_dim=E, d_ffn=d_ffn, n_heads=heads, dropout=0.0, n_levels=levels,
                                            n_points=4)
    dec = SalienceTransformerDecoder(layer, layers, classes)
    sd = syn.det_state_dict(dec.state_dict(), num_heads=heads, num_levels=levels, num_points=4)
    assert sorted(sd) == d[f"{tag}.sd_keys"].tolist()          # same parameter names as the reference module
    crc = [zlib.crc32(sd[k].numpy().tobytes()) for k in sorted(sd)]
    assert crc == d[f"{tag}.sd_crc"].tolist()
    dec.load_state_dict(sd)
    return dec.eval(), sd, heads


def test_helper_functions_match_reference(cases):
    d = cases
    pos = _t(d["helper.pos"])
    assert (get_sine_pos_embed(pos, 16) - _t(d["helper.sine16"])).abs().max() < 1e-6
    assert (get_sine_pos_embed(pos[..., :2], 16, exchange_xy=False) - _t(d["helper.sine16_noswap"])).abs().max() < 1e-6
    assert (R.coordinate_sine_embed(pos, 16) - _t(d["helper.sine16"])).abs().max() < 1e-6
    x = _t(d["helper.isig_x"])
    assert torch.equal(inverse_sigmoid(x), _t(d["helper.isig_y"]))
    assert torch.equal(R.inverse_sigmoid(x), _t(d["helper.isig_y"]))
    mlp = MLP(6, 10, 3, 3)
    msd = {k[len("helper.mlp_sd."):]: _t(d[k]) for k in d.files if k.startswith("helper.mlp_sd.")}
    mlp.load_state_dict(msd)
    assert (mlp(_t(d["helper.mlp_in"])) - _t(d["helper.mlp_out"])).abs().max() < 1e-6


@pytest.mark.parametrize("tag", ["small", "e256"])
@pytest.mark.parametrize("core", ["c", "torch"])
def test_oracle_decoder_matches_reference(cases, tag, core):
    d = cases
    _, sd, heads = build_decoder(d, tag)
    layers = int(d[f"{tag}.hyper"][3])
    fn = R.msda_core_c if core == "c" else R.msda_core_torch
    cls, box = R.decoder(sd, _t(d[f"{tag}.query"]), _t(d[f"{tag}.ref"]), _t(d[f"{tag}.memory"]), _t(d[f"{tag}.shapes"]),
                         _t(d[f"{tag}.lsi"]), _t(d[f"{tag}.valid_ratios"]), _t(d[f"{tag}.mask"]), layers, heads=heads,
                         core=fn)
    assert cls.shape == d[f"{tag}.classes"].shape and box.shape == d[f"{tag}.boxes"].shape
    assert (cls - _t(d[f"{tag}.classes"])).abs().max() < 2e-4
    assert (box - _t(d[f"{tag}.boxes"])).abs().max() < 2e-5


def test_decoder_refuses_cpu_deformable_attention(cases):
    """The product decoder has no CPU path for its hot op: the MSDA module fails loudly off-device."""
    d = cases
    dec, _, _ = build_decoder(d, "small")
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            dec(_t(d["small.query"]), _t(d["small.ref"]), _t(d["small.memory"]), _t(d["small.shapes"]),
                _t(d["small.lsi"]), _t(d["small.valid_ratios"]), _t(d["small.mask"]))
