"""Deterministic synthetic inputs and weights for the Salience-DETR encoder hot path.

There is no dataset and no checkpoint on the benchmark box, so every test and
``bench.py`` builds its inputs from the rules below.  All tensors are produced
on the CPU from name-/seed-derived ``torch.Generator`` streams, so the same
call yields bit-identical tensors in the authoring container (where the golden
vectors are produced with the imported reference) and on the GPU box.

Shapes follow SURVEY.md section 8(d): a batch of images padded to a multiple of
32 (reference ``util/misc.py:75-104``), a 4-level pyramid at strides 8/16/32/64
(ResNet50 C3..C5 + one stride-2 3x3 conv, reference
``models/necks/channel_mapper.py``), per-level masks by nearest interpolation of
the image mask (``models/detectors/salience_detr.py:172-176``).
"""
import math
import zlib
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F


def _gen(name: str, salt: int = 0) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def det_randn(name: str, shape, salt: int = 0) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=_gen(name, salt), dtype=torch.float32)


def det_rand(name: str, shape, salt: int = 0) -> torch.Tensor:
    return torch.rand(tuple(shape), generator=_gen(name, salt), dtype=torch.float32)


def _ring_bias(num_heads: int, num_levels: int, num_points: int) -> torch.Tensor:
    """The 8-direction ring initialisation of ``sampling_offsets.bias``
    (reference ``models/bricks/ms_deform_attn.py:266-276``), restated."""
    theta = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    ring = torch.stack([theta.cos(), theta.sin()], -1)
    ring = ring / ring.abs().max(-1, keepdim=True)[0]
    ring = ring.view(num_heads, 1, 1, 2).repeat(1, num_levels, num_points, 1)
    scale = torch.arange(1, num_points + 1, dtype=torch.float32).view(1, 1, num_points, 1)
    return (ring * scale).reshape(-1)


def det_state_dict(reference: Dict[str, torch.Tensor], num_heads: int = 8, num_levels: int = 4,
                   num_points: int = 4, salt: int = 0) -> Dict[str, torch.Tensor]:
    """"Trained-like" deterministic weights keyed by state_dict name.

    ``reference`` only supplies names, shapes and dtypes (any module's
    ``state_dict()``); values are replaced by the name-seeded rules below so
    that two different implementations with the same key names receive
    bit-identical parameters.  Non-float entries and the two filter-ratio
    buffers are passed through unchanged.
    """
    out = {}
    for key, ref in reference.items():
        # ``encoder.enhance_mcsp`` IS ``encoder_class_head`` (shared storage, reference
        # salience_transformer.py:79): both keys must carry the same values.
        name = key.replace("encoder.enhance_mcsp.", "encoder_class_head.")
        shape = tuple(ref.shape)
        if (not ref.is_floating_point()) or name.endswith("filter_ratio"):
            out[key] = ref.clone()
            continue
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("sampling_offsets.bias"):
            t = _ring_bias(num_heads, num_levels, num_points) + 0.25 * det_randn(name, shape, salt)
        elif name.endswith("sampling_offsets.weight"):
            t = 0.05 * det_randn(name, shape, salt)
        elif name.endswith("attention_weights.weight"):
            t = 0.08 * det_randn(name, shape, salt)
        elif name.endswith("attention_weights.bias"):
            t = 0.3 * det_randn(name, shape, salt)
        elif name.endswith("alpha"):
            t = 0.2 * det_randn(name, shape, salt)
        elif name.endswith("level_embeds"):
            t = det_randn(name, shape, salt)
        elif name.endswith("_embed.weight"):  # learned row/col background embeddings
            t = det_rand(name, shape, salt)
        elif leaf == "running_var":  # BatchNorm statistics of the neck: positive, away from zero
            t = 0.5 + det_rand(name, shape, salt)
        elif leaf == "running_mean":
            t = 0.1 * det_randn(name, shape, salt)
        elif len(shape) == 4:  # convolution kernels [out, in / groups, kh, kw]
            t = det_randn(name, shape, salt) * (1.0 / math.sqrt(shape[1] * shape[2] * shape[3]))
        elif len(shape) == 2:  # Linear / in_proj weights
            t = det_randn(name, shape, salt) * (1.0 / math.sqrt(shape[1]))
        elif len(shape) == 1 and leaf == "weight":  # LayerNorm gains
            t = 1.0 + 0.1 * det_randn(name, shape, salt)
        elif len(shape) == 1:  # biases
            t = 0.05 * det_randn(name, shape, salt)
        else:
            t = det_randn(name, shape, salt)
        out[key] = t.to(ref.dtype)
    return out


def pad_to_32(h: int, w: int) -> Tuple[int, int]:
    return (h + 31) // 32 * 32, (w + 31) // 32 * 32


def pyramid_shapes(h_pad: int, w_pad: int, num_levels: int = 4) -> List[Tuple[int, int]]:
    """Level shapes of the reference's ResNet50 + ChannelMapper pyramid for a padded canvas."""
    shapes = []
    h, w = h_pad // 8, w_pad // 8
    for lvl in range(num_levels):
        shapes.append((h, w))
        if lvl < 2:
            h, w = (h + 1) // 2, (w + 1) // 2  # stride-2 stages of the backbone
        else:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1  # 3x3 stride-2 pad-1 conv
    return shapes


def make_masks(image_sizes: Sequence[Tuple[int, int]],
               level_shapes: Sequence[Tuple[int, int]] = None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """Image padding mask (True on padding) and its per-level nearest down-samples."""
    h_pad, w_pad = pad_to_32(max(s[0] for s in image_sizes), max(s[1] for s in image_sizes))
    mask = torch.ones(len(image_sizes), h_pad, w_pad, dtype=torch.bool)
    for i, (h, w) in enumerate(image_sizes):
        mask[i, :h, :w] = False
    if level_shapes is None:
        level_shapes = pyramid_shapes(h_pad, w_pad)
    level_masks = [F.interpolate(mask[None].float(), size=tuple(s))[0].to(torch.bool) for s in level_shapes]
    return mask, level_masks


def make_feats(batch: int, level_shapes: Sequence[Tuple[int, int]], embed_dim: int = 256,
               seed: int = 0) -> List[torch.Tensor]:
    return [det_randn(f"feat.l{l}", (batch, embed_dim, h, w), salt=seed)
            for l, (h, w) in enumerate(level_shapes)]


def make_msda_inputs(B: int, Nq: int, level_shapes: Sequence[Tuple[int, int]], M: int = 8, D: int = 32,
                     P: int = 4, seed: int = 0, spread_px: float = 4.0, dtype=torch.float32):
    """Op-level micro-benchmark inputs: value ~ N(0,1); each query sits on the
    centre of a uniformly drawn token of a random level and samples U(-spread, spread) px around
    it on every level; weights = softmax(N(0,1)) over the L*P samples."""
    L = len(level_shapes)
    shapes = torch.tensor(level_shapes, dtype=torch.int64)
    sizes = shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    Nv = int(sizes.sum())
    g = _gen("msda_inputs", seed)
    value = torch.randn(B, Nv, M, D, generator=g, dtype=torch.float32)
    tok = torch.randint(0, Nv, (B, Nq), generator=g)
    lvl = (tok[..., None] >= lsi[None, None]).sum(-1) - 1
    rel = tok - lsi[lvl]
    Wl = shapes[lvl, 1]
    Hl = shapes[lvl, 0]
    cx = ((rel % Wl).float() + 0.5) / Wl.float()
    cy = (torch.div(rel, Wl, rounding_mode="floor").float() + 0.5) / Hl.float()
    centre = torch.stack([cx, cy], -1)  # [B, Nq, 2]
    off = (torch.rand(B, Nq, M, L, P, 2, generator=g) * 2 - 1) * spread_px
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()  # [L, 2] (W, H)
    loc = centre[:, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    aw = torch.randn(B, Nq, M, L * P, generator=g).softmax(-1).view(B, Nq, M, L, P)
    return (value.to(dtype), shapes, lsi, loc.to(dtype).contiguous(), aw.to(dtype).contiguous())


def make_encoder_like_queries(B: int, Nq: int, level_shapes: Sequence[Tuple[int, int]], M: int = 8, P: int = 4,
                              seed: int = 0, offset_px: float = 2.0, dtype=torch.float32):
    """Fused-MSDA inputs shaped like an encoder layer's: ``Nq`` distinct tokens of the pyramid per image,
    reference points = that token's centre on every level (``[B,Nq,L,2]``), projection rows
    ``[M*L*P*2 offsets ~ N(0, offset_px) | M*L*P logits ~ N(0,1)]``."""
    L = len(level_shapes)
    shapes = torch.tensor(level_shapes, dtype=torch.int64)
    sizes = shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    Nv = int(sizes.sum())
    g = _gen("encoder_like", seed)
    tok = torch.stack([torch.randperm(Nv, generator=g)[:Nq] for _ in range(B)])
    lvl = (tok[..., None] >= lsi[None, None]).sum(-1) - 1
    rel = tok - lsi[lvl]
    cx = ((rel % shapes[lvl, 1]).float() + 0.5) / shapes[lvl, 1].float()
    cy = (torch.div(rel, shapes[lvl, 1], rounding_mode="floor").float() + 0.5) / shapes[lvl, 0].float()
    ref = torch.stack([cx, cy], -1)[:, :, None, :].expand(B, Nq, L, 2).contiguous()
    proj = torch.cat([torch.randn(B, Nq, M * L * P * 2, generator=g) * offset_px,
                      torch.randn(B, Nq, M * L * P, generator=g)], -1)
    return tok, ref, proj.to(dtype), shapes, lsi


def sine_position_embedding(mask: torch.Tensor, num_pos_feats: int = 128, temperature: float = 10000.0,
                            eps: float = 1e-6, offset: float = -0.5) -> torch.Tensor:
    """Caller-side input generation: the normalised 2-d sine position map the reference's detector feeds the
    transformer (semantics of ``models/bricks/position_encoding.py:33-67`` with ``normalize=True``, pinned by the
    ``pos{l}`` arrays of ``tests/golden/hotpath_small_*.npz``).  ``mask`` ``[B,H,W]`` bool, True on padding ->
    ``[B, 2*num_pos_feats, H, W]`` fp32, channels = (y features, x features).

    Written from the definition: along each axis the coordinate of a pixel is its 1-based rank among the valid
    pixels of its column / row (+ offset), divided by that line's valid count and scaled to 2*pi; feature 2i / 2i+1
    are sin / cos of coordinate / temperature^(2i / num_pos_feats).
    """
    valid = (~mask).to(torch.float32)
    two_pi = 2.0 * math.pi
    # (the frequency table is computed on the host: a device pow() differs from the CPU's in the last bits, which the
    # phases of the high-frequency features amplify to ~1e-3)
    freq = (temperature ** (2.0 * torch.div(torch.arange(num_pos_feats, dtype=torch.float32), 2,
                                            rounding_mode="floor") / num_pos_feats)).to(mask.device)

    def axis_features(dim: int) -> torch.Tensor:
        rank = valid.cumsum(dim)
        total = rank.narrow(dim, rank.shape[dim] - 1, 1)
        coord = (rank + offset) / (total + eps) * two_pi                 # [B,H,W]
        phase = coord.unsqueeze(-1) / freq                               # [B,H,W,F]
        even, odd = phase[..., 0::2].sin(), phase[..., 1::2].cos()
        return torch.stack((even, odd), dim=-1).flatten(-2)              # sin, cos interleaved

    return torch.cat((axis_features(1), axis_features(2)), dim=-1).permute(0, 3, 1, 2).contiguous()


def train_loss_weights(shapes):
    """Fixed weights of the synthetic whole-transformer training loss ``sum_i (output_i * w_i).sum()``, one tensor per
    output (name-seeded: the golden generator and the tests build the same ones)."""
    return [det_randn(f"train.loss.w{i}", tuple(sh)) for i, sh in enumerate(shapes)]


def denoising_inputs(B: int, E: int, proposals: int, n_dn: int = 6):
    """Denoising queries as the detector hands them to the transformer (reference models/detectors/salience_detr.py:
    the GenerateCDNQueries outputs): label embeddings ``[B,n_dn,E]``, inverse-sigmoid boxes ``[B,n_dn,4]`` and the
    ``[n_dn+proposals]^2`` attention mask (True = may not attend) that hides the two groups from each other and from the
    matching queries."""
    label_q = det_randn("train.dn.label", (B, n_dn, E)) * 0.5
    box_q = det_randn("train.dn.box", (B, n_dn, 4))
    n = n_dn + proposals
    mask = torch.zeros(n, n, dtype=torch.bool)
    mask[n_dn:, :n_dn] = True
    half = n_dn // 2
    mask[:half, half:n_dn] = True
    mask[half:n_dn, :half] = True
    return label_q, box_q, mask

