"""Salience-DETR transformer decoder on MI355X (caller row D1).

Same classes, constructor arguments, parameter names and ``forward`` signatures as the reference
(``models/bricks/salience_transformer.py:500-674``, ``models/bricks/basic.py:6-26``,
``models/bricks/position_encoding.py:105-132``) so released checkpoints load and ``SalienceTransformer`` can hold
this decoder unchanged.  The no-grad path is arranged for the hardware like the encoder's:

* the layers' cross-attentions all sample the same, never-updated ``memory``, so their ``value_proj`` run as ONE
  token-resident projection straight into head-major 16-bit value maps (``batched_value_maps``);
* the cross-attention is the fused MSDA kernel with 4-d reference boxes (softmax, box-relative sampling locations and
  bilinear gather in one launch), its query projection has the position add in its prologue;
* residual adds + LayerNorms are single launches; the feed-forward block uses the one-launch MFMA kernel when the
  query count is large enough, library GEMMs otherwise (900 queries per image are not).

With autograd enabled every layer runs plain differentiable torch ops around the HIP forward/backward op.
"""
import copy
import math
from typing import Optional

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from .filter_ops import (attention_heads, attention_heads_applies, box_refine, decoder_head, decoder_head_applies,
                         decoder_query_sine_embed, fused_ffn,
                         fused_ffn_applies, fused_layer_norm, mlp_rows, mlp_rows_applies, ref_point_head,
                         ref_point_head_applies, rows_linear, rows_linear_applies, rows_linear_ln,
                         rows_linear_ln_applies)
from .layer_norm_train import add_layer_norm
from .ms_deform_attn import MultiScaleDeformableAttention, batched_value_maps


def inverse_sigmoid(x: Tensor, eps: float = 1e-3) -> Tensor:
    """``log(x / (1 - x))`` with both terms clamped at ``eps`` (reference ``util/misc.py:31-35``)."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def get_sine_pos_embed(pos_tensor: Tensor, num_pos_feats: int = 128, temperature: int = 10000,
                       scale: float = 2 * math.pi, exchange_xy: bool = True) -> Tensor:
    """Sine embedding of every coordinate of ``pos_tensor`` [..., n] -> [..., n * num_pos_feats]; with
    ``exchange_xy`` the first two coordinate blocks are swapped (position_encoding.py:105-132)."""
    idx = torch.arange(num_pos_feats, dtype=torch.float32, device=pos_tensor.device)
    dim_t = temperature ** (2 * torch.div(idx, 2, rounding_mode="floor") / num_pos_feats)
    ang = pos_tensor.unsqueeze(-1) * scale / dim_t                                   # [..., n, F]
    emb = torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), dim=-1).flatten(-2)
    if exchange_xy:
        emb = torch.cat((emb[..., 1:2, :], emb[..., 0:1, :], emb[..., 2:, :]), dim=-2)
    return emb.reshape(*pos_tensor.shape[:-1], -1)


class MLP(nn.Module):
    """``num_layers`` Linear layers with ReLU between them (basic.py:6-26); parameters ``layers.{i}``."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        for layer in self.layers:
            nn.init.xavier_uniform_(layer.weight)
            nn.init.constant_(layer.bias, 0.0)

    def forward(self, x, x_second=None):
        """``x_second`` (no-grad HIP path only): a second set of rows through the same layers, result stacked in front --
        ``forward(stack((x, x_second)))`` without the stacked copy."""
        if mlp_rows_applies(x, self.layers):
            return mlp_rows(x, self.layers, x_second)          # the whole chain in one launch (csrc/mlp_rows.hip)
        if x_second is not None:
            x = torch.stack((x, x_second))
        fused = x.is_cuda and not torch.is_grad_enabled()
        for i, layer in enumerate(self.layers):
            if i + 1 < self.num_layers:
                if fused:   # ReLU in the GEMM's epilogue (hipBLASLt) instead of a separate elementwise launch
                    try:
                        x = torch._addmm_activation(layer.bias, x.reshape(-1, x.shape[-1]), layer.weight.t(),
                                                    use_gelu=False).view(*x.shape[:-1], layer.out_features)
                        continue
                    except (RuntimeError, AttributeError):
                        fused = False
                x = F.relu(layer(x))
            else:
                x = layer(x)
        return x


def _needs_grad(module: nn.Module, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


class SalienceTransformerDecoderLayer(nn.Module):
    """Self-attention over the queries, deformable cross-attention into ``memory``, feed-forward
    (salience_transformer.py:500-588); note the module order norm2 / norm1 / norm3 of the reference."""

    def __init__(self, embed_dim=256, d_ffn=1024, n_heads=8, dropout=0.1, activation=nn.ReLU(inplace=True), n_levels=4,
                 n_points=4):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = n_heads
        self.cross_attn = MultiScaleDeformableAttention(embed_dim, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.self_attn = nn.MultiheadAttention(embed_dim, n_heads, dropout=dropout, batch_first=True)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.linear1 = nn.Linear(embed_dim, d_ffn)
        self.activation = activation
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, embed_dim)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(embed_dim)
        self.init_weights()

    def init_weights(self):
        nn.init.xavier_uniform_(self.self_attn.in_proj_weight)
        nn.init.xavier_uniform_(self.self_attn.out_proj.weight)
        nn.init.xavier_uniform_(self.linear1.weight)
        nn.init.xavier_uniform_(self.linear2.weight)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        tgt2 = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        return add_layer_norm(tgt, self.norm3, self.dropout4(tgt2))

    def _self_attention(self, qk: Tensor, v: Tensor, attn_mask: Optional[Tensor]) -> Tensor:
        return self.self_attn(query=qk, key=qk, value=v, attn_mask=attn_mask, need_weights=False)[0]

    def _self_attention_native(self, query: Tensor, query_pos: Tensor, attn_mask: Optional[Tensor], project: bool = True):
        """nn.MultiheadAttention(q = k = query + pos, v = query) on its parameters without the module's layout copies:
        two projections (q|k from query + pos, v from query), the flash kernel on strided head views, out_proj.
        ``project=False``: the concatenated heads BEFORE ``out_proj`` when the own attention kernel ran (the caller fuses
        ``out_proj`` with the residual and the norm).  Returns ``(tensor, out_proj applied)``."""
        mha = self.self_attn
        B, n, E = query.shape
        H = mha.num_heads
        w, b = mha.in_proj_weight, mha.in_proj_bias
        if rows_linear_applies(query, w, b) and query_pos.shape == query.shape:
            # q | k | v in one launch: the position add lives in the kernel (csrc/mlp_rows.hip, round 5).  (The
            # token-resident projection kernel of the encoder was tried first: flat 18 us at 2 x 900 rows.)
            qkv = rows_linear(query, w, b, pos=query_pos, pos_features=2 * E)
            qk2, v2 = qkv[..., :2 * E], qkv[..., 2 * E:]
        else:
            qk2 = F.linear(query + query_pos, w[:2 * E], b[:2 * E])
            v2 = F.linear(query, w[2 * E:], b[2 * E:])
        if attn_mask is None and attention_heads_applies(qk2[..., :E], qk2[..., E:], v2, H):
            # own flash kernel on the strided projection slices; the heads come out concatenated
            heads = attention_heads(qk2[..., :E], qk2[..., E:], v2, H)
            if not project:
                return heads, False
            return F.linear(heads, mha.out_proj.weight, mha.out_proj.bias), True
        qk = qk2.view(B, n, 2, H, E // H)
        v = v2.view(B, n, H, E // H)
        mask = attn_mask
        if mask is not None and mask.dtype == torch.bool:
            mask = torch.zeros_like(mask, dtype=query.dtype).masked_fill_(mask, float("-inf"))   # True = not allowed
        o = F.scaled_dot_product_attention(qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2), v.transpose(1, 2),
                                           attn_mask=mask)
        return F.linear(o.transpose(1, 2).reshape(B, n, E), mha.out_proj.weight, mha.out_proj.bias), True

    def forward(self, query, query_pos, reference_points, value, spatial_shapes, level_start_index,
                self_attn_mask=None, key_padding_mask=None, value_hm=None):
        """Reference signature (salience_transformer.py:553-563) plus the optional pre-projected head-major
        ``value_hm`` [B,M,Nv,D] the decoder's batched value projection supplies on the no-grad path."""
        native = query.is_cuda and not _needs_grad(self, query, value, reference_points)
        if native and query_pos is not None:
            fuse_tail = rows_linear_ln_applies(query, self.self_attn.out_proj, self.norm2)
            query2, projected = self._self_attention_native(query, query_pos, self_attn_mask, project=not fuse_tail)
            if not projected:   # out_proj + residual + norm2 in one launch (csrc/mlp_rows.hip, round 5)
                query = rows_linear_ln(query2, self.self_attn.out_proj, self.norm2, residual=query)
                query2 = None
        else:
            query2 = self._self_attention(self.with_pos_embed(query, query_pos), query, self_attn_mask)
        if not native:
            query = add_layer_norm(query, self.norm2, self.dropout2(query2))
            query2 = self.cross_attn(query=self.with_pos_embed(query, query_pos), reference_points=reference_points,
                                     value=value, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                     key_padding_mask=key_padding_mask)
            query = add_layer_norm(query, self.norm1, self.dropout1(query2))
            return self.forward_ffn(query)
        if query2 is not None:
            query = fused_layer_norm(query, self.norm2, residual=query2)
        if value_hm is None:
            value_hm = self.cross_attn.project_value(value, key_padding_mask)
        if rows_linear_ln_applies(query, self.cross_attn.output_proj, self.norm1):
            sampled = self.cross_attn.forward_native(query, reference_points.contiguous(), value_hm, spatial_shapes,
                                                     level_start_index, query_pos=query_pos, apply_output_proj=False)
            query = rows_linear_ln(sampled, self.cross_attn.output_proj, self.norm1, residual=query)
        else:
            query2 = self.cross_attn.forward_native(query, reference_points.contiguous(), value_hm, spatial_shapes,
                                                    level_start_index, query_pos=query_pos)
            query = fused_layer_norm(query, self.norm1, residual=query2)
        if fused_ffn_applies(query, self.linear1, self.linear2, self.norm3, self.activation):
            return fused_ffn(query, self.linear1, self.linear2, self.norm3)
        hidden = self.activation(self.linear1(query))
        return fused_layer_norm(query, self.norm3, residual=self.linear2(hidden))


class SalienceTransformerDecoder(nn.Module):
    """Iterative box-refining decoder (salience_transformer.py:591-674): returns the per-layer class logits
    ``[num_layers,B,Nq,num_classes]`` and boxes ``[num_layers,B,Nq,4]`` (cx, cy, w, h in [0,1])."""

    def __init__(self, decoder_layer, num_layers, num_classes):
        super().__init__()
        self.embed_dim = decoder_layer.embed_dim
        self.num_layers = num_layers
        self.num_classes = num_classes
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.ref_point_head = MLP(2 * self.embed_dim, self.embed_dim, self.embed_dim, 2)
        self.class_head = nn.ModuleList([nn.Linear(self.embed_dim, num_classes) for _ in range(num_layers)])
        self.bbox_head = nn.ModuleList([MLP(self.embed_dim, self.embed_dim, 4, 3) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(self.embed_dim)
        self.init_weights()

    def init_weights(self):
        for layer in self.layers:
            layer.init_weights()
        prior = -math.log((1 - 0.01) / 0.01)
        for head in self.class_head:
            nn.init.constant_(head.bias, prior)
        for head in self.bbox_head:
            nn.init.constant_(head.layers[-1].weight, 0.0)
            nn.init.constant_(head.layers[-1].bias, 0.0)

    def project_values(self, value, key_padding_mask=None):
        """The layers' cross-attention value maps ``[num_layers, B, M, Nv, D]`` (one projection: they all sample the same,
        never-updated memory).  ``forward(..., value_maps=...)`` takes them, so a caller may place the projection where
        it likes (beside the proposal stage on a second stream it measured slower under graph replay:
        benchmarks/experiments/README.md)."""
        return batched_value_maps([l.cross_attn for l in self.layers], value, key_padding_mask)

    def forward(self, query, reference_points, value, spatial_shapes, level_start_index, valid_ratios,
                key_padding_mask=None, attn_mask=None, value_maps=None):
        if query.is_cuda and not _needs_grad(self, query, value, reference_points):
            return self._forward_native(query, reference_points, value, spatial_shapes, level_start_index, valid_ratios,
                                        key_padding_mask, attn_mask, value_maps)
        ratio = torch.cat([valid_ratios, valid_ratios], -1)[:, None]          # [B,1,L,4]
        half = self.embed_dim // 2
        classes, coords = [], []
        for i, layer in enumerate(self.layers):
            ref_in = reference_points.detach()[:, :, None] * ratio            # [B,Nq,L,4]
            query_pos = self.ref_point_head(get_sine_pos_embed(ref_in[:, :, 0, :], num_pos_feats=half).to(query.dtype))
            query = layer(query=query, query_pos=query_pos, reference_points=ref_in, value=value,
                          spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                          key_padding_mask=key_padding_mask, self_attn_mask=attn_mask)
            normed = self.norm(query)
            classes.append(self.class_head[i](normed))
            coords.append((self.bbox_head[i](normed).float() + inverse_sigmoid(reference_points.float())).sigmoid())
            if i + 1 == self.num_layers:
                break
            # the refinement is NOT detached for the next layer's look-forward-twice gradient (:666-668)
            reference_points = (self.bbox_head[i](query).float() + inverse_sigmoid(reference_points.detach().float())).sigmoid()
        return torch.stack(classes), torch.stack(coords)

    def _forward_native(self, query, reference_points, value, spatial_shapes, level_start_index, valid_ratios,
                        key_padding_mask, attn_mask, value_maps=None):
        """No-grad path: one value projection for all layers (they sample the same memory), one launch each for the
        reference scaling + sine embedding and for the two box refinements of a layer (output boxes from the normed
        query, next reference from the raw query: one stacked ``bbox_head`` pass)."""
        if value_maps is None:
            value_maps = self.project_values(value, key_padding_mask)
        ref = reference_points.float()
        classes, coords = [], []
        for i, layer in enumerate(self.layers):
            if ref_point_head_applies(self.ref_point_head.layers, query.dtype, self.embed_dim // 2):
                ref_in, query_pos = ref_point_head(ref, valid_ratios, self.ref_point_head.layers, query.dtype)   # one launch
            else:
                ref_in, sine = decoder_query_sine_embed(ref, valid_ratios, self.embed_dim // 2, query.dtype)
                query_pos = self.ref_point_head(sine)
            query = layer(query=query, query_pos=query_pos, reference_points=ref_in, value=value,
                          spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                          key_padding_mask=key_padding_mask, self_attn_mask=attn_mask, value_hm=value_maps[i])
            last = i + 1 == self.num_layers
            if decoder_head_applies(query, self.norm, self.class_head[i], self.bbox_head[i].layers):
                # norm + class head + both bbox chains + refinement: one launch (csrc/mlp_rows.hip, round 5)
                logits, boxes = decoder_head(query, self.norm, self.class_head[i], self.bbox_head[i].layers, ref, not last)
                classes.append(logits)
                coords.append(boxes[0])
                if last:
                    break
                ref = boxes[1]
                continue
            normed = fused_layer_norm(query, self.norm)
            classes.append(self.class_head[i](normed))
            if i + 1 == self.num_layers:
                coords.append(box_refine(self.bbox_head[i](normed), ref))
                break
            both = box_refine(self.bbox_head[i](normed, query), ref)                      # [2,B,Nq,4]
            coords.append(both[0])
            ref = both[1]
        return torch.stack(classes), torch.stack(coords)
