"""Salience-DETR transformer encoder on MI355X (boundary B5).

Same classes, constructor arguments, parameter names and ``forward`` signatures as the reference
(``models/bricks/salience_transformer.py:298-497``) so ``SalienceTransformer`` can hold this encoder
unchanged; inside, the no-grad path is re-organised for the hardware:

* the six layers sample the SAME, never-updated feature map (``value = output = query``, :452), so all
  six ``value_proj`` GEMMs run as ONE [Nv,256]x[256,1536] GEMM and each layer's slice is re-laid
  head-major (optionally bf16) by ``value_to_head_major``;
* query / position / reference-point rows move with 16-byte-lane gather/scatter kernels (no expanded
  int64 index tensors), the per-image ``focus_token_nums`` prefix is applied on the device (no host
  sync, no python loop over images);
* the per-layer top-300 selection uses the single-workgroup top-k kernel;
* softmax + sampling locations + bilinear gather are one launch (``msda_fused_forward``).

With autograd enabled the layers fall back to differentiable torch indexing around the HIP
forward/backward op (``MultiScaleDeformableAttnFunction``).
"""
import copy
import math
from typing import List, Optional

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import attention_train, pyramid
from .filter_ops import (advance_rows, layer_row_orders, attention_heads, attention_heads_applies, attn_tail_ffn_advance,
                         attn_tail_ffn_applies, class_head_max_times,
                         class_max_times, encoder_finalize, encoder_prepare_sorted, encoder_reference_points, fused_ffn,
                         fused_ffn_advance, fused_ffn_applies, fused_layer_norm, gather_rows, masked_topk_desc, scatter_rows_,
                         select_stack, token_linear_applies, token_linear_ln, topk_self_attention_, topk_select_inproj,
                         topk_select_inproj_applies, prepare_class_score_applies,
                         topk_self_attention_applies)
from .layer_norm_train import add_layer_norm
from .linear_x3 import X3Linear, x3_ffn, x3_ffn_applies
from .ms_deform_attn import MultiScaleDeformableAttention, batched_value_maps, plan_batched_value_maps
from .pyramid import PositionEmbeddingLearned


def _needs_grad(module: nn.Module, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


class SalienceTransformerEncoderLayer(nn.Module):
    """One encoder layer (salience_transformer.py:298-396): top-``topk_sa`` dense self-attention,
    multi-scale deformable self-attention over the filtered queries, FFN."""

    def __init__(self, embed_dim=256, d_ffn=1024, dropout=0.1, n_heads=8, activation=nn.ReLU(inplace=True),
                 n_levels=4, n_points=4, topk_sa=300):
        super().__init__()
        self.embed_dim = embed_dim
        self.topk_sa = topk_sa
        self.n_heads = n_heads
        # training: the attention's in / out projections (F.linear on the MHA module's own parameters) through
        # linear_x3.x3_linear (set by use_x3_linear_)
        self.x3_projections = False
        # the default on the bf16 path: in-projection (with the gather and the position add in its operand loads) and
        # attention + out_proj + residual + pre_norm + scatter as two launches of csrc/topk_attention.hip
        self.two_launch_topk_attention = True
        # the MSDA offset | weight projection of the layer's rows rides in the top-k attention's launch
        # (csrc/fused_head_value.hip); False = a launch of its own after the attention
        self.carry_sampling_projection = True
        # output_proj + residual + norm1 inside the feed-forward's launch (csrc/ffn.hip, TAIL form); False = a launch of
        # their own in front of it (tests compare the two)
        self.fuse_attention_tail = True
        # pre attention
        self.pre_attention = nn.MultiheadAttention(embed_dim, n_heads, dropout, batch_first=True)
        self.pre_dropout = nn.Dropout(dropout)
        self.pre_norm = nn.LayerNorm(embed_dim)
        # self attention
        self.self_attn = MultiScaleDeformableAttention(embed_dim, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        # ffn
        self.linear1 = nn.Linear(embed_dim, d_ffn)
        self.activation = activation
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, embed_dim)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.init_weights()

    def init_weights(self):
        nn.init.xavier_uniform_(self.pre_attention.in_proj_weight)
        nn.init.xavier_uniform_(self.pre_attention.out_proj.weight)
        nn.init.xavier_uniform_(self.linear1.weight)
        nn.init.xavier_uniform_(self.linear2.weight)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, query):
        if (isinstance(self.linear1, X3Linear) and isinstance(self.linear2, X3Linear) and isinstance(self.activation, nn.ReLU)
                and not (self.training and self.dropout2.p > 0.0) and x3_ffn_applies(query, self.linear1, self.linear2)):
            src2 = x3_ffn(query, self.linear1, self.linear2)   # ReLU inside the products (linear_x3._FfnX3)
            return add_layer_norm(query, self.norm2, self.dropout3(src2))
        src2 = self.linear2(self.dropout2(self.activation(self.linear1(query))))
        return add_layer_norm(query, self.norm2, self.dropout3(src2))

    def _forward_ffn_native(self, query, advance=None):
        """No-grad FFN: ReLU in the first GEMM's epilogue when the activation is ReLU, residual + LayerNorm in
        one launch.  ``advance`` = ``(sorted_result, next_rows, tokens, sorted_index, count)`` (the encoder's sorted
        loop): the end-of-layer row bookkeeping (``advance_rows``) is part of the operator and the NEXT layer's
        queries are returned instead of this layer's output."""
        if fused_ffn_applies(query, self.linear1, self.linear2, self.norm2, self.activation):
            if advance is not None and query.dim() == 3 and query.is_contiguous():
                return fused_ffn_advance(query, self.linear1, self.linear2, self.norm2, *advance)
            out = fused_ffn(query, self.linear1, self.linear2, self.norm2)   # hidden state stays in registers
            return out if advance is None else advance_rows(out, *advance)
        if isinstance(self.activation, nn.ReLU):
            x2d = query.reshape(-1, query.shape[-1])
            try:
                hidden = torch._addmm_activation(self.linear1.bias, x2d, self.linear1.weight.t(), use_gelu=False)
            except (RuntimeError, AttributeError):
                hidden = F.relu(F.linear(x2d, self.linear1.weight, self.linear1.bias))
        else:
            hidden = self.activation(self.linear1(query))
        src2 = F.linear(hidden, self.linear2.weight, self.linear2.bias).view(query.shape)
        out = fused_layer_norm(query, self.norm2, residual=src2)
        return out if advance is None else advance_rows(out, *advance)

    def _pre_attention(self, qk: Tensor, v: Tensor) -> Tensor:
        """nn.MultiheadAttention(q=k=qk, value=v) with the module's own parameters
        (salience_transformer.py:371-376).  The 300-token problem is launch-latency bound, so it is arranged as
        few launches: ONE in-projection GEMM over the stacked [qk ; v] rows (q,k are read from the first half,
        v from the second), one fused scaled-dot-product attention, one out-projection.  Under autograd (fp32, 32-channel
        heads, no attention dropout): the q | k and the v projections as two GEMMs on their own rows and the attention
        core as one launch forward, two backward (``attention_train.attention_qk_v``) -- no head-split copies."""
        mha = self.pre_attention
        E = qk.shape[-1]
        if (torch.is_grad_enabled() and (qk.requires_grad or v.requires_grad or mha.in_proj_weight.requires_grad)
                and not (self.training and mha.dropout > 0.0)):
            w, b = mha.in_proj_weight, mha.in_proj_bias
            linear = F.linear
            if self.x3_projections and qk.dtype == torch.float32:
                from .linear_x3 import x3_linear as linear
            w_qk, w_v = split_prefix(w, 2 * E, 0)
            b_qk, b_v = split_prefix(b, 2 * E, 0)
            pqk = linear(qk, w_qk, b_qk)
            pv = linear(v, w_v, b_v)
            if attention_train.applies(pqk, pv, mha.num_heads):
                o = attention_train.attention_qk_v(pqk, pv, mha.num_heads)
                return linear(o, mha.out_proj.weight, mha.out_proj.bias)
        return self._pre_attention_stacked(torch.cat([qk, v], 1), qk.shape[1], qk.requires_grad)

    def _pre_attention_stacked(self, stacked: Tensor, N: int, needs_grad: bool = False,
                               apply_out_proj: bool = True) -> Tensor:
        """``stacked`` = ``[q/k rows ; value rows]`` [B, 2N, E] (see ``_pre_attention``).  ``apply_out_proj=False``
        returns the concatenated heads for a caller that fuses ``out_proj`` with the residual + norm."""
        mha = self.pre_attention
        B, _, E = stacked.shape
        H = mha.num_heads
        hd = E // H
        linear = F.linear
        if self.x3_projections and torch.is_grad_enabled() and stacked.dtype == torch.float32:
            from .linear_x3 import x3_linear as linear   # weight / bias gradients from the x3 kernel (the library takes
            # 145 us for the 256 x 600 x 256 weight gradient of out_proj: one 256 x 256 tile, no split of the reduction)
        proj = linear(stacked, mha.in_proj_weight, mha.in_proj_bias)   # [B, 2N, 3E]
        q = proj[:, :N, :E].view(B, N, H, hd).transpose(1, 2)
        k = proj[:, :N, E:2 * E].view(B, N, H, hd).transpose(1, 2)
        vv = proj[:, N:, 2 * E:].view(B, N, H, hd).transpose(1, 2)
        drop = mha.dropout if self.training else 0.0
        o = None
        if not (torch.is_grad_enabled() and needs_grad) and drop == 0.0:
            pq, pk, pv = proj[:, :N, :E], proj[:, :N, E:2 * E], proj[:, N:, 2 * E:]
            if attention_heads_applies(pq, pk, pv, H):
                # own flash kernel on the strided slices of the projection: heads come out concatenated, no layout copy
                o = attention_heads(pq, pk, pv, H)
                return o if not apply_out_proj else F.linear(o, mha.out_proj.weight, mha.out_proj.bias)
        if not (torch.is_grad_enabled() and needs_grad):
            try:
                o = F.scaled_dot_product_attention(q, k, vv, dropout_p=drop)
            except RuntimeError:
                o = None
        if o is None:
            att = torch.matmul(q * (1.0 / math.sqrt(hd)), k.transpose(-1, -2)).softmax(-1)
            if drop > 0:
                att = F.dropout(att, drop)
            o = torch.matmul(att, vv)
        o = o.transpose(1, 2).reshape(B, N, E)
        if not apply_out_proj:
            return o
        return linear(o, mha.out_proj.weight, mha.out_proj.bias)

    def forward_sorted(self, query, pos_sorted, ref_sorted, fg_sorted, value_hm, spatial_shapes, level_start_index,
                       class_head, level_shapes=None, selection_hook=None, advance=None, mc_score=None,
                       want_next_score=False, row_order=None, orders_job=None):
        """No-grad layer body for index sets that are prefixes of one sorted list (the encoder keeps the tokens
        in sorted order, see ``SalienceTransformerEncoder.forward``).  ``query`` [B,c,E] is this layer's own copy
        (updated in place); ``pos_sorted`` [B,n0,E], ``ref_sorted`` [B,n0,L,2], ``fg_sorted`` [B,n0] are the
        sorted-order buffers of which the first ``c`` rows belong to this layer.  Same arithmetic as ``forward``.
        With ``advance`` (see ``_forward_ffn_native``) the layer finishes with the encoder's row bookkeeping and
        returns the next layer's queries.  ``mc_score`` [B,c]: this layer's selection score if the previous layer's last
        launch already produced it; ``want_next_score``: return ``(next queries, next layer's score or None)``."""
        c = query.shape[1]
        proj = None
        if mc_score is not None:
            pass
        elif token_linear_applies(query, class_head.weight):
            mc_score = class_head_max_times(query, class_head, fg_sorted[:, :c])   # logits never materialised
        else:
            mc_score = class_max_times(class_head(query), fg_sorted[:, :c])
        fuse_tail = token_linear_applies(query, self.self_attn.output_proj.weight) and self.embed_dim == 256
        carried = (fuse_tail and not self.training and self.two_launch_topk_attention and self.carry_sampling_projection
                   and self.self_attn.head_major_projection_applies(query, value_hm))
        inproj = None
        if (carried and selection_hook is None
                and topk_select_inproj_applies(mc_score, self.topk_sa, query, pos_sorted, self.pre_attention, self.pre_norm)):
            # the selection's launch also does the in-projection of the rows it selects (csrc/topk.hip)
            inproj = topk_select_inproj(mc_score, self.topk_sa, query, pos_sorted, self.pre_attention, orders_job=orders_job)
            sel = inproj.selected
        else:
            # (``orders_job``: the encoder's pending row orders ride in this selection's launch -- layer 0)
            sel = masked_topk_desc(mc_score, self.topk_sa, want_scores=False, orders_job=orders_job)[1]
        if orders_job is not None:
            orders_job.run()             # (no-op when the launch carried it)
        if selection_hook is not None:   # instrumentation: record the layer's top-k set, or force a given one
            sel = selection_hook(sel)
        N = sel.shape[1]
        if (fuse_tail and not self.training and self.two_launch_topk_attention and sel.is_contiguous()
                and topk_self_attention_applies(query, pos_sorted, self.pre_attention, self.pre_norm, N)):
            # gather + (x + pos) + in-projection, then attention + out_proj + residual + pre_norm + scatter: two launches,
            # no library GEMM (csrc/topk_attention.hip)
            if self.carry_sampling_projection and self.self_attn.head_major_projection_applies(query, value_hm):
                # the MSDA offset | weight projection of all rows rides in the attention's launch
                proj = topk_self_attention_(query, pos_sorted, sel, self.pre_attention, self.pre_norm,
                                            projection=self.self_attn._fused_query_projection_head_major(),
                                            inprojection=inproj)
            else:
                topk_self_attention_(query, pos_sorted, sel, self.pre_attention, self.pre_norm)
            stacked = None
        else:
            stacked = select_stack(query, pos_sorted, sel)                   # [q+pos ; q] rows, [B,2N,E]
        if fuse_tail:
            # The 2 x 300 selected rows are too few for the token-resident kernel (its weight copy + row-strided
            # epilogue cost 16 us whatever the row count; library GEMM + fused norm/scatter: 11 us) ...
            if stacked is not None:
                tgt2 = self._pre_attention_stacked(stacked, N)
                fused_layer_norm(stacked[:, N:], self.pre_norm, residual=tgt2, scatter_index=sel, scatter_into=query)
            sampled = self.self_attn.forward_native(query, ref_sorted[:, :c], value_hm, spatial_shapes,
                                                    level_start_index, query_pos=pos_sorted[:, :c],
                                                    apply_output_proj=False, level_shapes=level_shapes,
                                                    head_major_projection=proj, row_order=row_order)
            if (advance is not None and self.fuse_attention_tail and not self.training
                    and attn_tail_ffn_applies(sampled, query, self.self_attn.output_proj, self.norm1, self.linear1,
                                              self.linear2, self.norm2, self.activation)):
                # output_proj + residual + norm1 run in FRONT of the feed-forward inside its launch (csrc/ffn.hip, TAIL
                # form): one launch and one [rows, 256] round trip less per layer
                if want_next_score:
                    # ... and the next layer's class score comes out of the same launch's epilogue when it can
                    return attn_tail_ffn_advance(sampled, query, self.self_attn.output_proj, self.norm1, self.linear1,
                                                 self.linear2, self.norm2, *advance, next_class_head=class_head,
                                                 foreground=fg_sorted)
                return attn_tail_ffn_advance(sampled, query, self.self_attn.output_proj, self.norm1, self.linear1,
                                             self.linear2, self.norm2, *advance)
            # output_proj + residual + norm1 in one launch of the token-resident kernel at every layer size (below ~12 000
            # rows the library GEMM + separate LayerNorm is 1-2 us faster, but keeps hipBLASLt in the hot-path graph)
            query = token_linear_ln(sampled, self.self_attn.output_proj, self.norm1, residual=query)
            out = self._forward_ffn_native(query, advance)
            return (out, None) if want_next_score else out
        tgt2 = self._pre_attention_stacked(stacked, N)
        # pre_norm(select_tgt + tgt2) written straight back to the selected rows of the layer's queries
        fused_layer_norm(stacked[:, N:], self.pre_norm, residual=tgt2, scatter_index=sel, scatter_into=query)
        src2 = self.self_attn.forward_native(query, ref_sorted[:, :c], value_hm, spatial_shapes, level_start_index,
                                             query_pos=pos_sorted[:, :c], level_shapes=level_shapes, row_order=row_order)
        out = self._forward_ffn_native(fused_layer_norm(query, self.norm1, residual=src2), advance)
        return (out, None) if want_next_score else out

    def forward(self, query, query_pos, value, reference_points, spatial_shapes, level_start_index,
                query_key_padding_mask=None, score_tgt=None, foreground_pre_layer=None, value_hm=None, level_shapes=None):
        """Reference signature (salience_transformer.py:353-364) plus an optional pre-projected
        head-major ``value_hm`` (``[B,M,Nv,D]``) supplied by the encoder's batched value projection."""
        native = not _needs_grad(self, query, value)
        if native:
            mc_score = class_max_times(score_tgt, foreground_pre_layer)
            select_tgt_index = masked_topk_desc(mc_score, self.topk_sa, want_scores=False)[1]
            select_tgt = gather_rows(query, select_tgt_index)
            select_pos = gather_rows(query_pos, select_tgt_index)
        else:
            if score_tgt.is_cuda and score_tgt.dtype == torch.float32 and score_tgt.shape[1] >= self.topk_sa:
                # the selection carries no gradient: the no-grad path's kernels (same order: descending, ties by position)
                with torch.no_grad():
                    mc_score = class_max_times(score_tgt.detach(), foreground_pre_layer.detach())
                    select_tgt_index = masked_topk_desc(mc_score, self.topk_sa, want_scores=False)[1]
            else:
                mc_score = score_tgt.max(-1)[0] * foreground_pre_layer
                select_tgt_index = torch.sort(mc_score, dim=1, descending=True, stable=True)[1][:, :self.topk_sa]
            index_e = select_tgt_index.unsqueeze(-1).expand(-1, -1, self.embed_dim)
            select_tgt = torch.gather(query, 1, index_e)
            select_pos = torch.gather(query_pos, 1, index_e)
        tgt2 = self._pre_attention(self.with_pos_embed(select_tgt, select_pos), select_tgt)
        if native:
            select_tgt = fused_layer_norm(select_tgt, self.pre_norm, residual=tgt2)
        else:
            select_tgt = add_layer_norm(select_tgt, self.pre_norm, self.pre_dropout(tgt2))
        if native:
            query = scatter_rows_(query, select_tgt_index, select_tgt)  # query is the layer's own gathered copy
        else:
            query = query.scatter(1, index_e, select_tgt)

        # multi-scale deformable self attention
        if native:
            if value_hm is None:
                value_hm = self.self_attn.project_value(value, query_key_padding_mask)
            src2 = self.self_attn.forward_native(self.with_pos_embed(query, query_pos), reference_points, value_hm,
                                                 spatial_shapes, level_start_index, level_shapes=level_shapes)
        else:
            src2 = self.self_attn(query=self.with_pos_embed(query, query_pos), reference_points=reference_points,
                                  value=value, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                  key_padding_mask=query_key_padding_mask)
        if native:
            return self._forward_ffn_native(fused_layer_norm(query, self.norm1, residual=src2))
        query = add_layer_norm(query, self.norm1, self.dropout1(src2))
        return self.forward_ffn(query)


class _SplitPrefix(torch.autograd.Function):
    """``x.narrow(dim, 0, n)`` and the rest as views whose backward is ONE concatenation (the two slices' own backward nodes
    each zero-fill a tensor of x's size, copy their part in, and the results are added: five launches where this is one)."""

    @staticmethod
    def forward(ctx, x, n, dim):
        ctx.n, ctx.tail, ctx.dim = n, x.shape[dim] - n, dim
        return x.narrow(dim, 0, n), x.narrow(dim, n, x.shape[dim] - n)

    @staticmethod
    def backward(ctx, g_head, g_tail):
        if g_head is None and g_tail is None:
            return None, None, None
        ref = g_head if g_head is not None else g_tail   # (an unused part arrives as None: zeros of its shape)
        shape = list(ref.shape)
        if g_head is None:
            shape[ctx.dim] = ctx.n
            g_head = ref.new_zeros(shape)
        if g_tail is None:
            shape[ctx.dim] = ctx.tail
            g_tail = ref.new_zeros(shape)
        return torch.cat([g_head, g_tail], ctx.dim), None, None


def split_prefix(x: Tensor, n: int, dim: int = 1):
    """``(x[..., :n, ...], x[..., n:, ...])`` along ``dim`` (views; see ``_SplitPrefix``)."""
    return _SplitPrefix.apply(x, n, dim)


class SalienceTransformerEncoder(nn.Module):
    """Encoder over salience-filtered queries (salience_transformer.py:399-497)."""

    def __init__(self, encoder_layer: nn.Module, num_layers: int = 6, max_num_embedding=200):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.embed_dim = encoder_layer.embed_dim
        # learnt background embed for prediction
        self.background_embedding = PositionEmbeddingLearned(max_num_embedding, num_pos_feats=self.embed_dim // 2)
        # optional instrumentation: a callable(tag) invoked at every layer boundary of the loop
        # (bench.py records a stream event there to report ms per encoder layer)
        self.layer_marker = None
        # optional instrumentation (tests / bench.py): callable(layer_id, sel [B,topk] int64) -> sel, invoked with every
        # layer's top-k selection (return it unchanged to record, or another index set to teacher-force the layer);
        # `max_layers` stops the loop after that many layers (truncated graphs: ms per encoder layer under replay)
        self.selection_hook = None
        self.max_layers = None
        self.init_weights()

    def init_weights(self):
        for layer in self.layers:
            if hasattr(layer, "init_weights"):
                layer.init_weights()

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel-centre reference points scaled by the valid ratios -> ``[B,S,L,2]``
        (salience_transformer.py:418-432).  ``spatial_shapes`` may be a tensor or a list of (h, w)."""
        shapes = spatial_shapes.tolist() if isinstance(spatial_shapes, Tensor) else list(spatial_shapes)

        def build():
            xs, ys, lv, sz = [], [], [], []
            for lvl, (h, w) in enumerate(shapes):
                ys.append((torch.arange(h, dtype=torch.float32) + 0.5).view(h, 1).expand(h, w).reshape(-1))
                xs.append((torch.arange(w, dtype=torch.float32) + 0.5).view(1, w).expand(h, w).reshape(-1))
                lv.append(torch.full((h * w,), lvl, dtype=torch.int64))
                sz.append(torch.tensor([float(w), float(h)]).expand(h * w, 2))
            pix = torch.stack([torch.cat(xs), torch.cat(ys)], -1)   # [S,2] pixel centres (x+0.5, y+0.5)
            return pix.to(device), torch.cat(lv).to(device), torch.cat(sz).to(device)

        pix, lv, sz = pyramid.static_tensor(("refgrid", tuple(map(tuple, shapes)), str(device)), build)
        own = valid_ratios[:, lv]                                         # [B,S,2] the token's own level ratio
        centre = pix[None] / (own * sz[None])                             # (idx + 0.5) / (valid_ratio * size)
        return centre[:, :, None] * valid_ratios[:, None]

    @staticmethod
    def _prefix_counts(foreground_inds) -> Optional[List[int]]:
        """Row counts per layer when ``foreground_inds[k]`` are views ``sorted[:, :c_k]`` of ONE contiguous
        ``[B,n0]`` tensor with non-increasing ``c_k`` (checked structurally: same storage offset and strides --
        no device read), else ``None``."""
        first = foreground_inds[0]
        if first.dim() != 2 or first.dtype != torch.int64 or not first.is_contiguous() or first.shape[1] == 0:
            return None
        counts = []
        for t in foreground_inds:
            if (t.dim() != 2 or t.dtype != torch.int64 or t.data_ptr() != first.data_ptr() or t.shape[0] != first.shape[0]
                    or t.shape[1] == 0 or (t.shape[1] > 1 and t.stride(1) != 1)
                    or (t.shape[0] > 1 and t.stride(0) != first.stride(0))
                    or (counts and t.shape[1] > counts[-1])):
                return None
            counts.append(int(t.shape[1]))
        return counts

    def _forward_sorted_autograd(self, value, ori_pos, padding_mask, foreground_score, focus_token_nums, foreground_inds,
                                 counts, spatial_shapes, level_start_index, valid_ratios, level_shapes, multi_level_masks):
        """The autograd path when every layer's index set is a prefix of ONE sorted list (what ``salience_filtering``
        produces): the tokens are gathered ONCE into sorted order, layer k works on the rows ``[:c_k]`` (a view), rows
        beyond an image's focus count keep their value, and the result goes back to token space once -- the reference's
        per-layer gather / scatter pair on the whole ``[B,Nv,E]`` tensor (salience_transformer.py:454-485; in the
        backward: a zero fill, a scatter-add and an add of that size per layer) is left to the general loop."""
        E = self.embed_dim
        b = value.shape[0]
        sorted_index = foreground_inds[0]
        n0 = counts[0]
        idx_e = sorted_index.unsqueeze(-1).expand(-1, -1, E)
        cur = torch.gather(value, 1, idx_e)
        pos_s = torch.gather(ori_pos, 1, idx_e)
        with torch.no_grad():   # (the selection score and the sampling centres carry no gradient)
            fg_s = torch.gather(foreground_score.detach(), 1, sorted_index)
            ref_s = encoder_reference_points(valid_ratios.float().contiguous(), spatial_shapes, level_start_index, n0,
                                             index=sorted_index)
            live = torch.arange(n0, device=value.device)[None] < focus_token_nums.to(torch.int64)[:, None]    # [B,n0]
        final = []
        q = cur                                 # the rows of the layer about to run (c_0 = all gathered rows)
        for layer_id, layer in enumerate(self.layers):
            if self.layer_marker is not None:
                self.layer_marker(layer_id)
            c = counts[layer_id]
            with torch.no_grad():   # (the selection score is only used detached: no graph for the class head here)
                score_tgt = self.enhance_mcsp(q.detach())
            out = layer(q, pos_s[:, :c], value, ref_s[:, :c], spatial_shapes, level_start_index, padding_mask,
                        score_tgt, fg_s[:, :c])
            out = torch.where(live[:, :c, None], out, q)
            nxt = counts[layer_id + 1] if layer_id + 1 < self.num_layers else 0
            if nxt == 0:
                final.append(out)
            elif nxt == c:
                q = out                         # (equal ratios: the next layer takes all of them, nothing is final yet)
            else:
                q, done = split_prefix(out, nxt)   # rows the next layer takes | rows no later layer touches
                final.append(done)
        if self.layer_marker is not None:
            self.layer_marker(self.num_layers)
        output = value.scatter(1, idx_e, torch.cat(final[::-1], 1))
        if multi_level_masks is not None:
            # learnt embedding for background tokens: neither padding nor in the LAST layer's set (:487-495)
            bg = self.background_embedding.flat(level_shapes).to(output.dtype)
            keep = torch.ones(b, value.shape[1], dtype=output.dtype, device=output.device)
            keep.scatter_(1, sorted_index[:, :counts[-1]], 0.0)
            keep = keep * (~padding_mask).to(output.dtype)
            output = torch.addcmul(output, bg.unsqueeze(0), keep.unsqueeze(-1))
        return output

    # the value maps take the bordered layout (zero records around every level) and the deformable attention walks
    # its rows in a per-layer spatial order (csrc/msda_resident.hip, msda_bordered_kernel); False = round 3's plain maps
    # (class attributes: an A/B script sets them on the class or an instance; nothing here reads the environment)
    bordered_value_maps = True
    row_order_tile = 16

    def project_values(self, value: Tensor, padding_mask: Optional[Tensor], level_shapes=None) -> Tensor:
        """Head-major value maps of ALL layers ``[num_layers,B,heads,Nv,D]`` (no-grad path).  The six layers sample
        the same, never-updated feature map (salience_transformer.py:452), so their ``value_proj`` run as one
        projection; it only depends on the flattened features, which lets a caller overlap it with the filtering
        stage on a second stream (``SalienceEncoderHotPath`` does)."""
        return batched_value_maps([l.self_attn for l in self.layers], value, padding_mask,
                                  level_shapes=level_shapes if self.bordered_value_maps else None)

    def plan_finalize(self, value: Tensor, padding_mask: Optional[Tensor], level_shapes):
        """The token-space pass of the output (``tokens + background`` outside the padding) as a pending
        ``filter_ops.FinalizeJob`` (16-bit tokens only; ``None`` otherwise)."""
        from .filter_ops import FinalizeJob
        if value.dtype not in (torch.bfloat16, torch.float16) or value.dim() != 3 or value.shape[2] != 256 or not value.is_contiguous():
            return None
        return FinalizeJob(value, self.background_embedding.flat_cached(level_shapes, value.dtype), padding_mask)

    def plan_values(self, value: Tensor, padding_mask: Optional[Tensor], parts=2, level_shapes=None):
        """``project_values`` as pending jobs ``(maps, [ValueProjectionJob, ...])`` (``None`` when the one-launch
        projection does not apply): the caller lets other launches carry the jobs (the salience head's stage 1 on the
        coarse levels), runs the rest, and hands ``maps`` to ``forward`` as ``precomputed_value_maps``."""
        return plan_batched_value_maps([l.self_attn for l in self.layers], value, padding_mask, parts=parts,
                                       level_shapes=level_shapes if self.bordered_value_maps else None)

    def forward(self, query, spatial_shapes, level_start_index, valid_ratios, query_pos=None,
                query_key_padding_mask=None, foreground_score=None, focus_token_nums=None, foreground_inds=None,
                multi_level_masks=None, precomputed_value_maps=None, finalize_job=None):
        """Reference signature (salience_transformer.py:434-447).  ``foreground_inds`` is the list of
        per-layer ``[B,Nq_k]`` index tensors, ``focus_token_nums`` ``[B]`` the per-image valid prefix.
        ``precomputed_value_maps``: the result of ``project_values(query, query_key_padding_mask)`` if the caller
        already launched it; ``finalize_job``: the token-space pass of the output if a caller had another launch carry it
        (``plan_finalize``)."""
        if not query.is_cuda:
            raise RuntimeError("SalienceTransformerEncoder: HIP device tensors required; there is no CPU fallback")
        native = not _needs_grad(self, query, query_pos)
        E = self.embed_dim
        level_shapes = pyramid.level_shapes_of(multi_level_masks) if multi_level_masks is not None \
            else [tuple(s) for s in spatial_shapes.tolist()]
        counts = self._prefix_counts(foreground_inds)
        b, n = query.shape[:2]
        s, p = len(level_shapes), 2
        from .filter_ops import LazyForegroundScore
        if isinstance(foreground_score, LazyForegroundScore) and not (native and counts is not None):
            foreground_score = foreground_score.materialize()   # (only the sorted no-grad loop fills the rows it gathers itself)
        if counts is not None and not native:
            return self._forward_sorted_autograd(query, query_pos, query_key_padding_mask, foreground_score, focus_token_nums,
                                                 foreground_inds, counts, spatial_shapes, level_start_index, valid_ratios,
                                                 level_shapes, multi_level_masks)
        if counts is None:
            reference_points = self.get_reference_points(level_shapes, valid_ratios, device=query.device)
            ori_reference_points = reference_points.reshape(b, n, s * p).contiguous()
        ori_pos = query_pos
        value = query
        output = query
        focus64 = focus_token_nums.to(torch.int64).contiguous()

        value_hm_all = precomputed_value_maps
        if native and value_hm_all is None:
            # (bordered maps only for the sorted loop below: the general loop's layers take plain maps)
            value_hm_all = self.project_values(value, query_key_padding_mask, level_shapes if counts is not None else None)

        if counts is not None:
            # every layer's set is a prefix of one sorted list (what salience_filtering produces): keep the tokens
            # in sorted order across the layers -- one gather in, one pass back to token space at the end
            sorted_index = foreground_inds[0]
            n0 = counts[0]
            if (value.is_contiguous() and ori_pos.is_contiguous() and ori_pos.dtype == value.dtype
                    and foreground_score.dtype == torch.float32 and foreground_score.is_contiguous()
                    and 256 % max(1, value.shape[-1] * value.element_size() // 16) == 0
                    and (value.shape[-1] * value.element_size()) % 16 == 0
                    # (the kernel's reference-point lanes: 2 x levels of a row's 16-byte lanes -- narrow rows, e.g. 32 bf16
                    # channels with 4 levels, take the gather path below instead of the kernel's hard failure)
                    and value.shape[-1] * value.element_size() // 16 >= 2 * len(level_shapes)):
                # query / position rows, foreground scores and reference points of the selected tokens: one launch
                # (... and the first layer's class score of those rows when the kernel covers the head)
                head0 = self.enhance_mcsp if prepare_class_score_applies(value, foreground_score, self.enhance_mcsp) else None
                prepared = encoder_prepare_sorted(value, ori_pos, foreground_score, sorted_index, valid_ratios,
                                                  spatial_shapes, level_start_index, class_head=head0)
                q, pos_s, fg_s, ref_s = prepared[:4]
                score0 = prepared[4] if head0 is not None else None
            else:
                score0 = None
                if isinstance(foreground_score, LazyForegroundScore):
                    foreground_score = foreground_score.materialize()
                q = gather_rows(value, sorted_index)
                pos_s = gather_rows(ori_pos, sorted_index)
                # reference points of the selected tokens only, straight from their indices
                ref_s = encoder_reference_points(valid_ratios.float().contiguous(), spatial_shapes, level_start_index, n0,
                                                 index=sorted_index)
                fg_s = torch.gather(foreground_score, 1, sorted_index)
            result = torch.empty_like(q)
            score = score0
            # per-layer row orders for the deformable attention (tile-major walk of each layer's rows)
            orders, orders_job = None, None
            from .ms_deform_attn import is_bordered
            if value_hm_all is not None and is_bordered(value_hm_all[0], level_shapes) and self.row_order_tile:
                # pending: the first layer's top-300 selection (2 workgroups on an empty chip) carries the job
                orders_job = layer_row_orders(sorted_index, counts, level_shapes, tile=self.row_order_tile, as_job=True)
                orders = None if orders_job is None else orders_job.orders
            for layer_id, layer in enumerate(self.layers):
                if self.max_layers is not None and layer_id >= self.max_layers:
                    break
                if self.layer_marker is not None:
                    self.layer_marker(layer_id)
                hook = None
                if self.selection_hook is not None:
                    hook = (lambda sel, k=layer_id: self.selection_hook(k, sel))
                nxt = counts[layer_id + 1] if layer_id + 1 < self.num_layers else 0
                # the layer ends with the row bookkeeping (live rows recorded in `result`, next layer's queries) and,
                # when its last launch can, the next layer's selection score
                q, score = layer.forward_sorted(q, pos_s, ref_s, fg_s, value_hm_all[layer_id], spatial_shapes,
                                                level_start_index, self.enhance_mcsp, level_shapes=level_shapes,
                                                selection_hook=hook, advance=(result, nxt, value, sorted_index, focus64),
                                                mc_score=score, want_next_score=True,
                                                row_order=None if orders is None else orders[layer_id],
                                                orders_job=orders_job if layer_id == 0 else None)
            if self.layer_marker is not None:
                self.layer_marker(self.num_layers)
            if multi_level_masks is not None:
                bg = self.background_embedding.flat_cached(level_shapes, value.dtype)
                return encoder_finalize(value, result, sorted_index, focus64, bg, query_key_padding_mask, counts[-1],
                                        finalize_job=finalize_job)
            return scatter_rows_(value.clone(), sorted_index, result, count=focus64)

        if native:
            output = query.clone()   # the general path scatters every layer's rows back in place
        inds = None
        for layer_id, layer in enumerate(self.layers):
            if self.layer_marker is not None:
                self.layer_marker(layer_id)
            inds = foreground_inds[layer_id]
            if native:
                inds = inds.contiguous()
                q = gather_rows(output, inds)
                q_pos = gather_rows(ori_pos, inds)
                fg = torch.gather(foreground_score, 1, inds)
                ref = gather_rows(ori_reference_points, inds).view(b, -1, s, p)
            else:
                inds_e = inds.unsqueeze(-1).expand(-1, -1, E)
                q = torch.gather(output, 1, inds_e)
                q_pos = torch.gather(ori_pos, 1, inds_e)
                fg = torch.gather(foreground_score, 1, inds)
                ref = torch.gather(ori_reference_points, 1, inds.unsqueeze(-1).repeat(1, 1, s * p)).view(b, -1, s, p)
            score_tgt = self.enhance_mcsp(q)
            q = layer(q, q_pos, value, ref, spatial_shapes, level_start_index, query_key_padding_mask, score_tgt, fg,
                      value_hm=value_hm_all[layer_id] if native else None, level_shapes=level_shapes if native else None)
            if native:
                scatter_rows_(output, inds, q, count=focus64)
            else:
                rows = torch.arange(inds.shape[1], device=inds.device)[None] < focus64[:, None]   # [B,Nq]
                keep_old = torch.gather(output, 1, inds.unsqueeze(-1).expand(-1, -1, E))
                output = output.scatter(1, inds.unsqueeze(-1).expand(-1, -1, E),
                                        torch.where(rows[..., None], q, keep_old))

        if self.layer_marker is not None:
            self.layer_marker(self.num_layers)
        # learnt embedding for background tokens: every token that is neither padding nor in the LAST
        # layer's index set (salience_transformer.py:487-495)
        if multi_level_masks is not None:
            bg = self.background_embedding.flat(level_shapes).to(output.dtype)        # [S, E]
            keep = torch.ones(b, n, dtype=output.dtype, device=output.device)
            keep.scatter_(1, inds, 0.0)
            keep = keep * (~query_key_padding_mask).to(output.dtype)
            output = torch.addcmul(output, bg.unsqueeze(0), keep.unsqueeze(-1))
        return output
