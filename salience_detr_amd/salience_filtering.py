"""Hierarchical salience filtering (boundary B4).

The reference keeps this logic inline in ``SalienceTransformer.forward``
(``models/bricks/salience_transformer.py:116-168``); here it is two operators with pinned
inputs/outputs:

* ``level_filtering``   == lines 123-154: coarse-to-fine salience prediction and the per-level
  masked top-k, whose sort/selection runs in the single-workgroup HIP kernel of ``csrc/topk.hip``.
* ``salience_filtering`` == lines 116-121 + 156-168: token budgets, global sort of the selected
  scores, per-layer prefixes and the flattened foreground score.

Index outputs are bit-identical to the reference on tie-free inputs; equal scores are ordered
lower-index-first (torch leaves tie order unspecified).  ``salience_score`` stays differentiable
w.r.t. the mask predictor and ``alpha`` because only the index selection goes through the kernel.
"""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import pyramid
from .layer_norm_train import add_layer_norm
from .salience_encoder import split_prefix
from . import filter_ops
from .filter_ops import (column_mean, fused_layer_norm, masked_fill_min, masked_topk_desc, merge_sorted_desc,
                         plan_masked_topk, salience_head, salience_head_hoist)

# The fused no-grad path takes both 256 x 256 products of the head's stage 1 out of the coarse-to-fine chain: one launch
# for all levels' tokens in front of the level loop (filter_ops.salience_head_hoist), per level only the modulation step.
# False = stage 1 per level (the form up to round 5), kept for same-box comparisons and as the reference of the tests.
HOIST_HEAD = True
# Does the hoisted launch carry the first pending value-projection job?
HOIST_CARRIES_VALUE = False
# Who carries the finalize pass: "merge" = the merge launch of the finest level's sliced top-k (a few workgroups on an
# idle chip; falls back to that level's modulation launch when the top-k takes another form), a level index = that
# level's modulation launch, None = the hoisted launch (a full chip, where the pass costs what it costs alone)
FINALIZE_LEVEL = "merge"
# The deferred rank of level 1 on the merge launch of the finest level's sliced top-k (else on its modulation launch)
RANK_ON_MERGE = True


class _Modulate(torch.autograd.Function):
    """``x + x * row_scale[..., None] * alpha`` (the coarse-to-fine update, salience_transformer.py:143) with ``alpha``'s
    gradient -- a sum over every element of the level -- taken by the device's own column-sum kernel: the framework's
    multi-block reduction relies on a hipMemsetAsync that a replayed hipGraph does not reproduce on this stack (see
    ``pyramid._LevelPosEmbed.backward``; ``alpha``'s gradient came out 3 % off in the replayed training step)."""

    @staticmethod
    def forward(ctx, x, row_scale, alpha):
        ctx.save_for_backward(x, row_scale, alpha)
        return torch.addcmul(x, x, row_scale.unsqueeze(-1) * alpha)

    @staticmethod
    def backward(ctx, g):
        from .filter_ops import column_mean
        x, row_scale, alpha = ctx.saved_tensors
        gx = g_rs = g_alpha = None
        dot = (g * x).sum(-1)                                   # [B,N]: per-row reductions over the 256 channels
        if ctx.needs_input_grad[0]:
            gx = torch.addcmul(g, g, row_scale.unsqueeze(-1) * alpha)
        if ctx.needs_input_grad[1]:
            g_rs = dot * alpha
        if ctx.needs_input_grad[2]:
            t = (dot * row_scale).unsqueeze(-1).expand(-1, -1, 4).contiguous()        # [B,N,4]: the kernel's row format
            g_alpha = (column_mean(t)[:, 0, 0] * float(t.shape[1])).sum().reshape(alpha.shape)
        return gx, g_rs, g_alpha


class _GlobalHalfMean(torch.autograd.Function):
    """``cat([z[..., :half], z[..., half:].mean(1, keepdim=True).expand_as(...)], -1)`` (salience_transformer.py:43-45) with
    both column reductions -- the mean over the level's tokens and, backward, the sum of the broadcast half's gradient --
    on the device's own deterministic column-mean kernel (see ``_Modulate``: the framework's multi-block reductions are
    not replay-safe under hipGraph on this stack)."""

    @staticmethod
    def forward(ctx, z, half):
        from .filter_ops import column_mean
        ctx.half = half
        out = z.clone()
        out[..., half:] = column_mean(z[..., half:])            # [B,1,h/2] broadcast over the tokens
        return out

    @staticmethod
    def backward(ctx, g):
        from .filter_ops import column_mean
        half = ctx.half
        gz = g.clone()
        # d mean / d z[b,n,c] = 1/N: every token of the global half receives the MEAN of that half's gradient
        gz[..., half:] = column_mean(g[..., half:])
        return gz, None


class _ReplaySafeMean(torch.autograd.Function):
    """``x.mean()`` of a fp32 ``[B, n, C]`` device tensor through the column-mean kernel (per-image column means, then a
    mean of ``B * C`` numbers): the framework's one-kernel reduction of millions of elements clears its semaphores with a
    ``hipMemsetAsync`` node that a replayed hipGraph does not reproduce on this stack (see ``_Modulate``)."""

    @staticmethod
    def forward(ctx, x):
        from .filter_ops import column_mean
        ctx.shape = x.shape
        return column_mean(x.contiguous()).mean()

    @staticmethod
    def backward(ctx, g):
        n = 1
        for d in ctx.shape:
            n *= d
        return (g / n).expand(ctx.shape)


def replay_safe_mean(x: Tensor) -> Tensor:
    """Differentiable ``x.mean()`` that is safe inside a captured training step (losses over ``memory``)."""
    return _ReplaySafeMean.apply(x)


class MaskPredictor(nn.Module):
    """Salience head (salience_transformer.py:16-47); identical parameter names
    (``layer1.0`` LayerNorm, ``layer1.1`` Linear, ``layer2.{0,2,4}`` Linear)."""

    def __init__(self, in_dim: int, h_dim: int):
        super().__init__()
        self.h_dim = h_dim
        self.layer1 = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, h_dim), nn.GELU())
        self.layer2 = nn.Sequential(nn.Linear(h_dim, h_dim // 2), nn.GELU(), nn.Linear(h_dim // 2, h_dim // 4),
                                    nn.GELU(), nn.Linear(h_dim // 4, 1))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def fused_kernels_apply(self, x: Tensor) -> bool:
        """The three-launch MFMA path (csrc/salience_head.hip) is built for the released configuration:
        in_dim == h_dim == 256, fp32."""
        return (self.h_dim == 256 and x.shape[-1] == 256 and x.dtype == torch.float32
                and self.layer1[1].weight.dtype == torch.float32 and x.dim() == 3 and x.stride(2) == 1)

    def forward(self, x: Tensor, row_scale: Optional[Tensor] = None, alpha: Optional[Tensor] = None) -> Tensor:
        """``x`` [B,N,C].  With ``row_scale`` [B,N] (+ ``alpha``, one element) the input is first modulated as
        ``x + x * row_scale * alpha`` (the coarse-to-fine update of salience_transformer.py:143)."""
        half = self.h_dim // 2
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad or not x.is_cuda:
            if row_scale is not None:
                x = _Modulate.apply(x, row_scale, alpha) if x.is_cuda and x.dtype == torch.float32 else \
                    x + x * row_scale.unsqueeze(-1) * alpha
            z = add_layer_norm(x, self.layer1[0]) if x.is_cuda else self.layer1[0](x)   # (one launch each way on the device)
            for m in list(self.layer1)[1:]:
                z = m(z)
            # the "global" half is replaced by its mean over ALL tokens of the level, masked ones included
            if z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and half % 4 == 0:
                z = _GlobalHalfMean.apply(z, half)
            else:
                local, glob = split_prefix(z, half, -1) if z.is_cuda else (z[..., :half], z[..., half:])
                z = torch.cat([local, glob.mean(dim=1, keepdim=True).expand(-1, z.shape[1], -1)], dim=-1)
            return self.layer2(z)
        if self.fused_kernels_apply(x):
            return salience_head(x, self, row_scale=row_scale, alpha=alpha).unsqueeze(-1)
        # other widths: modulation + LayerNorm in one launch; the global half enters layer2[0] as a per-image
        # constant  W[:, half:] @ mean + b  (no [B,N,h] concat, half the GEMM), its token mean from a
        # deterministic column-mean kernel
        B, N, _ = x.shape
        z = fused_layer_norm(x, self.layer1[0], row_scale=row_scale, alpha=alpha)
        z = F.gelu(self.layer1[1](z))                                     # [B,N,h]
        lin = self.layer2[0]
        const = F.linear(column_mean(z[..., half:]), lin.weight[:, half:], lin.bias)   # [B,1,h/2]
        w_local_t = lin.weight[:, :half].t()
        h = torch.stack([torch.addmm(const[b], z[b, :, :half], w_local_t) for b in range(B)])
        h = F.gelu(h)
        h = F.gelu(self.layer2[2](h))
        return self.layer2[4](h)


def token_budgets(multi_level_masks: Sequence[Tensor], level_filter_ratio: Tensor):
    """Device-side budgets (salience_transformer.py:116-121).  Returns
    ``(focus_token_nums [B] int64, level_token_nums [L] int32, valid_token_nums [B,L])`` as device tensors;
    turning ``level_token_nums`` into python ints costs the one host sync this stage needs -- callers that
    know their image sizes use ``pyramid.host_token_budgets`` instead and never synchronise."""
    valid = torch.stack([(~m).sum((1, 2)) for m in multi_level_masks], -1)
    focus = (valid * level_filter_ratio).int()
    return focus.sum(-1), focus.max(0)[0], valid


def _next_value_jobs(value_jobs, stage2_only=None) -> dict:
    """The next two pending value-projection jobs as ``salience_head``'s ``value_job`` / ``value_job2``; with
    ``stage2_only`` (a job list) the next pending one of it for the stage-2 launch alone (a level whose stage-1 launch
    fills the chip by itself)."""
    if stage2_only is not None:
        pending = [j for j in stage2_only if not j.done]
        return dict(value_job=None, value_job2=pending[0] if pending else None)
    pending = [j for j in (value_jobs or ()) if not j.done]
    return dict(value_job=pending[0] if pending else None, value_job2=pending[1] if len(pending) > 1 else None)


def level_filtering(backbone_output_memory: Tensor, mask_flatten: Tensor, level_shapes: Sequence[Tuple[int, int]],
                    level_start_index: Sequence[int], level_token_nums: Sequence[int], mask_predictor: nn.Module,
                    alpha: Tensor, enc_output: Optional[nn.Module] = None, enc_output_norm: Optional[nn.Module] = None,
                    memory_out: Optional[Tensor] = None, score_flat: Optional[Tensor] = None,
                    extras: Optional[dict] = None, value_jobs: Optional[list] = None, finalize_job=None):
    """Coarse-to-fine salience scores + per-level top-k (salience_transformer.py:123-154).

    ``level_shapes`` / ``level_start_index`` / ``level_token_nums`` are python ints (shapes come from the
    tensors' own sizes; budgets from ``token_budgets``/``host_token_budgets``).
    Returns ``(salience_score: list[L] of [B,1,H_l,W_l], level_inds: list[L] of [B,k_l] int64 (global token
    index), level_score: list[L] of [B,k_l])`` ordered low level -> high level.

    With ``enc_output`` / ``enc_output_norm`` the first argument is the INPUT of ``enc_output`` (the masked
    ``feat + pos`` tokens of base_transformer.py:107-109) and the projection + norm run inside the salience-head
    kernel (no-grad MI355X path only); ``memory_out`` [B,S,C] then optionally receives ``backbone_output_memory``
    and ``score_flat`` [B,S] the flattened scores.  ``extras`` (a dict) receives by-products that
    ``salience_filtering`` can reuse: ``level_min`` [L] (``score.min()`` per level) and ``selected`` (the
    concatenated ``(scores, indices)`` [B, sum k] the per-level top-k calls already wrote side by side).
    ``value_jobs``: pending ``filter_ops.ValueProjectionJob`` slices of the encoder's value projection; the stage-1
    launches of the two coarsest levels (few workgroups on an otherwise empty chip) carry one each -- stage 1, then
    stage 2 -- coarsest level first.  ``finalize_job``: the pending token-space pass of the encoder's output
    (``filter_ops.FinalizeJob``); the first stage-1 launch without a value job carries it.
    """
    B = backbone_output_memory.shape[0]
    L = len(level_shapes)
    salience_score: List[Optional[Tensor]] = [None] * L
    level_inds: List[Optional[Tensor]] = [None] * L
    level_score: List[Optional[Tensor]] = [None] * L
    score = None
    fused = (isinstance(mask_predictor, MaskPredictor) and backbone_output_memory.is_cuda
             and mask_predictor.fused_kernels_apply(backbone_output_memory)
             and not (torch.is_grad_enabled() and (backbone_output_memory.requires_grad or alpha.requires_grad or
                                                   any(p.requires_grad for p in mask_predictor.parameters()))))
    if enc_output is not None and not fused:
        raise RuntimeError("level_filtering: enc_output fusion needs the no-grad fp32 256-wide MaskPredictor path")
    level_min = sel_score = sel_inds = None
    pending_rank = None
    defer_ranks = fused
    if fused:
        # per-level minima (stage 2 takes them) and ONE [B, sum k] buffer pair the per-level top-k calls fill column
        # block by column block (the concatenation of :155 for free)
        dev = backbone_output_memory.device
        level_min = torch.empty(L, dtype=torch.float32, device=dev)
        ks = [int(k) for k in level_token_nums]
        offs = [sum(ks[:i]) for i in range(L)]
        sel_score = torch.empty((B, sum(ks)), dtype=torch.float32, device=dev)
        sel_inds = torch.empty((B, sum(ks)), dtype=torch.int64, device=dev)
    hoisted = None
    if fused and HOIST_HEAD and filter_ops.salience_head_bf16x3:
        pending = [j for j in (value_jobs or ()) if not j.done]
        hoisted = salience_head_hoist(backbone_output_memory, mask_predictor, enc_output=enc_output,
                                      enc_output_norm=enc_output_norm, memory_out=memory_out,
                                      value_job=pending[0] if pending and HOIST_CARRIES_VALUE else None,
                                      finalize_job=finalize_job if FINALIZE_LEVEL is None else None)
    # (the modulation launches from this level down may carry the finalize pass: none of them for "merge")
    fin_level = FINALIZE_LEVEL if isinstance(FINALIZE_LEVEL, int) else (-1 if FINALIZE_LEVEL == "merge" else 0)
    for lvl in range(L - 1, -1, -1):
        h, w = level_shapes[lvl]
        start = int(level_start_index[lvl])
        level_memory = backbone_output_memory[:, start:start + h * w, :]
        mask = mask_flatten[:, start:start + h * w]
        if fused and hoisted is not None:
            # the level's step of the hoisted head: modulation -> const -> stage 2; every stage-2 launch of the three
            # coarsest levels carries the next pending value-projection job
            token_score = salience_head(
                level_memory, mask_predictor, coarse_score=score, level_hw=(h, w),
                alpha=alpha[lvl:lvl + 1] if score is not None else None,
                score_flat=None if score_flat is None else score_flat[:, start:start + h * w],
                score_min=level_min[lvl:lvl + 1],
                # (the finest level's modulation launch is as long as its own traffic: the rank of the level before rides on
                # the merge of this level's sliced top-k instead -- a few workgroups on an idle chip)
                rank_job=pending_rank if lvl > 0 or not RANK_ON_MERGE else None,
                finalize_job=finalize_job if lvl <= fin_level else None,
                hoisted=hoisted.level(start, h * w),
                **_next_value_jobs(None, stage2_only=value_jobs if lvl >= L - 3 and L > 3 and value_jobs else None))
        elif fused:
            # resize of the coarser score, modulation, both LayerNorms, all five Linear layers: three launches
            token_score = salience_head(
                level_memory, mask_predictor, coarse_score=score, level_hw=(h, w),
                alpha=alpha[lvl:lvl + 1] if score is not None else None, enc_output=enc_output,
                enc_output_norm=enc_output_norm,
                memory_out=None if memory_out is None else memory_out[:, start:start + h * w, :],
                score_flat=None if score_flat is None else score_flat[:, start:start + h * w],
                score_min=level_min[lvl:lvl + 1],
                rank_job=pending_rank, finalize_job=finalize_job,
                **_next_value_jobs(value_jobs if lvl >= L - 2 and L > 2 else None,
                                   stage2_only=value_jobs if lvl == L - 3 and L > 3 else None))
        if fused:
            late_rank = None
            if pending_rank is not None:
                if hoisted is not None and lvl == 0 and RANK_ON_MERGE:
                    late_rank = pending_rank
                else:
                    pending_rank.run()          # (no-op when stage 1 carried it)
                pending_rank = None
            score = token_score.view(B, 1, h, w)
            # the strided mask slice and the minimum stage 2 already took go straight to the kernel.  The level's
            # INDICES are not needed before the merge at the end of the filtering (only its scores feed the next finer
            # level), so the rank launch of every level but the finest is deferred: the next level's stage 1 carries it
            ls, li = sel_score[:, offs[lvl]:offs[lvl] + ks[lvl]], sel_inds[:, offs[lvl]:offs[lvl] + ks[lvl]]
            if lvl > 0 and defer_ranks:
                pending_rank = plan_masked_topk(token_score, ks[lvl], mask, level_min[lvl:lvl + 1], start, (ls, li))
            else:
                masked_topk_desc(token_score, ks[lvl], mask=mask, fill_with_global_min=True, index_offset=start,
                                 fill_value=level_min[lvl:lvl + 1], out=(ls, li), carry_rank=late_rank,
                                 carry_finalize=finalize_job if hoisted is not None and FINALIZE_LEVEL == "merge" else None)
            if late_rank is not None:
                late_rank.run()             # (no-op when the merge carried it)
            salience_score[lvl], level_inds[lvl], level_score[lvl] = score, li, ls
            continue
        mask = mask.contiguous()
        if lvl != L - 1:
            up = F.interpolate(score, size=(h, w), mode="bilinear", align_corners=True)
            token_score = mask_predictor(level_memory, row_scale=up.reshape(B, h * w), alpha=alpha[lvl:lvl + 1])
        else:
            token_score = mask_predictor(level_memory)                  # [B, hw, 1]
        score = token_score.transpose(1, 2).reshape(B, -1, h, w)
        # masked_fill(mask, score.min()) + topk, fused in one launch; fp32 keys
        s32 = token_score.detach().squeeze(-1).float().contiguous()
        ls, li = masked_topk_desc(s32, int(level_token_nums[lvl]), mask=mask, fill_with_global_min=True,
                                  index_offset=start)
        salience_score[lvl] = score
        level_inds[lvl] = li
        level_score[lvl] = ls
    if extras is not None and fused:
        extras["level_min"] = level_min
        extras["selected"] = (sel_score, sel_inds)
        extras["segments"] = offs          # every column block is sorted descending, ties in index order
    return salience_score, level_inds, level_score


def salience_filtering(salience_score: Sequence[Tensor], level_inds: Sequence[Tensor], level_score: Sequence[Tensor],
                       mask_flatten: Tensor, layer_filter_ratio: Sequence[float], score_flat: Optional[Tensor] = None,
                       extras: Optional[dict] = None, lazy_foreground: bool = False):
    """Global sort, per-layer prefixes and foreground score (salience_transformer.py:156-168).

    Returns ``(foreground_inds: list[num_layers] of [B,Nq_k] int64, foreground_score [B,S])`` -- exactly the
    ``foreground_inds`` / ``foreground_score`` keyword arguments of ``SalienceTransformerEncoder.forward``.
    ``score_flat`` [B,S]: the already flattened ``salience_score`` (saves the concatenation); ``extras``: the
    by-products dict filled by ``level_filtering``.  ``lazy_foreground``: return ``foreground_score`` as a
    ``filter_ops.LazyForegroundScore`` when the by-products allow it (the encoder's entry gather then fills the masked
    rows itself: one launch less); ``.materialize()`` gives the tensor.
    """
    extras = extras or {}
    if "selected" in extras:
        selected_score, selected_inds = extras["selected"]
    else:
        selected_score = torch.cat(list(level_score), 1)
        selected_inds = torch.cat(list(level_inds), 1)
    n = selected_inds.shape[1]
    if "segments" in extras:
        # the per-level results are sorted already: the stable global sort is a 4-way merge
        _, sorted_inds = merge_sorted_desc(selected_score, selected_inds, extras["segments"])
    else:
        _, sorted_inds = masked_topk_desc(selected_score, n, payload=selected_inds, want_scores=False)
    counts = pyramid.layer_token_counts(n, layer_filter_ratio)
    # views of ONE sorted list: the encoder recognises the prefix structure and keeps the tokens in sorted order
    foreground_inds = [sorted_inds if c == n else sorted_inds[:, :c] for c in counts]
    fg = score_flat if score_flat is not None else pyramid.flatten_multi_level(salience_score).squeeze(-1)
    if "level_min" in extras and fg.is_cuda and fg.is_contiguous() and mask_flatten.is_contiguous():
        if lazy_foreground and fg.dtype == torch.float32:
            from .filter_ops import LazyForegroundScore
            fg = LazyForegroundScore(fg, mask_flatten, extras["level_min"])
        else:
            fg = masked_fill_min(fg, mask_flatten, extras["level_min"])   # fg.min() == min of the level minima
    else:
        fg = torch.where(mask_flatten, fg.min(), fg)
    return foreground_inds, fg
