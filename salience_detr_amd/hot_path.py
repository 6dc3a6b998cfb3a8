"""The Salience-DETR encoder hot path as one module: ``SalienceTransformer.forward`` from the
multi-level features down to ``memory`` (reference ``models/bricks/salience_transformer.py:97-183``).

``SalienceEncoderHotPath`` owns exactly the parameters of the reference ``SalienceTransformer`` that the
path touches, under the SAME state_dict keys (``level_embeds``, ``enc_output*``, ``alpha``,
``level_filter_ratio`` / ``layer_filter_ratio`` buffers, ``enc_mask_predictor.*``,
``encoder_class_head.*`` shared with ``encoder.enhance_mcsp.*``, ``encoder.*``), so a released
checkpoint loads with ``load_state_dict(strict=False)``; the neck, two-stage proposal head and decoder
(everything after ``memory``) are out of scope.
"""
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _hip, pyramid
from .salience_encoder import SalienceTransformerEncoder, SalienceTransformerEncoderLayer
from .salience_filtering import MaskPredictor, level_filtering, salience_filtering, token_budgets


def filter_ops_bf16x3() -> bool:
    from . import filter_ops
    return bool(filter_ops.salience_head_bf16x3)


def resolve_activation_dtype(dtype: torch.dtype, value_dtype: Optional[torch.dtype] = None):
    """The activation / value-map types a requested mode runs in.

    ``torch.bfloat16`` (BASELINE.json configs[1], the headline): bf16 activations; ``torch.float16`` (configs[4], the
    reference's ``--mixed-precision fp16``, main.py:24-56): **IEEE half activations** since round 5 -- the token-resident
    kernels built with ``-DSDETR_ACT_F16`` (``libsalience_hip_f16.so``: ``v_mfma_f32_*_f16`` at the bf16 rate, fp32
    accumulators / LayerNorm / softmax / class scores / sampling locations, stores saturating at +-65504).  Rounds 2-4
    served the fp16 request as bf16 activations + fp16 maps (a precision BELOW the one named); that substitution is gone.
    The head-major value maps of the fp16 mode are fp16 (consumed by ``v_fma_mix_f32`` without an unpack); the bf16 mode's
    default to bf16 unless the caller asks for fp16 maps (the benchmark does).
    """
    if dtype == torch.float16:
        return dtype, (value_dtype or torch.float16)
    return dtype, value_dtype


class SalienceEncoderHotPath(nn.Module):
    def __init__(self, encoder: SalienceTransformerEncoder, num_classes: int, num_feature_levels: int = 4,
                 level_filter_ratio: Tuple = (0.25, 0.5, 1.0, 1.0),
                 layer_filter_ratio: Tuple = (1.0, 0.8, 0.6, 0.6, 0.4, 0.2)):
        super().__init__()
        _hip.lib()  # fail loudly at construction if the HIP extension is not built
        self.embed_dim = encoder.embed_dim
        self.num_feature_levels = num_feature_levels
        self.num_classes = num_classes
        # DETRBaseTransformer / TwostageTransformer parts (base_transformer.py:11-72)
        self.level_embeds = nn.Parameter(torch.Tensor(num_feature_levels, self.embed_dim))
        self.enc_output = nn.Linear(self.embed_dim, self.embed_dim)
        self.enc_output_norm = nn.LayerNorm(self.embed_dim)
        # salience parameters (salience_transformer.py:68-70)
        self.register_buffer("level_filter_ratio", torch.Tensor(level_filter_ratio))
        self.register_buffer("layer_filter_ratio", torch.Tensor(layer_filter_ratio))
        self.alpha = nn.Parameter(torch.Tensor(3), requires_grad=True)
        self.encoder = encoder
        self.encoder_class_head = nn.Linear(self.embed_dim, num_classes)
        self.encoder.enhance_mcsp = self.encoder_class_head
        self.enc_mask_predictor = MaskPredictor(self.embed_dim, self.embed_dim)
        self._ratio_host = (tuple(float(r) for r in level_filter_ratio), tuple(float(r) for r in layer_filter_ratio))
        # the value projection rides in the stage-1 launches of the two coarsest levels (csrc/fused_head_value.hip):
        # same kernels, two launches' worth of an idle chip put to use
        self.fuse_value_projection = True
        self.init_weights()

    def init_weights(self):
        import math
        nn.init.normal_(self.level_embeds)
        nn.init.xavier_uniform_(self.enc_output.weight)
        nn.init.constant_(self.enc_output.bias, 0.0)
        nn.init.constant_(self.encoder_class_head.bias, -math.log((1 - 0.01) / 0.01))
        nn.init.uniform_(self.alpha, -0.3, 0.3)

    def set_encoder_dtype(self, dtype: torch.dtype, value_dtype: Optional[torch.dtype] = None):
        """Run the six encoder layers (and the shared class head) in ``dtype`` (bf16 for the inference
        benchmark).  The filtering stage -- token selection -- always stays fp32 so that the selected
        index sets do not depend on the encoder precision.  ``value_dtype`` is the storage type of the
        head-major value maps the MSDA kernel samples (default ``dtype``); ``torch.float16`` keeps 3 more
        mantissa bits than bf16 at the same size and lets the gather use ``v_fma_mix_f32`` (no unpack)."""
        dtype, value_dtype = resolve_activation_dtype(dtype, value_dtype)
        self.encoder.to(dtype)
        for layer in self.encoder.layers:
            layer.self_attn.value_dtype = value_dtype or dtype
        return self

    @property
    def encoder_dtype(self) -> torch.dtype:
        return self.encoder.layers[0].linear1.weight.dtype

    def _ratios(self):
        """Filter ratios as host floats (float32 values of the registered buffers; refreshed after a
        checkpoint load changed them -- one tiny D2H copy, then cached)."""
        key = (self.level_filter_ratio._version, self.layer_filter_ratio._version,
               self.level_filter_ratio.data_ptr())
        if getattr(self, "_ratio_key", None) != key:
            self._ratio_host = (tuple(self.level_filter_ratio.detach().cpu().tolist()),
                                tuple(self.layer_filter_ratio.detach().cpu().tolist()))
            self._ratio_key = key
        return self._ratio_host

    # layers of the batched value projection per carrier launch, in carrier order: stage 1 / stage 2 of the coarsest level,
    # of the next one, then stage 2 of the third-coarsest (level_filtering); pieces no launch carried run on their own
    value_projection_parts = (2, 1, 2, 1)
    # with the hoisted head (salience_filtering.HOIST_HEAD): the stage-2 launches of the two coarsest levels (same-box sweep,
    # benchmarks/hoist_sweep.sh: (3, 3) -26.7 us against the per-level form, (2, 2, 2) -19, (3, 2, 1) -15)
    value_projection_parts_hoisted = (3, 3)

    def forward(self, multi_level_feats: Sequence[Tensor], multi_level_masks: Sequence[Tensor],
                multi_level_pos_embeds: Sequence[Tensor],
                image_sizes: Optional[Sequence[Tuple[int, int]]] = None,
                canvas: Optional[Tuple[int, int]] = None, return_aux: bool = False):
        """``(memory [B,S,E], salience_score list[L] of [B,1,H_l,W_l])`` (+ aux dict).

        With ``image_sizes`` (+ the padded ``canvas`` size) the token budgets are computed on the host and
        the whole forward issues no device->host synchronisation; otherwise one sync reads them back.
        """
        level_ratio, layer_ratio = self._ratios()
        edt = self.encoder_dtype
        native = not (torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                   or any(f.requires_grad for f in multi_level_feats)))
        feat_enc = pos_enc = None
        if native and multi_level_feats[0].dtype == torch.float32:
            # F0 in one launch per level (flatten + level embedding + validity mask [+ bf16 copies])
            from .filter_ops import pyramid_flatten
            feat_flatten, lvl_pos_embed_flatten, enc_in, mask_flatten, feat_enc, pos_enc, valid_ratios_k = pyramid_flatten(
                multi_level_feats, multi_level_pos_embeds, multi_level_masks, self.level_embeds,
                want_bf16=(edt if _hip.is_act16(edt) else False), want_fp32=(return_aux or not _hip.is_act16(edt)))
        else:
            enc_in = valid_ratios_k = None
            feat_flatten = pyramid.flatten_multi_level(multi_level_feats)
            mask_flatten = pyramid.flatten_multi_level(multi_level_masks)
            lvl_pos_embed_flatten = pyramid.get_lvl_pos_embed(self.level_embeds.to(multi_level_pos_embeds[0].dtype),
                                                              multi_level_pos_embeds)
        # The value projection of all six layers depends only on the flattened features: the stage-1 / stage-2 launches
        # of the two coarsest levels carry it (a second stream / graph branch was measured slower: benchmarks/experiments)
        value_maps = None
        value_jobs = None
        if native and feat_enc is not None and self.fuse_value_projection:
            # four carriers (stage 1 and stage 2 of the two coarsest levels): two layers with each stage 1 (~20 us
            # of projection under ~20 us of head), one with each stage 2 (~10 under ~17)
            n_layers = len(self.encoder.layers)
            from . import salience_filtering as _sf
            hoisted = _sf.HOIST_HEAD and filter_ops_bf16x3()
            parts = ((self.value_projection_parts_hoisted if hoisted else self.value_projection_parts) if n_layers == 6
                     else min(3 if hoisted else 4, n_layers))
            plan = self.encoder.plan_values(feat_enc, mask_flatten, parts=parts,
                                            level_shapes=pyramid.level_shapes_of(multi_level_masks))
            if plan is not None:
                value_maps, value_jobs = plan
        level_shapes = pyramid.level_shapes_of(multi_level_masks)
        finalize_job = None
        if native and feat_enc is not None and self.fuse_value_projection:
            finalize_job = self.encoder.plan_finalize(feat_enc, mask_flatten, level_shapes)
        if valid_ratios_k is not None:
            spatial_shapes, level_start_index = pyramid.shape_tensors(level_shapes, mask_flatten.device)
            valid_ratios = valid_ratios_k
        else:
            spatial_shapes, level_start_index, valid_ratios = pyramid.multi_level_misc(multi_level_masks)
        starts = [0]
        for h, w in level_shapes[:-1]:
            starts.append(starts[-1] + h * w)

        # the no-grad path keeps enc_output + enc_output_norm inside the salience-head kernel
        fuse_enc = (enc_in is not None and self.enc_mask_predictor.fused_kernels_apply(enc_in)
                    and self.enc_output.weight.dtype == torch.float32)
        backbone_output_memory = None
        if fuse_enc:
            if return_aux:
                backbone_output_memory = torch.empty_like(enc_in)
        elif enc_in is not None:
            from .filter_ops import fused_layer_norm
            backbone_output_memory = fused_layer_norm(self.enc_output(enc_in), self.enc_output_norm)
        else:
            backbone_output_memory = pyramid.encoder_output_memory(
                self.enc_output, self.enc_output_norm, feat_flatten + lvl_pos_embed_flatten, mask_flatten,
                level_shapes)

        if image_sizes is not None:
            if canvas is None:
                raise ValueError("image_sizes needs the padded canvas size as well")
            def build():
                focus_host, level_host, _ = pyramid.host_token_budgets(image_sizes, canvas, level_shapes, level_ratio)
                return (torch.as_tensor(focus_host, dtype=torch.int64).to(mask_flatten.device),
                        [int(v) for v in level_host])
            focus_token_nums, level_token_nums = pyramid.static_tensor(
                ("budgets", tuple(map(tuple, image_sizes)), tuple(canvas), tuple(level_shapes), level_ratio,
                 str(mask_flatten.device)), build)
        else:
            focus_token_nums, level_dev, _ = token_budgets(multi_level_masks, self.level_filter_ratio.float())
            level_token_nums = level_dev.tolist()  # the stage's single host sync
            focus_token_nums = focus_token_nums.to(torch.int64)

        score_flat = None
        extras: dict = {}
        if fuse_enc:
            score_flat = torch.empty(mask_flatten.shape, dtype=torch.float32, device=mask_flatten.device)
            salience_score, level_inds, level_score = level_filtering(
                enc_in, mask_flatten, level_shapes, starts, level_token_nums, self.enc_mask_predictor, self.alpha,
                enc_output=self.enc_output, enc_output_norm=self.enc_output_norm, memory_out=backbone_output_memory,
                score_flat=score_flat, extras=extras, value_jobs=value_jobs, finalize_job=finalize_job)
        else:
            salience_score, level_inds, level_score = level_filtering(
                backbone_output_memory, mask_flatten, level_shapes, starts, level_token_nums, self.enc_mask_predictor,
                self.alpha)
        # (the masked fill of foreground_score happens in the encoder's entry gather unless the caller wants the tensor)
        foreground_inds, foreground_score = salience_filtering(salience_score, level_inds, level_score, mask_flatten,
                                                               layer_ratio, score_flat=score_flat, extras=extras,
                                                               lazy_foreground=native and not return_aux)
        if feat_enc is None:
            feat_enc, pos_enc = feat_flatten.to(edt), lvl_pos_embed_flatten.to(edt)
        for job in value_jobs or ():   # whatever no stage-1 launch carried
            job.run()
        memory = self.encoder(
            precomputed_value_maps=value_maps, query=feat_enc, query_pos=pos_enc, query_key_padding_mask=mask_flatten,
            spatial_shapes=spatial_shapes, level_start_index=level_start_index, valid_ratios=valid_ratios,
            foreground_score=foreground_score, focus_token_nums=focus_token_nums, foreground_inds=foreground_inds,
            multi_level_masks=multi_level_masks, finalize_job=finalize_job)
        if not return_aux:
            return memory, salience_score
        aux = dict(feat_flatten=feat_flatten, mask_flatten=mask_flatten, lvl_pos_embed_flatten=lvl_pos_embed_flatten,
                   spatial_shapes=spatial_shapes, level_start_index=level_start_index, valid_ratios=valid_ratios,
                   backbone_output_memory=backbone_output_memory, focus_token_nums=focus_token_nums,
                   level_token_nums=level_token_nums, level_inds=level_inds, level_score=level_score,
                   foreground_inds=foreground_inds, foreground_score=foreground_score)
        return memory, salience_score, aux


def build_hot_path(embed_dim=256, num_heads=8, d_ffn=2048, num_layers=6, num_classes=91, num_levels=4, num_points=4,
                   topk_sa=300, max_num_embedding=200, level_filter_ratio=(0.4, 0.8, 1.0, 1.0),
                   layer_filter_ratio=(1.0, 0.8, 0.6, 0.6, 0.4, 0.2)) -> SalienceEncoderHotPath:
    """The configuration of ``configs/salience_detr/salience_detr_resnet50_800_1333.py:22-82`` by default."""
    layer = SalienceTransformerEncoderLayer(embed_dim=embed_dim, d_ffn=d_ffn, dropout=0.0, n_heads=num_heads,
                                            activation=nn.ReLU(inplace=True), n_levels=num_levels,
                                            n_points=num_points, topk_sa=topk_sa)
    encoder = SalienceTransformerEncoder(layer, num_layers=num_layers, max_num_embedding=max_num_embedding)
    return SalienceEncoderHotPath(encoder, num_classes=num_classes, num_feature_levels=num_levels,
                                  level_filter_ratio=level_filter_ratio, layer_filter_ratio=layer_filter_ratio)
