"""``SalienceTransformer``: the encoder hot path plus the two-stage proposal selection (row N1) and the decoder
(row N2) -- the whole reference ``SalienceTransformer.forward`` for inference
(``models/bricks/salience_transformer.py:51-226``) under the reference's constructor arguments, parameter names and
return values.

The stages after ``memory`` on the no-grad path:

* ``gen_encoder_output_proposals`` (base_transformer.py:74-112): one launch for the keep flags + proposal logits
  (geometry only depends on the masks), ``enc_output`` + ``enc_output_norm`` through the fused kernels;
* ``encoder_class_head(...).max(-1)`` without materialising the ``[B,S,num_classes]`` logits, top-``4*num_proposals``
  by the rank-by-counting kernel of the filtering stage (ties: lower token index first);
* ``nms_on_topk_index`` (:249-295) as grid-neighbour suppression in one workgroup per image -- no torchvision, no
  per-image python loop; one host read-back of the kept counts (the reference has the same data-dependent shape);
* class / box heads only for the surviving tokens (gathers instead of ``[B,S,*]`` GEMMs), the box sigmoid fused with
  the proposal-logit gather.

Denoising queries (training) are accepted and concatenated as in the reference; the neck (row N3) is not built: a
module passed as ``neck`` is applied with plain torch ops exactly where the reference applies it.
"""
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import pyramid
from .filter_ops import (class_head_max_times, encoder_output_proposals, fused_layer_norm, gather_rows,
                         grid_nms_topk, masked_topk_desc, proposal_refine, token_linear, token_linear_applies)
from .hot_path import SalienceEncoderHotPath
from .salience_decoder import MLP, SalienceTransformerDecoder, SalienceTransformerDecoderLayer
from .salience_encoder import SalienceTransformerEncoder, SalienceTransformerEncoderLayer


class SalienceTransformer(SalienceEncoderHotPath):
    def __init__(self, encoder: nn.Module, neck: Optional[nn.Module], decoder: nn.Module, num_classes: int,
                 num_feature_levels: int = 4, two_stage_num_proposals: int = 900,
                 level_filter_ratio: Tuple = (0.25, 0.5, 1.0, 1.0),
                 layer_filter_ratio: Tuple = (1.0, 0.8, 0.6, 0.6, 0.4, 0.2)):
        super().__init__(encoder, num_classes, num_feature_levels, level_filter_ratio, layer_filter_ratio)
        self.two_stage_num_proposals = two_stage_num_proposals
        self.neck = neck
        self.decoder = decoder
        self.tgt_embed = nn.Embedding(two_stage_num_proposals, self.embed_dim)
        self.encoder_bbox_head = MLP(self.embed_dim, self.embed_dim, 4, 3)
        self.nms_iou_threshold = 0.3
        self.last_proposal_index = None
        # The reference truncates the proposals to the smallest per-image survivor count (:286-295), a data-dependent
        # shape that costs a device->host read-back -- and then fails in the decoder unless that count reaches
        # two_stage_num_proposals (tgt_embed has exactly that many rows).  With static_proposals the count is not read
        # back (hipGraph-capturable); images with fewer survivors get token 0 as filler instead of an exception.
        self.static_proposals = False
        nn.init.normal_(self.tgt_embed.weight)
        nn.init.constant_(self.encoder_bbox_head.layers[-1].weight, 0.0)
        nn.init.constant_(self.encoder_bbox_head.layers[-1].bias, 0.0)

    def set_dtype(self, dtype: torch.dtype, value_dtype: Optional[torch.dtype] = None):
        """Encoder, proposal heads and decoder in ``dtype`` (the salience filtering stays fp32, see
        ``set_encoder_dtype``)."""
        from .hot_path import resolve_activation_dtype
        dtype, value_dtype = resolve_activation_dtype(dtype, value_dtype)   # fp16 request -> bf16 activations, fp16 maps
        self.set_encoder_dtype(dtype, value_dtype)
        self.decoder.to(dtype)
        self.tgt_embed.to(dtype)
        self.encoder_bbox_head.to(dtype)
        for layer in self.decoder.layers:
            layer.cross_attn.value_dtype = value_dtype or dtype
        return self

    # ------------------------------------------------------------------------------------------ row N1
    def gen_encoder_output_proposals(self, memory: Tensor, memory_padding_mask: Tensor, spatial_shapes):
        """Reference signature (base_transformer.py:74): ``(output_memory [B,S,E], output_proposals [B,S,4])``."""
        level_shapes = spatial_shapes if isinstance(spatial_shapes, (list, tuple)) else spatial_shapes.tolist()
        keep, logit = encoder_output_proposals(memory_padding_mask, level_shapes)
        return self._output_memory(memory, keep), logit

    def _output_memory(self, memory: Tensor, keep: Tensor) -> Tensor:
        x = memory * keep.unsqueeze(-1).to(memory.dtype)
        w, b = self.enc_output.weight, self.enc_output.bias
        if w.dtype != x.dtype:
            # enc_output stays fp32 for the filtering stage (set_dtype); the proposal stage reads a cached copy in the
            # activation dtype, refreshed when the parameters change
            tag = (w.data_ptr(), w._version, b.data_ptr(), b._version, x.dtype)
            hit = self.__dict__.get("_enc_output_cast")
            if hit is None or hit[0] != tag:
                hit = (tag, w.detach().to(x.dtype), b.detach().to(x.dtype))
                self.__dict__["_enc_output_cast"] = hit
            w, b = hit[1], hit[2]
        if token_linear_applies(x, w):
            y = token_linear(x, w, b)
        else:
            y = torch.nn.functional.linear(x, w, b)
        return fused_layer_norm(y, self.enc_output_norm)

    def nms_on_topk_index(self, topk_scores, topk_index, spatial_shapes, level_start_index, iou_threshold=0.3):
        """Reference signature (:249-251).  ``topk_index`` must be in descending score order (it comes from topk)."""
        level_shapes = spatial_shapes if isinstance(spatial_shapes, (list, tuple)) else spatial_shapes.tolist()
        S = sum(h * w for h, w in level_shapes)
        kept, count = grid_nms_topk(topk_index, level_shapes, S, iou_threshold, self.two_stage_num_proposals)
        if self.static_proposals:
            return kept
        n = min(int(count.min()), self.two_stage_num_proposals)       # the stage's host sync (reference: :286-294)
        return kept[:, :n]

    def select_proposals(self, memory: Tensor, mask_flatten: Tensor, level_shapes):
        """salience_transformer.py:194-212 -> (enc_outputs_class [B,n,C], enc_outputs_coord [B,n,4] fp32)."""
        keep, logit = encoder_output_proposals(mask_flatten, level_shapes)
        output_memory = self._output_memory(memory, keep)
        head = self.encoder_class_head
        B, S, _ = output_memory.shape
        if token_linear_applies(output_memory, head.weight):
            ones = pyramid.static_tensor(("ones", B, S, str(memory.device)),
                                         lambda: torch.ones((B, S), dtype=torch.float32, device=memory.device))
            best = class_head_max_times(output_memory, head, ones)
        else:
            best = head(output_memory).float().max(-1)[0]
        k = min(self.two_stage_num_proposals * 4, S)
        topk_scores, topk_index = masked_topk_desc(best.contiguous(), k)
        index = self.nms_on_topk_index(topk_scores, topk_index, level_shapes, None, self.nms_iou_threshold)
        self.last_proposal_index = index          # [B,n] token ids, for inspection / tests
        selected = gather_rows(output_memory, index.contiguous())
        enc_outputs_class = head(selected)
        enc_outputs_coord = proposal_refine(self.encoder_bbox_head(selected), logit, index.contiguous())
        return enc_outputs_class, enc_outputs_coord

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, multi_level_feats: Sequence[Tensor], multi_level_masks: Sequence[Tensor],
                multi_level_pos_embeds: Sequence[Tensor], noised_label_query=None, noised_box_query=None,
                attn_mask=None, image_sizes=None, canvas=None):
        """Reference signature and return values (:97-226): ``(outputs_classes [Ld,B,Nq,C], outputs_coords
        [Ld,B,Nq,4], enc_outputs_class, enc_outputs_coord, salience_score)``.  ``image_sizes`` / ``canvas`` as in
        ``SalienceEncoderHotPath.forward`` (host-side token budgets, no sync in the filtering stage)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("SalienceTransformer.forward is INFERENCE-ONLY: call it under torch.no_grad() (model.eval() "
                               "alone leaves grad mode on).  The proposal stage (top-k + NMS on the device) has no "
                               "autograd path; for training use SalienceEncoderHotPath (encoder hot path) and "
                               "SalienceTransformerDecoder, which have autograd paths of their own.")
        memory, salience_score, aux = SalienceEncoderHotPath.forward(
            self, multi_level_feats, multi_level_masks, multi_level_pos_embeds, image_sizes=image_sizes, canvas=canvas,
            return_aux=True)
        mask_flatten, spatial_shapes = aux["mask_flatten"], aux["spatial_shapes"]
        level_shapes = pyramid.level_shapes_of(multi_level_masks)
        if self.neck is not None and hasattr(self.neck, "forward_memory"):
            # row N3 (salience_neck.py): token-major in, token-major out -- no NCHW round trip
            memory = self.neck.forward_memory(memory, level_shapes)
        elif self.neck is not None:  # a foreign neck with the reference's NCHW interface (:185-192)
            B = memory.shape[0]
            feats, cur = {}, 0
            for i, (h, w) in enumerate(level_shapes):
                feats[i] = memory[:, cur:cur + h * w].transpose(1, 2).contiguous().reshape(B, self.embed_dim, h, w)
                cur += h * w
            memory = torch.cat([f.flatten(2).transpose(1, 2) for f in self.neck(feats).values()], dim=1)
        enc_outputs_class, enc_outputs_coord = self.select_proposals(memory, mask_flatten, level_shapes)
        reference_points = enc_outputs_coord.detach()
        target = self.tgt_embed.weight.expand(memory.shape[0], -1, -1)
        if noised_label_query is not None and noised_box_query is not None:
            target = torch.cat([noised_label_query.to(target.dtype), target], 1)
            reference_points = torch.cat([noised_box_query.sigmoid().float(), reference_points], 1)
        outputs_classes, outputs_coords = self.decoder(
            query=target, value=memory, key_padding_mask=mask_flatten, reference_points=reference_points,
            spatial_shapes=spatial_shapes, level_start_index=aux["level_start_index"],
            valid_ratios=aux["valid_ratios"], attn_mask=attn_mask)
        return outputs_classes, outputs_coords, enc_outputs_class, enc_outputs_coord, salience_score


def build_salience_transformer(embed_dim=256, num_heads=8, d_ffn=2048, num_encoder_layers=6, num_decoder_layers=6,
                               num_classes=91, num_levels=4, num_points=4, topk_sa=300, max_num_embedding=200,
                               two_stage_num_proposals=900, level_filter_ratio=(0.4, 0.8, 1.0, 1.0),
                               layer_filter_ratio=(1.0, 0.8, 0.6, 0.6, 0.4, 0.2),
                               with_neck: bool = False, neck_groups: int = 4) -> SalienceTransformer:
    """The transformer of ``configs/salience_detr/salience_detr_resnet50_800_1333.py:22-103``; ``with_neck=True`` adds
    its RepVGGPluX neck (:57-63, row N3; eval mode only), the default leaves it out."""
    enc_layer = SalienceTransformerEncoderLayer(embed_dim=embed_dim, d_ffn=d_ffn, dropout=0.0, n_heads=num_heads,
                                                activation=nn.ReLU(inplace=True), n_levels=num_levels,
                                                n_points=num_points, topk_sa=topk_sa)
    encoder = SalienceTransformerEncoder(enc_layer, num_layers=num_encoder_layers, max_num_embedding=max_num_embedding)
    dec_layer = SalienceTransformerDecoderLayer(embed_dim=embed_dim, d_ffn=d_ffn, n_heads=num_heads, dropout=0.0,
                                                activation=nn.ReLU(inplace=True), n_levels=num_levels,
                                                n_points=num_points)
    decoder = SalienceTransformerDecoder(dec_layer, num_decoder_layers, num_classes)
    neck = None
    if with_neck:
        from .salience_neck import build_neck
        neck = build_neck(embed_dim, num_levels, neck_groups)
    return SalienceTransformer(encoder, neck, decoder, num_classes, num_levels, two_stage_num_proposals,
                               level_filter_ratio, layer_filter_ratio)
