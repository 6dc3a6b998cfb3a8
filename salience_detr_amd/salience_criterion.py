"""``SalienceCriterion`` on MI355X: the supervision of the salience head --
scale-independent salience targets from the ground-truth boxes and a sigmoid focal loss on the predicted maps
(reference ``models/detectors/salience_detr.py:13-116``, ``models/bricks/losses.py:4-13``).

Same constructor arguments, ``forward`` signature and return value as the reference.  Three launches instead of ~25 per
(level, image): the targets of all levels and images, the loss (deterministic two-stage reduction, ``num_pos`` included)
and, in backward, the gradient with respect to the logits.  ``noise_scale`` > 0 draws one ``[B,S]`` uniform tensor from
torch's generator (the reference draws per level and image: same distribution, different stream).
"""
import ctypes
from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _hip


def _host_array(ctype, values):
    flat = [v for row in values for v in row]
    return (ctype * len(flat))(*flat)


def salience_targets(boxes_xyxy: Tensor, box_offset: Tensor, level_shapes: Sequence[Tuple[int, int]],
                     feature_strides: Sequence[Tuple[float, float]], limit_range: Sequence[Tuple[float, float]],
                     noise_scale: float = 0.0, noise: Tensor = None) -> Tensor:
    """``mask_targets`` ``[B,S]`` of ``SalienceCriterion.forward`` (:36-46): ``boxes_xyxy`` ``[sum m,4]`` fp32 in input-image
    pixels, ``box_offset`` int32 ``[B+1]`` (device), per level host-side shapes (h, w), strides (sy, sx), ranges (lo, hi)."""
    _hip.require_device("salience_targets", boxes_xyxy=boxes_xyxy, box_offset=box_offset, noise=noise)
    if boxes_xyxy.dtype != torch.float32 or box_offset.dtype != torch.int32 or len(limit_range) < len(level_shapes):
        raise RuntimeError("salience_targets: fp32 boxes, int32 offsets and a limit range per level expected")
    B = box_offset.numel() - 1
    S = sum(h * w for h, w in level_shapes)
    boxes = boxes_xyxy.contiguous()
    target = torch.empty((B, S), dtype=torch.float32, device=box_offset.device)
    if noise_scale and (noise is None or tuple(noise.shape) != (B, S) or noise.dtype != torch.float32):
        raise RuntimeError("salience_targets: noise_scale needs an fp32 noise tensor [B,S]")
    with torch.cuda.device(target.device):
        code = _hip.lib().sdetr_salience_targets(
            _hip.stream_ptr(), boxes.data_ptr(), box_offset.data_ptr(), B,
            _host_array(ctypes.c_int64, [(int(h), int(w)) for h, w in level_shapes]),
            _host_array(ctypes.c_float, [(float(a), float(b)) for a, b in feature_strides]),
            _host_array(ctypes.c_float, [(float(a), float(b)) for a, b in limit_range[:len(level_shapes)]]),
            len(level_shapes), float(noise_scale), _hip.ptr(noise.contiguous() if noise_scale else None), target.data_ptr())
    _hip.check(code, "salience_targets")
    return target


class _FocalLoss(torch.autograd.Function):
    """sum_i focal(logit_i, target_i) / max(#positives, 1) with the reference's gradient (the weight is not detached)."""

    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, alpha: float, gamma: float, positive_threshold: float):
        _hip.require_device("salience_focal_loss", logits=logits, target=target)
        x = logits.detach().float().contiguous()
        t = target.detach().float().contiguous()
        lib = _hip.lib()
        n = x.numel()
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        ws_bytes = max(int(lib.sdetr_focal_loss_workspace_bytes(n)), 8)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            code = lib.sdetr_salience_focal_loss(_hip.stream_ptr(), x.data_ptr(), t.data_ptr(), n, float(alpha), float(gamma),
                                                 float(positive_threshold), ws.data_ptr(), ws_bytes, out.data_ptr())
        _hip.check(code, "salience_focal_loss")
        ctx.save_for_backward(x, t, out)
        ctx.alpha, ctx.gamma, ctx.in_dtype = float(alpha), float(gamma), logits.dtype
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss: Tensor):
        x, t, out = ctx.saved_tensors
        grad = torch.empty_like(x)
        g = grad_loss.detach().float().reshape(1).contiguous()
        with torch.cuda.device(x.device):
            code = _hip.lib().sdetr_salience_focal_loss_backward(_hip.stream_ptr(), x.data_ptr(), t.data_ptr(), x.numel(),
                                                                 ctx.alpha, ctx.gamma, out.data_ptr(), g.data_ptr(),
                                                                 grad.data_ptr())
        _hip.check(code, "salience_focal_loss_backward")
        return grad.to(ctx.in_dtype), None, None, None, None


class SalienceCriterion(nn.Module):
    def __init__(self, limit_range: Tuple = ((-1, 64), (64, 128), (128, 256), (256, 99999)), noise_scale: float = 0.0,
                 alpha: float = 0.25, gamma: float = 2.0):
        super().__init__()
        self.limit_range = limit_range
        self.noise_scale = noise_scale
        self.alpha = alpha
        self.gamma = gamma

    @staticmethod
    def stage_boxes(targets: List[Dict[str, Tensor]], image_sizes, device):
        """Host side of the supervision: the ground-truth boxes of the batch as one device tensor of absolute
        ``(x0, y0, x1, y1)`` plus the per-image offsets -- input staging (host lists, host -> device copies), done by
        the data loader's side of a training loop; ``mask_targets`` / ``forward`` take the result as ``staged``."""
        xyxy, counts = [], []
        for t, (img_h, img_w) in zip(targets, image_sizes):
            b = t["boxes"].to(device=device, dtype=torch.float32)
            cx, cy, w, h = b.unbind(-1)
            scale = torch.tensor([img_w, img_h, img_w, img_h], dtype=torch.float32, device=device)
            xyxy.append(torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1) * scale)
            counts.append(int(b.shape[0]))
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        boxes = torch.cat(xyxy, 0) if offs[-1] else torch.zeros((1, 4), dtype=torch.float32, device=device)
        box_offset = torch.tensor(offs, dtype=torch.int32).to(device)
        return boxes, box_offset

    def mask_targets(self, targets: List[Dict[str, Tensor]], level_shapes, feature_strides, image_sizes, device,
                     staged=None) -> Tensor:
        """The supervision maps ``[B,S]`` (levels flattened back to back) for ``targets[i]["boxes"]`` (cx, cy, w, h in
        [0,1] of image i, as in the reference's datasets)."""
        boxes, box_offset = staged if staged is not None else self.stage_boxes(targets, image_sizes, device)
        S = sum(h * w for h, w in level_shapes)
        noise = torch.rand((len(targets), S), dtype=torch.float32, device=device) if self.noise_scale else None
        return salience_targets(boxes, box_offset, level_shapes, feature_strides, self.limit_range, self.noise_scale, noise)

    def forward(self, foreground_mask: Sequence[Tensor], targets, feature_strides, image_sizes, staged=None):
        """Reference signature (:27): ``foreground_mask`` = the salience maps ``[B,1,H_l,W_l]`` per level.  ``staged``
        = ``stage_boxes(...)`` of the same targets when the caller has the boxes on the device already (a step captured
        in a hipGraph: no host -> device copies inside); the target maps themselves are still built here."""
        if not foreground_mask[0].is_cuda:
            raise RuntimeError("SalienceCriterion: HIP device tensors required; there is no CPU fallback")
        level_shapes = [tuple(m.shape[-2:]) for m in foreground_mask]
        target = self.mask_targets(targets, level_shapes, feature_strides, image_sizes, foreground_mask[0].device,
                                   staged=staged)
        logits = torch.cat([m.flatten(-2) for m in foreground_mask], -1).squeeze(1)
        loss = _FocalLoss.apply(logits, target, self.alpha, self.gamma, 0.5 * self.noise_scale)
        return {"loss_salience": loss}
