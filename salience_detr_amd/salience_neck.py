"""Row N3 of SURVEY.md section 8(f): the RepVGGPluX neck that every reference config puts between the encoder and the
proposal stage (``models/necks/repnet.py:12-245``, called at ``models/bricks/salience_transformer.py:185-192``).

Same class names, constructor arguments and state-dict keys as the reference, so its checkpoints load unchanged:
``lateral_convs.{i}.{0,1}.*``, ``layer_blocks.{i}.{conv1,conv2}.{0,1}.*``,
``layer_blocks.{i}.bottlenecks.{j}.{conv1,conv2}.{0,1}.*``, ``...bottlenecks.{j}.se_module.{conv_mask,se_module.0,
se_module.2}.*``, ``downsample_blocks.{i}.{0,1}.*``, ``pan_blocks.{i}....``.  The ``nn.Conv2d`` / ``nn.BatchNorm2d``
objects below only HOLD those parameters; nothing here calls a library convolution.

How it runs (eval mode):
  * feature maps stay token-major ``[B, H*W, C]`` -- the encoder's memory IS channels-last, the reference's two
    transposes per level disappear (``forward_memory``);
  * every BatchNorm folds into the convolution before it; ``conv1(x) + alpha * conv2(x)`` of a RepVGG block
    (3x3 + 1x1, each with its norm, repnet.py:61-62) folds into ONE grouped 3x3 kernel with a bias;
  * a 1x1 convolution is a GEMM on the token rows (library GEMM) + ``sdetr_neck_combine`` (bias, SiLU); it commutes
    with nearest up-sampling, so the coarse half of ``cat([upsample(high), low])`` is multiplied at the COARSE
    resolution (4x fewer rows) and up-sampled inside the combine kernel; ``conv1`` and ``conv2`` of a CSP layer read
    the same input and run as one GEMM of twice the width;
  * 3x3 convolutions (fp32 LDS-tiled kernel; bf16 maps with the real channel counts: the MFMA kernel), the
    attention-pooled gate and the shortcuts are the HIP kernels of ``include/salience_hip.h`` (13).
Training mode (``.train()``): the differentiable form -- NCHW convolutions on the parameter holders, every BatchNorm on
BATCH statistics taken over all ranks' pixels (``data_parallel.sync_batch_norm_train``: one all-reduce per norm and
direction where the reference's ``nn.SyncBatchNorm`` issues an all_gather + all_reduce pair, ``main.py:126-127``),
running statistics updated as ``nn.BatchNorm2d`` does.  It is the autograd path of this row (device tensors only, like
every other operator here); the HIP kernels above are the inference path.
"""
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import filter_ops as FO


class Conv2dNormActivation(nn.Sequential):
    """Parameter holder with the reference's layout (``models/bricks/misc.py:61-158``): ``0`` = convolution
    (bias-free when a norm follows), ``1`` = norm, then the activation."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, padding=None,
                 groups: int = 1, norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, inplace: bool = True):
        if padding is None:
            padding = (kernel_size - 1) // 2
        layers: List[nn.Module] = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, groups=groups,
                                             bias=norm_layer is None)]
        if norm_layer is not None:
            layers.append(norm_layer(out_channels))
        if activation_layer is not None:
            layers.append(activation_layer(inplace=inplace))
        super().__init__(*layers)

    def folded(self) -> Tuple[Tensor, Tensor]:
        """(weight [out, in / groups, k, k], bias [out]) of conv + norm on the running statistics, fp32."""
        conv = self[0]
        w = conv.weight.detach().float()
        b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
        if len(self) > 1 and isinstance(self[1], nn.modules.batchnorm._BatchNorm):
            bn = self[1]
            scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
            w = w * scale.view(-1, 1, 1, 1)
            b = (b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
        return w, b


class SqueezeAndExcitation(nn.Module):
    """Parameter holder of ``models/bricks/basic.py:29-41``: ``conv_mask`` (C -> 1 attention-pooling logits) and the
    bias-free C -> C/16 -> C gate."""

    def __init__(self, channels: int, reduction: int = 16):
        super().__init__()
        self.conv_mask = nn.Conv2d(channels, 1, kernel_size=1)
        self.se_module = nn.Sequential(
            nn.Conv2d(channels, channels // reduction, kernel_size=1, bias=False), nn.ReLU(inplace=True),
            nn.Conv2d(channels // reduction, channels, kernel_size=1, bias=False), nn.Sigmoid())


class RepVggPluXBlock(nn.Module):
    """``models/necks/repnet.py:12-64``: grouped 3x3 + grouped 1x1 (own BatchNorms), activation, gate, shortcut."""

    def __init__(self, in_channels: int, out_channels: int, activation_layer=nn.ReLU, inplace: bool = True,
                 groups: int = 4, alpha: bool = False):
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("RepVggPluXBlock: built for in_channels == out_channels (every reference config)")
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.activation = activation_layer(inplace=True)
        self.conv1 = Conv2dNormActivation(in_channels, out_channels, 3, 1, 1, groups=groups, activation_layer=None)
        self.conv2 = Conv2dNormActivation(in_channels, out_channels, 1, 1, 0, groups=groups, activation_layer=None)
        self.alpha = nn.Parameter(torch.tensor(1.0)) if alpha else 1.0
        self.se_module = SqueezeAndExcitation(out_channels)
        self.identity = nn.Identity()

    def folded(self, dtype=torch.float32) -> Dict[str, Tensor]:
        w3, b3 = self.conv1.folded()
        w1, b1 = self.conv2.folded()
        alpha = float(self.alpha)
        w = w3.clone()
        w[:, :, 1, 1] += alpha * w1[:, :, 0, 0]
        G, C = self.groups, self.out_channels
        # [out, in/G, 3, 3] -> [G, 3, 3, in/G, out/G]
        packed = w.view(G, C // G, C // G, 3, 3).permute(0, 3, 4, 2, 1).contiguous()
        se = self.se_module
        R = se.se_module[0].weight.shape[0]
        return dict(weight=packed, bias=(b3 + alpha * b1).contiguous(),
                    mfma=FO.neck_pack_conv3x3(packed, dtype) if dtype in (torch.bfloat16, torch.float16) and packed.is_cuda else None,
                    mask=se.conv_mask.weight.detach().float().reshape(C).contiguous(),
                    squeeze=se.se_module[0].weight.detach().float().reshape(R, C).contiguous(),
                    excite=se.se_module[2].weight.detach().float().reshape(C, R).contiguous())


class CSPRepPluXLayer(nn.Module):
    """``models/necks/repnet.py:67-123``."""

    def __init__(self, in_channels: int, out_channels: int, num_blocks: int = 3, expansion: float = 1.0,
                 groups: int = 4, norm_layer=nn.BatchNorm2d, activation_layer=nn.SiLU):
        super().__init__()
        hidden = int(out_channels * expansion)
        if hidden != out_channels:
            raise ValueError("CSPRepPluXLayer: built for expansion = 1 (every reference config)")
        self.conv1 = Conv2dNormActivation(in_channels, hidden, 1, 1, norm_layer=norm_layer,
                                          activation_layer=activation_layer)
        self.conv2 = Conv2dNormActivation(in_channels, hidden, 1, 1, norm_layer=norm_layer,
                                          activation_layer=activation_layer)
        self.bottlenecks = nn.Sequential(*[RepVggPluXBlock(hidden, hidden, groups=groups,
                                                           activation_layer=activation_layer)
                                           for _ in range(num_blocks)])
        self.conv3 = nn.Identity()

    def folded(self, dtype) -> Dict[str, object]:
        w1, b1 = self.conv1.folded()
        w2, b2 = self.conv2.folded()
        w = torch.cat([w1[:, :, 0, 0], w2[:, :, 0, 0]], 0)  # [2 * hidden, in]: conv1's outputs, then conv2's
        half = w.shape[1] // 2
        return dict(first=w[:, :half].to(dtype).contiguous(), second=w[:, half:].to(dtype).contiguous(),
                    bias=torch.cat([b1, b2]).contiguous(), blocks=[blk.folded(dtype) for blk in self.bottlenecks])


def _token_matmul(x: Tensor, weight: Tensor) -> Tensor:
    """``x @ weight.T`` for token-major ``[B, N, C]`` maps; a row range of a longer buffer (batch stride > N * C) goes
    through the strided-batched GEMM instead of being copied first."""
    if x.is_contiguous() or x.dim() != 3 or x.stride(2) != 1 or x.stride(1) != x.shape[2]:
        return torch.matmul(x, weight.t())
    return torch.bmm(x, weight.t().expand(x.shape[0], -1, -1))


def _is_silu(act) -> bool:
    return act is nn.SiLU or isinstance(act, nn.SiLU)


class RepVGGPluXNetwork(nn.Module):
    """``models/necks/repnet.py:125-245``.  ``forward`` keeps the reference's interface (ordered dict of NCHW maps in,
    same keys out); ``forward_memory`` is the token-major entry the transformer uses."""

    def __init__(self, in_channels_list: List[int], out_channels_list: List[int], groups: int = 4,
                 norm_layer=nn.BatchNorm2d, activation=nn.SiLU, extra_block: bool = False):
        super().__init__()
        if any(c == 0 for c in in_channels_list):
            raise ValueError("in_channels=0 is currently not supported")
        if len(set(in_channels_list) | set(out_channels_list)) != 1:
            raise ValueError("RepVGGPluXNetwork: built for one channel count on all levels (every reference config)")
        if not _is_silu(activation):
            raise ValueError("RepVGGPluXNetwork: the kernels implement SiLU (every reference config)")
        C = out_channels_list[0]
        if C % (4 * groups) or C > 256 or C < 16:
            raise ValueError("RepVGGPluXNetwork: channels must be a multiple of 4 * groups, 16 <= channels <= 256")
        self.channels, self.groups = C, groups
        n = len(out_channels_list)
        self.lateral_convs = nn.ModuleList(Conv2dNormActivation(C, C, 1, 1, norm_layer=norm_layer,
                                                                activation_layer=activation) for _ in range(1, n))
        self.layer_blocks = nn.ModuleList(CSPRepPluXLayer(2 * C, C, groups=groups, norm_layer=norm_layer,
                                                          activation_layer=activation) for _ in range(1, n))
        self.downsample_blocks = nn.ModuleList(Conv2dNormActivation(C, C, 3, 2, 1, norm_layer=norm_layer,
                                                                    activation_layer=activation) for _ in range(n - 1))
        self.pan_blocks = nn.ModuleList(CSPRepPluXLayer(2 * C, C, groups=groups, norm_layer=norm_layer,
                                                        activation_layer=activation) for _ in range(n - 1))
        self.extra_block = extra_block
        self.process_group = None   # ranks whose pixels share the batch statistics in training mode (None = all)
        self._plan = None
        self.init_weights()

    def init_weights(self):
        """repnet.py:202-208: every convolution kaiming-uniform (a = 1), biases zero."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ---- folded parameters ------------------------------------------------------------------------------------------
    def _folded(self, dtype) -> Dict[str, object]:
        tensors = list(self.parameters()) + list(self.buffers())
        tag = (dtype, tuple((t.data_ptr(), t._version) for t in tensors))
        if self._plan is not None and self._plan[0] == tag:
            return self._plan[1]
        plan: Dict[str, object] = dict(lateral=[], layer=[], down=[], pan=[])
        for m in self.lateral_convs:
            w, b = m.folded()
            plan["lateral"].append((w[:, :, 0, 0].to(dtype).contiguous(), b.contiguous()))
        for m in self.layer_blocks:
            plan["layer"].append(m.folded(dtype))
        for m in self.downsample_blocks:
            w, b = m.folded()  # dense [out, in, 3, 3] -> [1, 3, 3, in, out]
            wd = w.permute(2, 3, 1, 0).contiguous().unsqueeze(0)
            plan["down"].append((wd, b.contiguous(),
                                 FO.neck_pack_conv3x3(wd, dtype) if dtype in (torch.bfloat16, torch.float16) and wd.is_cuda else None))
        for m in self.pan_blocks:
            plan["pan"].append(m.folded(dtype))
        self._plan = (tag, plan)
        return plan

    # ---- token-major forward ----------------------------------------------------------------------------------------
    @staticmethod
    def _csp(p, first: Tensor, first_hw, second: Tensor, hw, upsample: bool) -> Tensor:
        """CSPRepPluXLayer on ``cat([first (up-sampled to hw when asked), second], channels)`` (repnet.py:120-123)."""
        h, w = hw
        C = second.shape[2]
        a_second = _token_matmul(second, p["second"])  # [B, N, 2C]: conv1 | conv2 outputs
        if upsample:
            a_first = torch.matmul(first, p["first"].t())  # at the coarse resolution
            both = FO.neck_combine(a_second, h, w, up=a_first, up_hw=first_hw, bias=p["bias"], activation=True)
        else:
            # (in place: the out-of-place form copies its first operand into the result before the GEMM)
            a_second = a_second.baddbmm_(first, p["first"].t().expand(first.shape[0], -1, -1))
            both = FO.neck_combine(a_second, h, w, bias=p["bias"], activation=True)
        x, branch = both[:, :, :C], both[:, :, C:]
        last = len(p["blocks"]) - 1
        for j, blk in enumerate(p["blocks"]):
            y = FO.neck_conv3x3(x, h, w, blk["weight"], blk["bias"], stride=1, activation=True, packed=blk["mfma"])
            x = FO.neck_gate_shortcut(y, blk["mask"], blk["squeeze"], blk["excite"], shortcut=x,
                                      shortcut2=branch if j == last else None)
        if last < 0:
            x = x + branch
        return x

    def forward_levels(self, levels: Sequence[Tensor], shapes: Sequence[Tuple[int, int]],
                       differentiable: bool = False) -> List[Tensor]:
        """Token-major levels ``[B, h*w, C]`` (fine to coarse) -> the same (repnet.py:211-245).  ``differentiable``: take
        the torch-op form also in eval mode (running statistics in the norms) -- the fused kernels have no backward."""
        if len(levels) != len(self.layer_blocks) + 1:
            raise RuntimeError("RepVGGPluXNetwork: wrong number of levels")
        if self.training or differentiable:
            if not levels[0].is_cuda:
                raise RuntimeError("RepVGGPluXNetwork: HIP device tensors required; there is no CPU fallback")
            maps = [x.transpose(1, 2).reshape(x.shape[0], x.shape[2], int(h), int(w)) for x, (h, w) in zip(levels, shapes)]
            return [o.flatten(2).transpose(1, 2) for o in self.forward_train(maps)]
        plan = self._folded(levels[0].dtype)  # (the kernels' wrappers refuse CPU tensors: no fallback)
        shapes = [(int(h), int(w)) for h, w in shapes]
        L = len(levels)
        inner = [levels[-1]]
        for idx in range(L - 1, 0, -1):  # top-down
            hh, hw_ = shapes[idx]
            wl, bl = plan["lateral"][idx - 1]
            high = FO.neck_combine(_token_matmul(inner[0], wl), hh, hw_, bias=bl, activation=True)
            inner[0] = high
            inner.insert(0, self._csp(plan["layer"][idx - 1], high, shapes[idx], levels[idx - 1], shapes[idx - 1], True))
        outs = [inner[0]]
        for idx in range(L - 1):  # bottom-up
            h, w = shapes[idx]
            wd, bd, pd = plan["down"][idx]
            down = FO.neck_conv3x3(outs[-1], h, w, wd, bd, stride=2, activation=True, packed=pd)
            if down.shape[1] != shapes[idx + 1][0] * shapes[idx + 1][1]:
                raise RuntimeError("RepVGGPluXNetwork: level sizes are not a stride-2 pyramid")
            outs.append(self._csp(plan["pan"][idx], down, shapes[idx + 1], inner[idx + 1], shapes[idx + 1], False))
        return outs

    # ---- training form (repnet.py:211-245 with batch-statistics norms) ---------------------------------------------
    def _conv_norm(self, m: Conv2dNormActivation, x: Tensor, act: bool = True) -> Tensor:
        from .data_parallel import sync_batch_norm_train
        conv = m[0]
        y = torch.nn.functional.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, 1, conv.groups)
        if self.training:
            y = sync_batch_norm_train(y, m[1], self.process_group)
        else:   # eval mode under autograd: running statistics, as nn.BatchNorm2d.eval()
            bn = m[1]
            y = torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        return torch.nn.functional.silu(y) if act else y

    @staticmethod
    def _gate(se: "SqueezeAndExcitation", x: Tensor) -> Tensor:
        """Attention-pooled gate (models/bricks/basic.py:29-54): softmax over the pixels of conv_mask(x) weights the
        pixels, the pooled vector goes through C -> C/16 -> ReLU -> C -> sigmoid and scales x."""
        B, C, H, W = x.shape
        logit = torch.nn.functional.conv2d(x, se.conv_mask.weight, se.conv_mask.bias).view(B, H * W)
        context = torch.einsum("bcp,bp->bc", x.reshape(B, C, H * W), logit.softmax(-1))
        hidden = torch.relu(context @ se.se_module[0].weight.view(-1, C).t())
        gate = torch.sigmoid(hidden @ se.se_module[2].weight.view(C, -1).t())
        return gate.view(B, C, 1, 1) * x

    def _csp_train(self, layer: CSPRepPluXLayer, x: Tensor) -> Tensor:
        y = self._conv_norm(layer.conv1, x)
        for blk in layer.bottlenecks:
            z = self._conv_norm(blk.conv1, y, act=False) + blk.alpha * self._conv_norm(blk.conv2, y, act=False)
            y = self._gate(blk.se_module, torch.nn.functional.silu(z)) + y
        return y + self._conv_norm(layer.conv2, x)

    def forward_train(self, feats: Sequence[Tensor]) -> List[Tensor]:
        """NCHW levels (fine to coarse) -> the same, training mode: differentiable, batch statistics over the
        process group ``self.process_group`` (None = all ranks / this process), running statistics updated."""
        L = len(feats)
        inner = [feats[-1]]
        for idx in range(L - 1, 0, -1):  # top-down
            high = self._conv_norm(self.lateral_convs[idx - 1], inner[0])
            inner[0] = high
            up = torch.nn.functional.interpolate(high, size=feats[idx - 1].shape[-2:], mode="nearest")
            inner.insert(0, self._csp_train(self.layer_blocks[idx - 1], torch.cat([up, feats[idx - 1]], 1)))
        outs = [inner[0]]
        for idx in range(L - 1):  # bottom-up
            down = self._conv_norm(self.downsample_blocks[idx], outs[-1])
            outs.append(self._csp_train(self.pan_blocks[idx], torch.cat([down, inner[idx + 1]], 1)))
        return outs

    def forward_memory(self, memory: Tensor, level_shapes: Sequence[Tuple[int, int]], differentiable: bool = False) -> Tensor:
        """``memory`` ``[B, sum h*w, C]`` -> the neck's output in the same layout
        (models/bricks/salience_transformer.py:185-192 without its transposes)."""
        sizes = [int(h) * int(w) for h, w in level_shapes]
        # (row ranges of the memory as they lie: the levels only enter the neck through ``_token_matmul``, which takes
        #  the batch stride as it is -- four copies of the pyramid less per step)
        levels = list(memory.split(sizes, 1))
        return torch.cat(self.forward_levels(levels, level_shapes, differentiable), 1)

    def forward(self, x: "OrderedDict[str, Tensor]"):
        keys = list(x.keys())
        feats = list(x.values())
        shapes = [tuple(f.shape[-2:]) for f in feats]
        outs = self.forward_levels([f.flatten(2).transpose(1, 2).contiguous() for f in feats], shapes)
        output = OrderedDict()
        for k, o, (h, w) in zip(keys, outs, shapes):
            output[k] = o.transpose(1, 2).reshape(o.shape[0], o.shape[2], h, w)
        if self.extra_block:  # F.max_pool2d(last, 1, 2, 0): every second pixel
            output["pool"] = list(output.values())[-1][:, :, ::2, ::2]
        return output


def build_neck(channels: int = 256, num_levels: int = 4, groups: int = 4) -> RepVGGPluXNetwork:
    """The neck of ``configs/salience_detr/salience_detr_resnet50_800_1333.py:57-63``."""
    return RepVGGPluXNetwork([channels] * num_levels, [channels] * num_levels, groups=groups,
                             norm_layer=nn.BatchNorm2d, activation=nn.SiLU)
