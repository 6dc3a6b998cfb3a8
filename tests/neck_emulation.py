"""TEST INFRASTRUCTURE: torch restatements of the three neck kernels' CONTRACTS (include/salience_hip.h (13)), with the
signatures of their wrappers in salience_detr_amd/filter_ops.py.  The CPU suite patches them in to check the host
logic of salience_neck.py (BatchNorm / RepVGG folding, weight layouts, the split 1x1 convolutions) against the
reference's vectors; the GPU suite checks each kernel against them.  Never imported by the product."""
import torch
import torch.nn.functional as F


def _nchw(t, h, w):
    B, _, C = t.shape
    return t.float().transpose(1, 2).reshape(B, C, h, w)


def _tokens(t):
    return t.flatten(2).transpose(1, 2).contiguous()


def conv3x3(x, height, width, weight, bias, stride=1, activation=False, packed=None):
    G, _, _, ci, co = weight.shape
    w = weight.permute(0, 4, 3, 1, 2).reshape(G * co, ci, 3, 3)  # [G][ky][kx][ci][co] -> [G*co, ci, ky, kx]
    y = F.conv2d(_nchw(x[:, :, :G * ci], height, width), w, bias, stride=stride, padding=1, groups=G)
    return _tokens(F.silu(y) if activation else y).to(x.dtype)


def combine(a, height, width, up=None, up_hw=None, bias=None, activation=True):
    v = a.float()
    if up is not None:
        v = v + _tokens(F.interpolate(_nchw(up, *up_hw), size=(height, width), mode="nearest"))
    if bias is not None:
        v = v + bias
    return (F.silu(v) if activation else v).to(a.dtype).contiguous()


def gate_shortcut(y, mask_weight, squeeze_weight, excite_weight, shortcut, shortcut2=None):
    yf = y.float()
    context = torch.einsum("bnc,bn->bc", yf, (yf @ mask_weight).softmax(1))
    gate = torch.sigmoid(torch.relu(context @ squeeze_weight.t()) @ excite_weight.t())
    out = gate[:, None] * yf + shortcut.float()
    if shortcut2 is not None:
        out = out + shortcut2.float()
    return out.to(y.dtype)
