"""Scores of the hoisted salience head against the per-level form and an fp64 evaluation.  python benchmarks/hoist_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as F, salience_filtering as SF
from salience_detr_amd.salience_filtering import MaskPredictor
DEV = "cuda:0"
torch.manual_seed(0)
B, C = 2, 256
pred = MaskPredictor(C, C).to(DEV)
with torch.no_grad():
    pred.layer1[0].weight.add_(0.1 * torch.randn(C, device=DEV)); pred.layer1[0].bias.add_(0.1 * torch.randn(C, device=DEV))
    pred.layer1[1].bias.add_(0.1 * torch.randn(C, device=DEV))
enc, norm = torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)
alpha = torch.tensor([0.3], device=DEV)
for (h, w) in [(13, 21), (50, 84), (100, 167)]:
    n = h * w
    x = torch.randn(B, n, C, device=DEV)
    coarse = torch.randn(B, 1, (h + 1) // 2, (w + 1) // 2, device=DEV)
    for use_enc in (True, False):
        for use_coarse in (True, False):
            kw = dict(enc_output=enc, enc_output_norm=norm) if use_enc else {}
            kc = dict(coarse_score=coarse, level_hw=(h, w), alpha=alpha) if use_coarse else {}
            with torch.no_grad():
                mo_a = torch.empty_like(x) if use_enc else None
                mo_b = torch.empty_like(x) if use_enc else None
                a = F.salience_head(x, pred, memory_out=mo_a, **kw, **kc)
                hh = F.salience_head_hoist(x, pred, memory_out=mo_b, **kw)
                b = F.salience_head(x, pred, hoisted=hh.level(0, n), **kc)
                # fp64 reference
                xd = x.double()
                if use_enc:
                    xd = torch.nn.functional.layer_norm(torch.nn.functional.linear(xd, enc.weight.double(), enc.bias.double()), (C,), norm.weight.double(), norm.bias.double(), norm.eps)
                if use_coarse:
                    up = torch.nn.functional.interpolate(coarse.double(), size=(h, w), mode="bilinear", align_corners=True).view(B, n, 1)
                    xd = xd + xd * up * alpha.double()
                l1n, l1 = pred.layer1[0], pred.layer1[1]
                z = torch.nn.functional.gelu(torch.nn.functional.linear(torch.nn.functional.layer_norm(
                    xd, (C,), l1n.weight.double(), l1n.bias.double(), l1n.eps), l1.weight.double(), l1.bias.double()))
                z = torch.cat([z[..., :128], z[..., 128:].mean(1, keepdim=True).expand(-1, n, -1)], -1)
                for i, m in enumerate(pred.layer2):
                    z = torch.nn.functional.linear(z, m.weight.double(), m.bias.double()) if i % 2 == 0 else torch.nn.functional.gelu(z)
                ref = z.squeeze(-1)
            print(h, w, "enc" if use_enc else "   ", "coarse" if use_coarse else "      ",
                  "direct-vs-hoisted %.3g" % (a - b).abs().max().item(), "direct-vs-f64 %.3g" % (a.double() - ref).abs().max().item(),
                  "hoisted-vs-f64 %.3g" % (b.double() - ref).abs().max().item(),
                  "memory_out equal" if (not use_enc or torch.equal(mo_a, mo_b)) else "MEMORY_OUT DIFFERS", "score scale %.3g" % ref.abs().max().item())
