import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from salience_detr_amd import _hip, synthetic as syn
from test_topk_attention_gpu import _modules, _reference
DEV = "cuda:0"
B, rows, N = 1, 64, 32
mha, norm = _modules(seed=1)
x = syn.det_randn("tk.x", (B, rows, 256)).to(DEV).to(torch.bfloat16)
pos = syn.det_randn("tk.pos", (B, rows, 256)).to(DEV).to(torch.bfloat16)
sel = torch.arange(N, device=DEV)[None].contiguous()
lib = _hip.lib()
npad = (N + 31) // 32 * 32
ws = torch.full((lib.sdetr_topk_attention_workspace_bytes(B, N) // 2,), float("nan"), dtype=torch.bfloat16, device=DEV)
q = x.clone()
code = lib.sdetr_topk_attention_bf16(_hip.stream_ptr(), q.data_ptr(), rows * 256, pos.data_ptr(), rows * 256, sel.data_ptr(), B, rows, N,
    mha.in_proj_weight.data_ptr(), mha.in_proj_bias.data_ptr(), mha.out_proj.weight.data_ptr(), mha.out_proj.bias.data_ptr(),
    norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), 256, 8, ws.data_ptr(), ws.numel() * 2)
torch.cuda.synchronize()
print("code", code)
qk = ws[:B * npad * 512].view(B, npad, 512).float()
vt = ws[B * npad * 512:].view(B, 8, 32, npad).float()
print("nan in qk", torch.isnan(qk).sum().item(), "vt", torch.isnan(vt).sum().item(), "out", torch.isnan(q.float()).sum().item())
w, b = mha.in_proj_weight.float(), mha.in_proj_bias.float()
xs, ps = x.float()[:, :N], pos.float()[:, :N]
eq = torch.nn.functional.linear(xs + ps, w[:512], b[:512])
ev = torch.nn.functional.linear(xs, w[512:], b[512:])
print("qk err", (qk[:, :N] - eq).abs().max().item())
print("vt err", (vt[:, :, :, :N] - ev.view(B, N, 8, 32).permute(0, 2, 3, 1)).abs().max().item())
exp = _reference(x, pos, sel, mha, norm)
print("out err", (q.float() - exp).abs()[:, :N].max().item(), "nan rows", torch.isnan(q.float()).any(-1).sum().item())
print(q.float()[0, :4, :8]); print(exp[0, :4, :8])
