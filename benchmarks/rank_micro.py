import sys, torch
sys.path.insert(0, "/root/repo")
from salience_detr_amd import filter_ops as F
DEV = "cuda:0"
for n, k in ((4200, 3360), (1050, 1050), (273, 273)):
    s = torch.randn(2, n, device=DEV); m = torch.zeros(2, n, dtype=torch.bool, device=DEV); fv = s.min().reshape(1)
    for _ in range(10):
        F.masked_topk_desc(s, k, mask=m, fill_with_global_min=True, fill_value=fv)
torch.cuda.synchronize()
