"""Top-k kernel timings (kernel time via rocprofv3 --kernel-trace --stats of this script)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import filter_ops as FO
B = 2
for N, k in ((16800, 6680), (4200, 3360), (1050, 1050), (273, 273), (11363, 11363), (11363, 300), (9090, 300),
             (6817, 300), (2272, 300), (22323, 3600)):
    s = torch.randn(B, N, device="cuda")
    for _ in range(5):
        FO.masked_topk_desc(s, k)
torch.cuda.synchronize()
