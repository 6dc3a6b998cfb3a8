"""Run-to-run determinism of stage 1 of the salience head at level-0 size (z_local, partial sums, memory_out compared bit
for bit with the first run).  LIB=benchmarks/libv_NAME.so for a scratch build; RUNS=400."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from salience_detr_amd import _hip
if os.environ.get("LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
from salience_detr_amd import filter_ops as F
from salience_detr_amd.salience_filtering import MaskPredictor
DEV = "cuda:0"
torch.manual_seed(0)
B, C = 2, 256
pred = MaskPredictor(C, C).to(DEV)
enc, norm = torch.nn.Linear(C, C).to(DEV), torch.nn.LayerNorm(C).to(DEV)
alpha = torch.tensor([0.2], device=DEV)
lib = _hip.lib()
h, w = 100, 167
n = h * w
x = torch.randn(B, n, C, device=DEV)
coarse = torch.randn(B, 1, (h + 1) // 2, (w + 1) // 2, device=DEV)
l1n, l1 = pred.layer1[0], pred.layer1[1]
w_enc = F.packed_linear_weight(enc.weight, split3=True)
w1 = F.packed_linear_weight(l1.weight, split3=True)
nblk = lib.sdetr_salience_head_blocks(B, n)
def run():
    z = torch.zeros((B, n, 128), device=DEV); part = torch.zeros((B, nblk, 128), device=DEV); mo = torch.zeros((B, n, C), device=DEV)
    code = lib.sdetr_salience_head_stage1_x3(_hip.stream_ptr(), x.data_ptr(), x.stride(0), x.stride(1), B, n, C, w_enc.data_ptr(),
        enc.bias.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), None, coarse.data_ptr(),
        coarse.shape[-2], coarse.shape[-1], h, w, alpha.data_ptr(), l1n.weight.data_ptr(), l1n.bias.data_ptr(), float(l1n.eps),
        w1.data_ptr(), l1.bias.data_ptr(), mo.data_ptr(), mo.stride(0), z.data_ptr(), part.data_ptr())
    _hip.check(code, "s1")
    torch.cuda.synchronize()
    return z, part, mo
z0, p0, m0 = run()
bz = bp = bm = 0
rows = set()
for i in range(int(os.environ.get("RUNS", "50"))):
    z, p, m = run()
    if not torch.equal(z, z0):
        bz += 1
        d = (z != z0).any(-1).nonzero()
        for r in d[:4].tolist():
            rows.add((r[0], r[1], r[1] // 32, r[1] % 32))
    bp += not torch.equal(p, p0)
    bm += not torch.equal(m, m0)
print("z_local differs in", bz, "partial", bp, "memory", bm, "of N runs; sample (image, token, block, row in block):", sorted(rows)[:12])
# which columns of a bad row differ, and by how much
z, p, m = run()
d = (z != z0)
rows_bad = d.any(-1).nonzero()
print("bad rows this run:", len(rows_bad), "of", B * n)
for r in rows_bad[:6].tolist():
    cols = d[r[0], r[1]].nonzero().flatten().tolist()
    print("row", r, "cols differing:", len(cols), cols[:10], "max abs diff", float((z[r[0], r[1]] - z0[r[0], r[1]]).abs().max()))
