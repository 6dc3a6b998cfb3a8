import sys, os, torch
sys.path.insert(0, os.getcwd())
from salience_detr_amd import linear_x3 as X
sys.path.insert(0, os.path.join(os.getcwd(), "benchmarks"))
from gemm_x3_bench import time_us
for T in (9090, 13634, 18180, 22726):
    K, N = 2048, 256
    x = torch.randn(T, K, device="cuda"); w = torch.randn(N, K, device="cuda"); gy = torch.randn(T, K, device="cuda") ; w2 = torch.randn(K, N, device="cuda")
    out = torch.zeros(T, N, device="cuda")
    row = {"T": T}
    row["torch_y"] = round(time_us(lambda: x @ w.t()), 1)
    for sp in (1, 2, 3, 4):
        row[f"x3_y_s{sp}"] = round(time_us(lambda: X.gemm_x3(x, True, w, True, T, N, K, reduction_splits=sp, out=out.zero_() if sp > 1 else out)), 1)
    # dx of linear1: dy [T,2048] . w1 [2048,256] -> [T,256]: A = dy k-major, B = w1 reduction-major
    row["torch_dx"] = round(time_us(lambda: gy @ w2), 1)
    for sp in (1, 2, 4):
        row[f"x3_dx_s{sp}"] = round(time_us(lambda: X.gemm_x3(gy, True, w2, False, T, N, K, reduction_splits=sp, out=out.zero_() if sp > 1 else out)), 1)
    print(row, flush=True)
