"""Row N2 timing: the six-layer decoder (900 queries/image, 800x1333 memory) in bf16, eager and under a hipGraph.

    python benchmarks/decoder_micro.py [--batch 2] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import synthetic as syn  # noqa: E402
from salience_detr_amd.salience_decoder import SalienceTransformerDecoder, SalienceTransformerDecoderLayer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--queries", type=int, default=900)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    level_shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    layer = SalienceTransformerDecoderLayer(dropout=0.0)
    dec = SalienceTransformerDecoder(layer, 6, 91)
    dec.load_state_dict(syn.det_state_dict(dec.state_dict()))
    dec = dec.eval().to(dev).bfloat16()
    for l in dec.layers:
        l.cross_attn.value_dtype = torch.float16
    shapes = torch.tensor(level_shapes, dtype=torch.int64, device=dev)
    sizes = shapes.prod(1)
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    Nv, B, Nq = int(sizes.sum()), a.batch, a.queries
    q = syn.det_randn("dm.q", (B, Nq, 256)).to(dev).bfloat16()
    mem = syn.det_randn("dm.m", (B, Nv, 256)).to(dev).bfloat16()
    ref = torch.cat([syn.det_rand("dm.c", (B, Nq, 2)), syn.det_rand("dm.w", (B, Nq, 2)) * 0.5 + 0.01], -1).to(dev)
    vr = torch.ones(B, 4, 2, device=dev)
    mask = torch.zeros(B, Nv, dtype=torch.bool, device=dev)

    def step():
        with torch.no_grad():
            return dec(q, ref, mem, shapes, lsi, vr, mask)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    eager = timeit(step)
    print(f"eager {eager:.3f} ms", flush=True)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    graph = timeit(g.replay)
    print(f"decoder B={B} Nq={Nq}: eager {eager:.3f} ms, graph {graph:.3f} ms ({B / graph * 1e3:.1f} img/s)")


if __name__ == "__main__":
    main()
