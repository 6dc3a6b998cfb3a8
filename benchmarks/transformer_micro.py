"""Rows N1 + N2 timing on top of the encoder hot path: the whole SalienceTransformer.forward (bf16 encoder/decoder,
fp16 value maps, 800x1333 + 800x1066, 900 proposals), per stage with stream events (eager) and end to end under a
hipGraph (static_proposals: no host read-back).

    python benchmarks/transformer_micro.py [--iters 20] [--neck]     (--neck: with the RepVGGPluX neck, row N3)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import synthetic as syn  # noqa: E402
from salience_detr_amd.hot_path import SalienceEncoderHotPath  # noqa: E402
from salience_detr_amd.salience_transformer import build_salience_transformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--neck", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    image_sizes = [(800, 1333), (800, 1066)]
    tr = build_salience_transformer(with_neck=a.neck)
    tr.load_state_dict(syn.det_state_dict(tr.state_dict()))
    tr = tr.eval().to(dev).set_dtype(torch.bfloat16, torch.float16)
    tr.static_proposals = True
    img_mask, masks = syn.make_masks(image_sizes)
    canvas = tuple(img_mask.shape[-2:])
    shapes = [tuple(m.shape[-2:]) for m in masks]
    feats = [f.to(dev) for f in syn.make_feats(2, shapes, 256, 0)]
    masks = [m.to(dev) for m in masks]
    pe = lambda mask: syn.sine_position_embedding(mask, 128)
    pos = [pe(m) for m in masks]

    def whole():
        with torch.no_grad():
            return tr(feats, masks, pos, image_sizes=image_sizes, canvas=canvas)

    def stages():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        with torch.no_grad():
            ev[0].record()
            memory, sal, aux = SalienceEncoderHotPath.forward(tr, feats, masks, pos, image_sizes=image_sizes,
                                                              canvas=canvas, return_aux=True)
            ev[4].record()
            if tr.neck is not None:
                memory = tr.neck.forward_memory(memory, shapes)
            ev[1].record()
            enc_cls, enc_box = tr.select_proposals(memory, aux["mask_flatten"], shapes)
            ev[2].record()
            tr.decoder(query=tr.tgt_embed.weight.expand(2, -1, -1), value=memory, key_padding_mask=aux["mask_flatten"],
                       reference_points=enc_box, spatial_shapes=aux["spatial_shapes"],
                       level_start_index=aux["level_start_index"], valid_ratios=aux["valid_ratios"])
            ev[3].record()
        return ev

    for _ in range(5):
        whole()
    torch.cuda.synchronize()
    acc = [0.0, 0.0, 0.0, 0.0]
    for _ in range(a.iters):
        ev = stages()
        torch.cuda.synchronize()
        acc[0] += ev[0].elapsed_time(ev[4])
        acc[3] += ev[4].elapsed_time(ev[1])
        for i in range(1, 3):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
    print("eager per stage (ms): encoder path %.3f, proposals %.3f, decoder %.3f, neck %.3f" % tuple(v / a.iters for v in acc),
          flush=True)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            whole()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        whole()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print("hipGraph whole transformer: %.3f ms per batch of 2 (%.1f img/s)" % (ms, 2e3 / ms))


if __name__ == "__main__":
    main()
