"""What the finest level's salience scores look like to the histogram sort (bins of span / 4096): crowded bins and the
distinct values inside them.  Benchmark inputs, batch 2."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_hot_path()
model.load_state_dict(syn.det_state_dict(model.state_dict()))
model = model.to(dev).eval()
model.set_encoder_dtype(torch.bfloat16, torch.float16)
sizes, canvas, level_shapes, _, (feats, masks, pos) = bench.make_inputs(2, 800, 1333, dev, seed=0)
with torch.no_grad():
    memory, score_maps, aux = model(feats, masks, pos, image_sizes=sizes, canvas=canvas, return_aux=True)
for lvl in range(4):
    s = score_maps[lvl].flatten(1).float().cpu().numpy()
    m = masks[lvl].flatten(1).cpu().numpy()
    for b in range(s.shape[0]):
        v = s[b].copy()
        fill = s.min()
        v[m[b]] = fill
        other = v[v != v.min()]
        top, span = other.max(), v.max() - v.min()
        bins = np.minimum(((top - other) * (4096 / span)).astype(np.int64), 4095)
        cnt = np.bincount(bins, minlength=4096)
        crowded = np.nonzero(cnt > 48)[0]
        print(f"level {lvl} image {b}: n {v.size} floor {int((v == v.min()).sum())} max bin {cnt.max()} crowded bins {len(crowded)}")
        for c in crowded[:12]:
            vals, counts = np.unique(other[bins == c], return_counts=True)
            order = np.argsort(-counts)[:4]
            print(f"   bin {c}: {cnt[c]} keys, {len(vals)} distinct, top counts {counts[order].tolist()}")
