"""Where do the index sets of the hoisted and the per-level salience head differ on the full-size "mixed" digest inputs
(tests/test_hotpath_gpu.py::test_hotpath_full_size_digest), and how close are the scores involved?"""
import os
import sys
import zlib

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_hotpath_gpu as T  # noqa: E402
from salience_detr_amd import salience_filtering as SF  # noqa: E402

DEV = "cuda:0"
d = np.load(os.path.join(T.G, "hotpath_full_digest.npz"))
for tag, sizes in (("single", [(800, 1333)]), ("mixed", [(800, 1333), (800, 1066)])):
    m, feats, masks, pos = T._full_model_and_inputs(sizes)
    m = m.to(DEV).eval()
    outs = {}
    for hoist in (False, True):
        SF.HOIST_HEAD = hoist
        with torch.no_grad():
            memory, score_maps, aux = m([f.to(DEV) for f in feats], [x.to(DEV) for x in masks], [p.to(DEV) for p in pos],
                                        return_aux=True)
        outs[hoist] = (aux, score_maps)
    SF.HOIST_HEAD = True
    a0, a1 = outs[False][0], outs[True][0]
    fs0, fs1 = a0["foreground_score"].cpu(), a1["foreground_score"].cpu()
    print(tag, "foreground_score max dev %.3g" % (fs0 - fs1).abs().max().item())
    focus = a0["focus_token_nums"].cpu()
    for k in range(len(a0["foreground_inds"])):
        i0, i1 = a0["foreground_inds"][k].cpu(), a1["foreground_inds"][k].cpu()
        for b in range(i0.shape[0]):
            n = min(int(focus[b]), i0.shape[1])
            s0, s1 = set(i0[b, :n].tolist()), set(i1[b, :n].tolist())
            ref_ok = [None, None]
            if n == i0.shape[1]:
                want = int(d[f"{tag}.inds{k}_set_crc"][b])
                ref_ok = [zlib.crc32(np.sort(x[b].numpy()).astype(np.int64).tobytes()) == want for x in (i0, i1)]
            if s0 != s1 or ref_ok[1] is False:
                only0, only1 = sorted(s0 - s1), sorted(s1 - s0)
                print(f"  layer {k} image {b}: n={n} per-level-only {only0} hoisted-only {only1}  crc ok (per-level, hoisted) {ref_ok}")
                for t in only0 + only1:
                    print(f"     token {t}: score per-level {fs0[b, t].item():.9g} hoisted {fs1[b, t].item():.9g}")
    li0, li1 = a0["level_inds"], a1["level_inds"]
    for l in range(4):
        for b in range(li0[l].shape[0]):
            s0, s1 = set(li0[l][b].tolist()), set(li1[l][b].tolist())
            if s0 != s1:
                print(f"  level {l} image {b}: top-k sets differ: {sorted(s0 - s1)} vs {sorted(s1 - s0)}")
                for t in sorted(s0 ^ s1):
                    print(f"     token {t}: score per-level {fs0[b, t].item():.9g} hoisted {fs1[b, t].item():.9g}")
