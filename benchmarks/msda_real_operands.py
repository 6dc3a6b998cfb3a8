"""The bordered MSDA kernel on the operands the benchmark step really hands it (value maps from the value projection,
projection slabs / reference points / row orders of the six encoder layers of one eager forward), per layer:
time with the step's row order, with the list (score) order, with other tile sizes, the ablations and the phase stamps of
benchmarks/msda_bordered_ab.py.

    python benchmarks/msda_real_operands.py [--ablate 1,2,3,12,15,16] [--stamps] [--tiles 8,32]
"""
import argparse
import ctypes
import json
import os
import sys

# the ablated kernels and the phase stamps live in the benchmark build of the library only (benchmarks/libsalience_hip_ablate.so,
# `python salience_detr_amd/csrc/build.py --ablations`): bound below, before the first operator call
USE_ABLATION_BUILD = any(a.startswith(("--ablate", "--stamps")) for a in sys.argv[1:])

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from salience_detr_amd import _hip, ms_deform_attn as M  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402
from salience_detr_amd.hot_path import build_hot_path  # noqa: E402
from benchmarks.msda_resident_ab import timeit  # noqa: E402

if USE_ABLATION_BUILD:
    _hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsalience_hip_ablate.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--ablate", default="")
    ap.add_argument("--tiles", default="")
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--layers", default="0,1,2,3,4,5")
    ap.add_argument("--out", default="gpurun_out/msda_real_operands.json")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = build_hot_path()
    model.load_state_dict(syn.det_state_dict(model.state_dict()))
    model = model.to(dev).eval()
    model.set_encoder_dtype(torch.bfloat16, torch.float16)
    sizes, canvas, level_shapes, _, (feats, masks, pos) = bench.make_inputs(2, 800, 1333, dev, seed=0)
    calls = []
    real = M.msda_bordered_forward

    def grab(value_hm, levels, ref, proj, row_order=None, out_dtype=None, chunks=0):
        calls.append(dict(value=value_hm.clone(), levels=levels, ref=ref.clone(), proj=proj.clone(),
                          order=None if row_order is None else row_order.clone(), out_dtype=out_dtype))
        return real(value_hm, levels, ref, proj, row_order=row_order, out_dtype=out_dtype, chunks=chunks)

    sorted_index = {}
    M.msda_bordered_forward = grab
    try:
        with torch.no_grad():
            _, _, aux = model(feats, masks, pos, image_sizes=sizes, canvas=canvas, return_aux=True)
    finally:
        M.msda_bordered_forward = real
    tok = aux["foreground_inds"][0]
    rows = []
    for k in [int(x) for x in args.layers.split(",")]:
        c = calls[k]
        nq = c["proj"].shape[2]
        B, Mh, Np, D = c["value"].shape
        Nv = sum(h * w for h, w in c["levels"])
        alg = bench.algorithmic_bytes(B, Nv, nq, Mh, D, 4, 4, 2, 2, 2, 2)
        row = {"layer": k, "nq": nq, "algorithmic_MB": round(alg / 1e6, 2)}

        def run(order):
            return lambda: real(c["value"], c["levels"], c["ref"], c["proj"], row_order=order, out_dtype=c["out_dtype"])

        def record(name, fn):
            fn()
            t = timeit(fn, args.reps)
            row[name + "_us"] = round(t, 2)
            row[name + "_frac"] = round(alg / t / 1e6 / 8.0, 4)

        record("step_order", run(c["order"]))
        record("list_order", run(None))
        for tile in [int(t) for t in args.tiles.split(",") if t]:
            record(f"tile{tile}", run(M.spatial_row_order(tok[:, :nq], c["levels"], tile)))
        for abl in [a for a in args.ablate.split(",") if a]:
            os.environ["SDETR_MSDA_ABLATE"] = abl
            record(f"ablate{abl}", run(c["order"]))
            os.environ.pop("SDETR_MSDA_ABLATE")
        if args.stamps:
            st = torch.zeros(256 * 8 + 64, dtype=torch.int64, device=dev)
            fn_ = _hip.lib().sdetr_msda_debug_stamps
            fn_.restype, fn_.argtypes = None, [ctypes.c_void_p]
            fn_(st.data_ptr())
            os.environ["SDETR_MSDA_ABLATE"] = "32"
            for _ in range(3):
                run(c["order"])()
            torch.cuda.synchronize()
            os.environ.pop("SDETR_MSDA_ABLATE")
            t = st[:2048].view(256, 8).cpu().double() * 0.01
            t0 = t[:, 0].min()
            row["stamps_us"] = {"first_loads_issued": round(float((t[:, 1] - t0).mean()), 2),
                                "past_barrier": round(float((t[:, 2] - t0).mean()), 2),
                                "loop": round(float((t[:, 3] - t[:, 2]).mean()), 2),
                                "end_first": round(float(t[:, 3].min() - t0), 2), "end_mean": round(float((t[:, 3] - t0).mean()), 2),
                                "end_last": round(float(t[:, 3].max() - t0), 2),
                                "end_by_head": [round(float((t[h::8, 3] - t0).mean()), 2) for h in range(8)]}
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
