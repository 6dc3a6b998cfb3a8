"""Where the waves of bt_main_kernel (msda_backward_tiled.hip) spend their cycles: a -DBT_STAMPS scratch build
(benchmarks/bt_variant.sh stamps -DBT_STAMPS) sums s_memtime differences per phase over all waves into the workspace
header; this script runs one launch per query count and prints the shares.  LIB=benchmarks/libbt_stamps.so."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from salience_detr_amd import _hip  # noqa: E402

_hip.LIB_PATH = os.path.abspath(os.environ.get("LIB", "benchmarks/libbt_stamps.so"))
from salience_detr_amd import synthetic as syn  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
PHASES = ["prologue", "set-up", "records+loads issued", "scatter", "dots+sums", "stores+atomic samples",
          "wait item's waves", "flush", "wait flush", "TOTAL"]


def main():
    lib = _hip.lib()
    out = []
    for nq in [int(a) for a in sys.argv[1:]] or [11363, 4545]:
        value, shapes, lsi, loc, aw = syn.make_msda_inputs(2, nq, LEVELS, 8, 32, 4, seed=11, spread_px=4.0)
        go = syn.det_randn("gout_ab", (2, nq, 256))
        value, shapes, lsi, loc, aw, go = [t.cuda() for t in (value, shapes, lsi, loc, aw, go)]
        B, Nv, M, D = value.shape
        gv, gl, ga = torch.zeros_like(value), torch.empty_like(loc), torch.empty_like(aw)
        nbytes = lib.sdetr_msda_col2im_lds_workspace_bytes(B, nq, M, 4)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            code = lib.sdetr_msda_col2im_lds_f32(_hip.stream_ptr(), go.data_ptr(), value.data_ptr(), shapes.data_ptr(),
                                                 lsi.data_ptr(), loc.data_ptr(), aw.data_ptr(), B, Nv, M, D, 4, nq, 4,
                                                 gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), nbytes)
            _hip.check(code, "bt_stamps")
        torch.cuda.synchronize()
        st = ws[32:112].cpu().view(torch.int64).tolist()
        tot = st[9] or 1
        out.append({"queries": nq, "share": {n: round(v / tot, 4) for n, v in zip(PHASES, st)},
                    "mean_wave_cycles": round(tot / (256 * 8))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
