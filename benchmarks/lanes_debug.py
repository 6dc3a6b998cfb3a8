"""Debug aid: several independent batches through the hot path, each with its own hipGraph and stream.

    python benchmarks/lanes_debug.py --lanes 3 --layers 6 --how together|solo|eager

Replays the lanes side by side (together), one after the other with a sync in between (solo), or runs the eager
launches of each lane with a device sync and the entry's name on stderr after every library call (eager)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--how", default="together")
    ap.add_argument("--rounds", type=int, default=50)
    ap.add_argument("--snapshot", default="")
    a = ap.parse_args()
    if a.snapshot:
        torch.cuda.memory._record_memory_history(max_entries=200000)
    dev = torch.device("cuda", 0)
    model = bench.build_hot_path()
    model.load_state_dict(syn.det_state_dict(model.state_dict()))
    model = model.to(dev).eval()
    model.set_encoder_dtype(torch.bfloat16, torch.float16)
    model.encoder.max_layers = a.layers
    lanes = []
    for i in range(a.lanes):
        sizes, canvas, _, _, (f, m, p) = bench.make_inputs(2, 800, 1333, dev, seed=100 * i)

        def step(f=f, m=m, p=p, sizes=sizes, canvas=canvas):
            with torch.no_grad():
                return model(f, m, p, image_sizes=sizes, canvas=canvas)[0]
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        if a.how == "eager":
            lanes.append((torch.cuda.Stream(), step, None, step))
            continue
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            g, o = bench.capture(step, {})
        lanes.append((st, g.replay, o, step))      # `step` owns the lane's inputs: the graph only has their addresses
        print("captured lane", i, flush=True)
    torch.cuda.synchronize()
    if a.snapshot:
        import json
        snap = torch.cuda.memory._snapshot()
        segs = []
        for sg in snap["segments"]:
            blocks, addr = [], sg["address"]
            for b in sg["blocks"]:
                fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in b.get("frames", [])
                      if "/repo/" in f["filename"] or "salience" in f["filename"]][:4]
                blocks.append({"address": b.get("address", addr), "size": b["size"], "state": b["state"], "frames": fr})
                addr += b["size"]
            segs.append({"address": sg["address"], "size": sg["total_size"], "pool": str(sg.get("segment_pool_id")),
                         "blocks": blocks})
        events = []
        for ev in snap["device_traces"][0]:
            fr = [f"{os.path.basename(f['filename'])}:{f['line']}" for f in ev.get("frames", [])
                  if "/repo/" in f["filename"] or "salience" in f["filename"]][:3]
            events.append([ev["action"], ev["addr"], ev["size"], ev.get("stream", 0), fr])
        with open(a.snapshot, "w") as fh:
            json.dump({"segments": segs, "events": events}, fh)
        print("snapshot written", len(segs), flush=True)
    for r in range(a.rounds):
        for st, fn, _, _ in lanes:
            with torch.cuda.stream(st):
                fn()
            if a.how != "together":
                torch.cuda.synchronize()
                if r == 0:
                    print("lane replayed", flush=True)
        torch.cuda.synchronize()
    print("ok", a.lanes, a.layers, a.how, flush=True)


if __name__ == "__main__":
    main()
