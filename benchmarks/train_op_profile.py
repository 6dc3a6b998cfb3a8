"""Which framework ops (and which call sites) the training step of the hot path spends its device time in.

    python benchmarks/train_op_profile.py > gpurun_out/train_ops.txt

The eager step of `bench.py --mode train` (same model, inputs, loss, optimizer) under torch.profiler: self device time
per op, and per (op, input shapes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from salience_detr_amd import synthetic as syn  # noqa: E402
from salience_detr_amd.linear_x3 import use_x3_linear_  # noqa: E402
from salience_detr_amd.salience_criterion import SalienceCriterion  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = bench.build_hot_path()
    model.load_state_dict(syn.det_state_dict(model.state_dict()))
    model = model.to(dev).train()
    use_x3_linear_(model)
    sizes, canvas, level_shapes, _, (feats, masks, pos) = bench.make_inputs(2, 800, 1333, dev, seed=0)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, capturable=True, fused=True)
    criterion = SalienceCriterion()
    strides = [(canvas[0] / h, canvas[1] / w_) for h, w_ in level_shapes]
    targets = []
    for i in range(2):
        c = syn.det_rand(f"bench.box.c{i}", (12, 2), salt=0) * 0.8 + 0.1
        wh = 0.02 + syn.det_rand(f"bench.box.wh{i}", (12, 2), salt=0) ** 2 * 0.9
        targets.append({"boxes": torch.cat([c, wh], -1).to(dev)})
    staged = criterion.stage_boxes(targets, sizes, dev)
    w = None

    def step():
        nonlocal w
        opt.zero_grad(set_to_none=True)
        memory, score_maps = model(feats, masks, pos, image_sizes=sizes, canvas=canvas)
        if w is None:
            w = torch.randn_like(memory)
        loss = (memory * w).mean() + criterion(score_maps, targets, strides, sizes, staged=staged)["loss_salience"]
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    ka = prof.key_averages()
    print(ka.table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
    # the small-op families by input shape (which tensors are copied / filled / added / reduced)
    by_shape = {}
    for ev in prof.events():
        t = getattr(ev, "self_device_time_total", 0) or 0
        if t <= 0 or not ev.name.startswith("aten::"):
            continue
        key = (ev.name, str(ev.input_shapes)[:110])
        a = by_shape.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += t
    print("\nself device time per (op, input shapes), two steps:")
    for (name, shp), (n, t) in sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:90]:
        print(f"{t / 2e3:9.3f} ms/step  {n / 2:6.1f} calls/step  {name:34s} {shp}")


if __name__ == "__main__":
    main()
