"""CPU, world_size 2, gloo: the N > 1 path (image sharding + flat gradient all-reduce) that the driver
runs at 2/4/8 GPUs over RCCL.  Gradients of a 2-rank data-parallel step must equal the single-process
gradients of the full batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from salience_detr_amd import synthetic as syn
from salience_detr_amd.data_parallel import (FlatGradAllReducer, OverlappedGradReducer, broadcast_parameters,
                                             shard_range)
from salience_detr_amd.salience_filtering import MaskPredictor


def test_shard_range_covers_everything_once():
    for n in (0, 1, 2, 7, 16, 17):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = shard_range(n, r, world)
                assert 0 <= a <= b <= n
                seen += list(range(a, b))
            assert seen == list(range(n))
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _model():
    torch.manual_seed(0)
    m = MaskPredictor(32, 32)
    m.load_state_dict(syn.det_state_dict(m.state_dict()))
    return m


def _loss(model, x):
    return (model(x) ** 2).mean()


def _worker(rank, world, port, bucket_bytes, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        if rank == 1:  # replicas start different; broadcast must fix that
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        broadcast_parameters(model, src=0)
        x = syn.det_randn("dp.x", (4, 50, 32))
        a, b = shard_range(x.shape[0], rank, world)
        # mean over the global batch = average of per-rank means when shards are equal
        _loss(model, x[a:b]).backward()
        if rank == 1:  # a parameter that only one rank touched
            model.layer2[4].bias.grad = None
        red = FlatGradAllReducer(model.parameters(), bucket_bytes=bucket_bytes)
        red.all_reduce(average=True)
        torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [None, 4096])
def test_two_rank_gradients_match_single_process(tmp_path, bucket_bytes):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, bucket_bytes, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(os.path.join(tmp_path, "g0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "g1.pt"))
    model = _model()
    x = syn.det_randn("dp.x", (4, 50, 32))
    # reference: average of the two half-batch losses.  NOTE the MaskPredictor couples the tokens of ONE
    # image batch through its global mean over dim 1 only, so per-rank half batches are exactly the two terms.
    (0.5 * (_loss(model, x[:2]) + _loss(model, x[2:]))).backward()
    for name, p in model.named_parameters():
        expect = p.grad
        if name == "layer2.4.bias":  # rank 1 contributed zeros for it
            m2 = _model()
            (0.5 * _loss(m2, x[:2])).backward()
            expect = dict(m2.named_parameters())[name].grad
        assert torch.allclose(g0[name], expect, atol=1e-6), name
        assert torch.equal(g0[name], g1[name]), name


# ---- the REAL hot-path parameter list (the module bench.py --mode train reduces): 300+ tensors, the class head shared
# between `encoder_class_head` and `encoder.enhance_mcsp`, parameters one rank did not use, overlapped bucket hooks ----
def _hot_path_model():
    from salience_detr_amd.hot_path import build_hot_path
    m = build_hot_path(embed_dim=32, num_heads=4, d_ffn=64, num_layers=2, num_classes=7, topk_sa=4, max_num_embedding=16)
    m.load_state_dict(syn.det_state_dict(m.state_dict(), num_heads=4))
    return m


def _synthetic_loss(model, rank_salt, skip=()):
    """A loss that touches every parameter (the HIP forward cannot run on the CPU): sum_p <p, r_p(rank)>^2."""
    total = 0.0
    for name, p in model.named_parameters():
        if name in skip:
            continue
        total = total + (p * syn.det_randn("dp.w." + name, p.shape, salt=rank_salt)).sum() ** 2
    return total


def _hot_worker(rank, world, port, overlapped, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _hot_path_model()
        params = list(model.parameters())
        red = (OverlappedGradReducer(params, bucket_bytes=16 << 10) if overlapped
               else FlatGradAllReducer(params, bucket_bytes=16 << 10))
        assert len(red.buckets) > 3
        skip = ("alpha", "encoder.layers.1.norm2.bias") if rank == 1 else ()
        for step in range(2):   # two steps: the hook state must reset
            model.zero_grad(set_to_none=True)
            _synthetic_loss(model, rank + 10 * step, skip).backward()
            red.all_reduce(average=True)
        torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"h{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlapped", [False, True])
def test_hot_path_parameter_list_two_ranks(tmp_path, overlapped):
    model = _hot_path_model()
    # the aliased class head is ONE parameter in the reducer's list
    assert model.encoder.enhance_mcsp.weight is model.encoder_class_head.weight
    names = [n for n, _ in model.named_parameters()]
    assert ("encoder_class_head.weight" in names) != ("encoder.enhance_mcsp.weight" in names)   # listed once
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_hot_worker, args=(2, port, overlapped, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(os.path.join(tmp_path, "h0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "h1.pt"))
    grads = []
    for rank in range(2):
        m = _hot_path_model()
        skip = ("alpha", "encoder.layers.1.norm2.bias") if rank == 1 else ()
        _synthetic_loss(m, rank + 10, skip).backward()      # second step's data
        grads.append({n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()})
    for n in g0:
        expect = 0.5 * (grads[0][n] + grads[1][n])
        assert torch.allclose(g0[n], expect, rtol=1e-5, atol=1e-6), n
        assert torch.equal(g0[n], g1[n]), n


# ---- SyncBatchNorm of the neck's training form (row N3): batch statistics over all ranks' pixels, one all-reduce per
# direction; two ranks with half the batch each == one process with the whole batch (outputs, input gradients, summed
# parameter gradients, running statistics) ----
def _bn_case():
    bn = torch.nn.BatchNorm2d(6)
    with torch.no_grad():
        bn.weight.copy_(1.0 + 0.1 * syn.det_randn("sbn.w", (6,)))
        bn.bias.copy_(0.1 * syn.det_randn("sbn.b", (6,)))
        bn.running_mean.copy_(0.2 * syn.det_randn("sbn.rm", (6,)))
        bn.running_var.copy_(0.5 + syn.det_rand("sbn.rv", (6,)))
    x = syn.det_randn("sbn.x", (4, 6, 5, 7)) * 2.0 + 0.5
    probe = syn.det_randn("sbn.p", (4, 6, 5, 7))
    return bn, x, probe


def _sbn_worker(rank, world, port, out_dir):
    from salience_detr_amd.data_parallel import sync_batch_norm_train
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bn, x, probe = _bn_case()
        a, b = shard_range(x.shape[0], rank, world)
        xs = x[a:b].clone().requires_grad_(True)
        y = sync_batch_norm_train(xs, bn)
        (y * probe[a:b]).sum().backward()
        torch.save(dict(y=y.detach(), dx=xs.grad, dw=bn.weight.grad, db=bn.bias.grad, rm=bn.running_mean.clone(),
                        rv=bn.running_var.clone()), os.path.join(out_dir, f"sbn{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_sync_batch_norm_two_ranks_equal_one_process_full_batch(tmp_path):
    from salience_detr_amd.data_parallel import sync_batch_norm_train
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"sbn{r}.pt")) for r in range(2))
    # one process, whole batch: torch's own BatchNorm2d in training mode is the statement of what must come out
    bn, x, probe = _bn_case()
    bn.train()
    xf = x.clone().requires_grad_(True)
    y = bn(xf)
    (y * probe).sum().backward()
    assert torch.allclose(torch.cat([r0["y"], r1["y"]]), y.detach(), atol=1e-5)
    assert torch.allclose(torch.cat([r0["dx"], r1["dx"]]), xf.grad, atol=1e-5)
    assert torch.allclose(r0["dw"] + r1["dw"], bn.weight.grad, atol=1e-4)
    assert torch.allclose(r0["db"] + r1["db"], bn.bias.grad, atol=1e-4)
    for r in (r0, r1):
        assert torch.allclose(r["rm"], bn.running_mean, atol=1e-6) and torch.allclose(r["rv"], bn.running_var, atol=1e-5)
    # single process without a process group: plain batch statistics
    bn2, x2, probe2 = _bn_case()
    x2 = x2.clone().requires_grad_(True)
    y2 = sync_batch_norm_train(x2, bn2)
    (y2 * probe2).sum().backward()
    assert torch.allclose(y2.detach(), y.detach(), atol=1e-5) and torch.allclose(x2.grad, xf.grad, atol=1e-5)
    assert torch.allclose(bn2.weight.grad, bn.weight.grad, atol=1e-4)


def test_sync_batch_norm_large_mean_and_optional_parts():
    """E[x^2] - mean^2 cancels when |mean| >> std; the shifted sums do not once the running mean tracks the
    data.  affine=False / track_running_stats=False / momentum=None follow nn.BatchNorm2d."""
    from salience_detr_amd.data_parallel import sync_batch_norm_train
    x = syn.det_randn("sbn.big", (4, 3, 6, 5)) * 0.01 + 1000.0
    bn = torch.nn.BatchNorm2d(3)
    with torch.no_grad():
        bn.running_mean.fill_(1000.0)
    ref = torch.nn.BatchNorm2d(3)
    with torch.no_grad():
        ref.running_mean.fill_(1000.0)
    ref.train()
    y = sync_batch_norm_train(x, bn)
    # fp64 statement of the same normalisation
    xd = x.double()
    m = xd.mean((0, 2, 3), keepdim=True)
    v = xd.var((0, 2, 3), unbiased=False, keepdim=True)
    want = ((xd - m) / torch.sqrt(v + bn.eps)).float()
    assert torch.allclose(y, want, atol=2e-3), float((y - want).abs().max())
    ref(x)
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-3)

    # affine=False, no running statistics
    plain = torch.nn.BatchNorm2d(6, affine=False, track_running_stats=False)
    _, x2, probe = _bn_case()
    xa = x2.clone().requires_grad_(True)
    ya = sync_batch_norm_train(xa, plain)
    (ya * probe).sum().backward()
    xb = x2.clone().requires_grad_(True)
    yb = plain.train()(xb)
    (yb * probe).sum().backward()
    assert torch.allclose(ya, yb, atol=1e-5) and torch.allclose(xa.grad, xb.grad, atol=1e-5)

    # momentum=None: cumulative moving average
    cma, cma_ref = torch.nn.BatchNorm2d(6, momentum=None), torch.nn.BatchNorm2d(6, momentum=None)
    cma_ref.train()
    for k in range(3):
        xk = x2 + 0.3 * k
        sync_batch_norm_train(xk, cma)
        cma_ref(xk)
    assert int(cma.num_batches_tracked) == 3
    assert torch.allclose(cma.running_mean, cma_ref.running_mean, atol=1e-5)
    assert torch.allclose(cma.running_var, cma_ref.running_var, atol=1e-4)


# ---- gradient accumulation (the reference's --accumulate-steps, util/engine.py:44): no_sync() for the first micro-steps;
# a second backward without it is detected and reduced correctly (not overlapped) ----
def _accum_worker(rank, world, port, use_no_sync, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _hot_path_model()
        red = OverlappedGradReducer(list(model.parameters()), bucket_bytes=16 << 10)
        skip = ("alpha",) if rank == 1 else ()
        for step in range(2):
            model.zero_grad(set_to_none=True)
            if use_no_sync:
                with red.no_sync():
                    _synthetic_loss(model, rank + 10 * step, skip).backward()
            else:
                _synthetic_loss(model, rank + 10 * step, skip).backward()
            _synthetic_loss(model, rank + 10 * step + 100, skip).backward()
            red.all_reduce(average=True)
        torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"a{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_no_sync", [True, False])
def test_overlapped_reducer_gradient_accumulation(tmp_path, use_no_sync):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_accum_worker, args=(2, port, use_no_sync, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(os.path.join(tmp_path, "a0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "a1.pt"))
    grads = []
    for rank in range(2):
        m = _hot_path_model()
        skip = ("alpha",) if rank == 1 else ()
        _synthetic_loss(m, rank + 10, skip).backward()
        _synthetic_loss(m, rank + 110, skip).backward()
        grads.append({n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()})
    for n in g0:
        expect = 0.5 * (grads[0][n] + grads[1][n])
        assert torch.allclose(g0[n], expect, rtol=1e-5, atol=1e-6), n
        assert torch.equal(g0[n], g1[n]), n


# ---- the reducer of the graph-replayed training step: pack (end of the captured region) -> one all-reduce -> p.grad
# becomes a slice of the reduced flat buffer ----
def _static_worker(rank, world, port, out_dir):
    from salience_detr_amd.data_parallel import StaticGradAllReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _hot_path_model()
        red = StaticGradAllReducer(model.parameters())
        skip = ("alpha", "encoder.layers.1.norm2.bias") if rank == 1 else ()
        for step in range(2):   # two steps: stale slices of the first must not leak into the second
            model.zero_grad(set_to_none=True)
            _synthetic_loss(model, rank + 10 * step, skip).backward()
            red.pack()
            red.all_reduce(average=True)
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.params, red.views))
        torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"s{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_static_grad_all_reducer_two_ranks(tmp_path):
    from salience_detr_amd.data_parallel import StaticGradAllReducer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_static_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(os.path.join(tmp_path, "s0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "s1.pt"))
    grads = []
    for rank in range(2):
        m = _hot_path_model()
        skip = ("alpha", "encoder.layers.1.norm2.bias") if rank == 1 else ()
        _synthetic_loss(m, rank + 10, skip).backward()      # second step's data
        grads.append({n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()})
    for n in g0:
        assert torch.allclose(g0[n], 0.5 * (grads[0][n] + grads[1][n]), rtol=1e-5, atol=1e-6), n
        assert torch.equal(g0[n], g1[n]), n
    # one process, no process group: the gradients pass through unchanged
    m = _hot_path_model()
    red = StaticGradAllReducer(m.parameters())
    _synthetic_loss(m, 3).backward()
    want = {n: p.grad.clone() for n, p in m.named_parameters()}
    red.pack()
    red.all_reduce()
    for n, p in m.named_parameters():
        assert torch.equal(p.grad, want[n]), n


def test_static_grad_all_reducer_accumulates_in_place():
    """after the first ``all_reduce`` every ``p.grad`` IS its slice of the flat buffer; a following backward
    under ``zero_grad(set_to_none=False)`` (or gradient accumulation over micro-steps) adds into that slice in place, and
    ``pack()`` must keep what it finds there instead of zeroing the buffer."""
    from salience_detr_amd.data_parallel import StaticGradAllReducer
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    unused = torch.nn.Parameter(torch.ones(4))                 # never receives a gradient: its slice must stay zero
    params = list(m.parameters()) + [unused]
    red = StaticGradAllReducer(params)
    xs = [torch.randn(7, 6) for _ in range(3)]
    m(xs[0]).square().sum().backward()
    red.pack()
    red.all_reduce()
    first = [p.grad.clone() for p in m.parameters()]
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.params, red.views))
    # (1) zero_grad(set_to_none=False): the views are zeroed in place, the next backward fills them
    for p in params:
        if p.grad is not None:
            p.grad.zero_()
    m(xs[1]).square().sum().backward()
    want = [g.clone() for g in (p.grad for p in m.parameters())]
    red.pack()
    red.all_reduce()
    for p, w in zip(m.parameters(), want):
        assert torch.equal(p.grad, w) and p.grad.abs().sum() > 0
    assert torch.equal(unused.grad, torch.zeros(4))
    # (2) accumulation over two micro-steps without zeroing
    m(xs[2]).square().sum().backward()
    red.pack()
    red.all_reduce()
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref.load_state_dict(m.state_dict())
    ref(xs[1]).square().sum().backward()
    ref(xs[2]).square().sum().backward()
    for p, q in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-6)
    assert first[0].abs().sum() > 0
