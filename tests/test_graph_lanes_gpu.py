"""GPU: several batches of the hot path in flight (salience_detr_amd/graph_lanes.py) -- each lane's hipGraph replay,
side by side with the others, returns the bits of the eager forward on the same batch."""
import pytest
import torch

from salience_detr_amd import synthetic as syn
from salience_detr_amd.graph_lanes import GraphLanes
from salience_detr_amd.hot_path import build_hot_path

# (thread method: a replay that never completes blocks inside a C call, where the signal method cannot interrupt)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180, method="thread")]
DEV = "cuda:0"


def _inputs(sizes, seed):
    _, masks = syn.make_masks(sizes)
    shapes = [tuple(x.shape[-2:]) for x in masks]
    feats = syn.make_feats(len(sizes), shapes, 256, seed=seed)
    pos = [syn.sine_position_embedding(x, 128) for x in masks]
    return tuple([t.to(DEV) for t in ts] for ts in (feats, masks, pos))


def test_lanes_return_the_eager_bits():
    """bf16 mode: every launch of the forward is one of this repository's kernels with per-call scratch.  (The fp32
    mode routes its projections through the framework's library GEMMs; three such graphs replayed side by side did
    not complete on MI355X / ROCm 7.2 -- see the note in graph_lanes.py -- so lanes are a bf16-path feature.)"""
    sizes = [(480, 640), (448, 600)]
    canvas = syn.pad_to_32(480, 640)
    m = build_hot_path()
    m.load_state_dict(syn.det_state_dict(m.state_dict()))
    m = m.to(DEV).eval()
    m.set_encoder_dtype(torch.bfloat16, torch.float16)

    def fn(f, mk, p):
        return m(f, mk, p, image_sizes=sizes, canvas=canvas)[0]

    batches = [_inputs(sizes, seed) for seed in range(5)]
    with torch.no_grad():
        expect = [fn(*b).clone() for b in batches]
    lanes = GraphLanes(fn, batches[0], lanes=3)
    assert len(lanes) == 3
    # five batches over three lanes: results are read before a lane is reused
    pending = []
    got = [None] * len(batches)
    for i, b in enumerate(batches):
        if len(pending) == len(lanes):
            j, lane = pending.pop(0)
            got[j] = lane.synchronize().outputs.clone()
        pending.append((i, lanes.submit(b)))
    for j, lane in pending:
        got[j] = lane.synchronize().outputs.clone()
    for g, e in zip(got, expect):
        assert torch.equal(g, e)
    # replays of the resident inputs, all lanes at once
    for _ in range(6):
        lanes.launch_next()
    lanes.synchronize()
    for lane, j in zip(lanes.lanes, (3, 4, 2)):   # lane 0 last held batch 3, lane 1 batch 4, lane 2 batch 2
        assert torch.equal(lane.outputs, expect[j])


def test_lane_rejects_inputs_of_another_shape():
    m = build_hot_path().to(DEV).eval()
    m.set_encoder_dtype(torch.bfloat16, torch.float16)
    sizes = [(320, 480)]
    canvas = syn.pad_to_32(320, 480)
    a = _inputs(sizes, 0)
    lanes = GraphLanes(lambda f, mk, p: m(f, mk, p, image_sizes=sizes, canvas=canvas)[0], a, lanes=1)
    other = _inputs([(352, 480)], 0)
    with pytest.raises(ValueError):
        lanes.submit(other)
    with pytest.raises(ValueError):
        lanes.submit(a[:2])


def test_submit_of_temporaries_survives_reallocation():
    """`lanes.submit(make_batch())` drops the sources as soon as submit returns; the caching allocator may
    hand their memory to new current-stream work before the lane's asynchronous copy has run.  `Lane.load` marks the
    sources as in use on the lane's stream: the reallocated scratch must land elsewhere (or wait)."""
    sizes = [(320, 480)]
    canvas = syn.pad_to_32(320, 480)
    m = build_hot_path()
    m.load_state_dict(syn.det_state_dict(m.state_dict()))
    m = m.to(DEV).eval()
    m.set_encoder_dtype(torch.bfloat16, torch.float16)

    def fn(f, mk, p):
        return m(f, mk, p, image_sizes=sizes, canvas=canvas)[0]

    keep = _inputs(sizes, 7)
    with torch.no_grad():
        expect = fn(*keep).clone()
    lanes = GraphLanes(fn, keep, lanes=2)
    for _ in range(4):
        # hold the lane's stream back so that the copy is still pending when the sources are released
        lane = lanes.next_lane()
        with torch.cuda.stream(lane.stream):
            torch.cuda._sleep(20_000_000)
        tmp = tuple([t.clone() for t in ts] for ts in keep)
        lane.load(tmp)
        shapes = [[t.shape for t in ts] for ts in tmp]
        dtypes = [[t.dtype for t in ts] for ts in tmp]
        del tmp
        # same sizes -> the allocator's first candidates are the blocks just released
        junk = [[torch.full(s, 3, dtype=d, device=DEV) if d != torch.bool else torch.ones(s, dtype=d, device=DEV)
                 for s, d in zip(ss, dd)] for ss, dd in zip(shapes, dtypes)]
        lane.launch().synchronize()
        assert torch.equal(lane.outputs, expect)
        del junk


def test_capture_guard_flags_memset_nodes():
    """salience_detr_amd/graph_guard.py: a hipMemsetAsync captured into a graph is not reproduced by replay on this stack
    (CHANGELOG round 4); the guard finds such nodes at capture time.  ``torch.zeros`` of a fresh block inside a captured
    region is a memset node, ``tensor.zero_()`` is a kernel."""
    from salience_detr_amd import graph_guard
    x = torch.ones(1 << 16, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x.zero_()
    torch.cuda.current_stream().wait_stream(side)
    g = graph_guard.new_graph()
    with torch.cuda.graph(g):
        x.zero_()
        y = x + 1
    n = graph_guard.assert_replay_safe(g, "kernel fills")
    if n == 0:
        pytest.skip("this torch exposes no graph handle")
    assert n >= 2 and graph_guard.memset_nodes(g) == 0
    import ctypes
    rt = graph_guard._runtime()
    rt.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    g2 = graph_guard.new_graph()
    with torch.cuda.graph(g2):
        rt.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, x.numel() * 4, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        z = x * 2
    assert graph_guard.memset_nodes(g2) == 1
    with pytest.raises(RuntimeError, match="memset node"):
        graph_guard.assert_replay_safe(g2, "raw memset")
    del y, z
