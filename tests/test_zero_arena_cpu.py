"""CPU: the training step's zero arena (salience_detr_amd/zero_arena.py) -- sizing pass, slices in request order, one clear per
step, fall-through outside a step / for other dtypes, growth when a later step asks for more."""
import pytest
import torch

from salience_detr_amd import zero_arena as Z


def test_first_step_measures_then_slices_are_served_and_cleared():
    a = Z.ZeroArena("cpu")
    with a.step():
        x = Z.zeros((3, 5), torch.float32, "cpu")
        y = Z.zeros(7, torch.float32, "cpu")
        assert a.fills_saved == 0 and x.shape == (3, 5) and not x.any() and not y.any()
    assert a.buf is not None and a.buf.numel() >= 128
    for _ in range(2):
        with a.step():
            x = Z.zeros((3, 5), torch.float32, "cpu")
            y = Z.zeros(7, torch.float32, "cpu")
            assert a.fills_saved == 2
            assert x.data_ptr() == a.buf.data_ptr() and y.data_ptr() == a.buf.data_ptr() + 64 * 4
            assert not x.any() and not y.any()
            x.fill_(3.0)
            y.fill_(-1.0)             # dirt for the next step's single clear to remove


def test_fall_through_and_growth():
    assert Z.zeros((2, 2), torch.float32, "cpu").shape == (2, 2)       # no active step: torch.zeros
    a = Z.ZeroArena("cpu")
    with a.step():
        Z.zeros(10, torch.float32, "cpu")
    with a.step():
        h = Z.zeros(4, torch.float16, "cpu")                            # other dtypes are not served
        assert h.dtype == torch.float16 and a.fills_saved == 0
        big = Z.zeros(1000, torch.float32, "cpu")                       # more than measured: torch.zeros now, a larger buffer next
        assert big.numel() == 1000 and not big.any() and a.fills_saved == 0
    assert a.buf.numel() >= 1000
    with a.step():
        assert Z.zeros(1000, torch.float32, "cpu").data_ptr() == a.buf.data_ptr()
    with a.step():
        with pytest.raises(RuntimeError):
            with Z.ZeroArena("cpu").step():
                pass
