"""Host logic of the deferred parameter-gradient scheduler (tatt_amd.functional.SIDE): which side lane a registered closure runs
with (current stage, `lag`, the bucket its parameters are filed under), generator closures, accumulation into `.grad`.  No GPU: the
batched split-K reduction switch around a flush is stubbed out."""
import pytest
import torch

from tatt_amd import functional as Fh


@pytest.fixture
def side(monkeypatch):
    calls = []
    monkeypatch.setattr(Fh.ops, "reduce_defer", lambda on: calls.append(bool(on)))
    s = Fh.SIDE
    old = (s.enabled, s.stage, s.due_of)
    s.enabled, s.stage, s.due_of = True, 0, {}
    s._pending.clear()
    s._keep.clear()
    yield s, calls
    s.flush()
    s.release()
    s.enabled, s.stage, s.due_of = old


def _leaf(v=0.0):
    return torch.full((2,), v, requires_grad=True)


def test_closures_run_with_the_lane_of_their_stage_lag_or_bucket(side):
    s, calls = side
    a, b, c = _leaf(), _leaf(), _leaf()
    s.due_of = {id(b): 3}                               # b is filed under bucket 3
    ran = []
    s.stage = 1
    assert s.submit((a,), lambda: (ran.append("a") or torch.ones(2),)) == (None,)
    s.submit((b,), lambda: (ran.append("b") or torch.ones(2) * 2,))
    s.submit((c,), lambda: (ran.append("c") or torch.ones(2) * 3,), lag=1)
    s.flush(0)
    assert ran == [] and calls == []                   # nothing is due before stage 1
    s.flush(1)
    assert ran == ["a"] and a.grad is not None and b.grad is None and c.grad is None
    assert calls == [True, False]                       # reductions are batched around the closures of ONE flush
    s.flush(2)
    assert ran == ["a", "c"] and torch.equal(c.grad, torch.ones(2) * 3)
    s.flush(None)                                       # the last lane takes whatever is left
    assert ran == ["a", "c", "b"] and torch.equal(b.grad, torch.ones(2) * 2)
    s.flush(None)
    assert ran == ["a", "c", "b"]                       # nothing runs twice


def test_generator_closures_resume_after_the_batched_reduction_and_grads_accumulate(side):
    s, calls = side
    a = _leaf()
    order = []

    def gen():
        order.append("gemms issued")
        yield
        order.append("tail")
        return (torch.ones(2),)
    s.submit((a,), gen)
    s.submit((a,), lambda: (order.append("plain") or torch.ones(2) * 4,))
    s.flush(0)
    # both closures issue their GEMMs, THEN the batched reduction runs (reduce_defer(False)), then the generator's tail
    assert order == ["gemms issued", "plain", "tail"]
    assert torch.equal(a.grad, torch.ones(2) * 5)       # second result added to the first


def test_disabled_or_non_leaf_parameters_run_inline(side):
    s, _ = side
    a = _leaf()
    s.enabled = False
    out = s.submit((a,), lambda: (torch.ones(2),))
    assert torch.equal(out[0], torch.ones(2)) and a.grad is None
    s.enabled = True
    nl = _leaf() * 2.0                                  # not a leaf: autograd must get the gradient itself
    out = s.submit((nl,), lambda: (torch.ones(2) * 7,))
    assert torch.equal(out[0], torch.ones(2) * 7)
