"""Logic of the accumulation window (ops/linear.py) on the CPU: which (dY, X) pairs are kept, when
they are multiplied, in which groups, and what happens at the memory budget — with the GPU kernels
replaced by fp32 torch products that count their calls."""
import importlib

import pytest
import torch

from pytorch_distributed_amd import ops

L = importlib.import_module("pytorch_distributed_amd.ops.linear")  # `ops.linear` is the function of that name


class FakeExt:
    """Stands in for the native extension: same entry points, torch arithmetic, call log."""

    def __init__(self, cap=4):
        self.cap = cap
        self.calls = []

    def wgrad_supported(self, m, i, j):
        return True

    def wgrad_max_segments(self):
        return self.cap

    def wgrad_segments_supported(self, rows, i, j, n):
        steps = rows // 32
        return rows % 32 == 0 and steps & (steps - 1) == 0 and 1 <= n <= self.cap

    def _finish(self, dw, db, want_bias, out, acc, bout, bacc):
        if out is not None:
            out.copy_(out + dw.to(out.dtype) if acc else dw.to(out.dtype))
            dw = out
        if want_bias and bout is not None:
            bout.copy_(bout + db.to(bout.dtype) if bacc else db.to(bout.dtype))
            db = bout
        return [dw, db if want_bias else None]

    def linear_wgrad(self, dy, x, want_bias, out=None, acc=False, bout=None, bacc=False, out_rows=-1):
        self.calls.append(("one", 1))
        return self._finish(dy.t() @ x, dy.sum(0), want_bias, out, acc, bout, bacc)

    def linear_wgrad_segments(self, dys, xs, want_bias, out=None, acc=False, bout=None, bacc=False):
        assert 1 <= len(dys) <= self.cap and len(dys) == len(xs)
        self.calls.append(("segments", len(dys)))
        dy, x = torch.cat(dys), torch.cat(xs)
        return self._finish(dy.t() @ x, dy.sum(0), want_bias, out, acc, bout, bacc)


@pytest.fixture
def fake(monkeypatch):
    ext = FakeExt()
    monkeypatch.setattr(L, "_native_wgrad_ok", lambda dy2, x2: True)
    monkeypatch.setattr(L._ext, "require", lambda: ext)
    monkeypatch.setattr(L, "_free_bytes", lambda device: 1 << 40)
    monkeypatch.setenv("PDA_DEFER_WGRAD", "1")
    L._WINDOW.clear()
    yield ext
    L._WINDOW.clear()


def _layers(seed=0):
    torch.manual_seed(seed)
    return [torch.nn.Linear(24, 16), torch.nn.Linear(16, 8)]


def _backward(layers, x, g):
    h = x
    for lin in layers:
        h = L._LinearFn.apply(h, lin.weight, lin.bias, None)
    h.backward(g)


def _data(n, rows=64):
    gen = torch.Generator().manual_seed(3)
    return ([torch.randn(rows, 24, generator=gen) for _ in range(n)], [torch.randn(rows, 8, generator=gen) for _ in range(n)])


def _reference(n, rows=64):
    layers = _layers()
    xs, gs = _data(n, rows)
    for x, g in zip(xs, gs):
        torch.nn.functional.linear(torch.nn.functional.linear(x, layers[0].weight, layers[0].bias),
                                   layers[1].weight, layers[1].bias).backward(g)
    return [(lin.weight.grad.clone(), lin.bias.grad.clone()) for lin in layers]


@pytest.mark.parametrize("n,expected_groups", [(2, [2]), (4, [4]), (6, [4, 2]), (9, [4, 4, 1])])
def test_products_are_formed_once_in_groups_of_at_most_four(fake, n, expected_groups):
    layers = _layers()
    xs, gs = _data(n)
    for i, (x, g) in enumerate(zip(xs, gs)):
        with ops.accumulation_window(i, n):
            _backward(layers, x, g)
        if i < n - 1:
            assert ops.pending_weight_grads() == 2 and fake.calls == []
            assert all(lin.weight.grad is None and lin.bias.grad is None for lin in layers)
    assert ops.pending_weight_grads() == 0
    per_weight = [c[1] for c in fake.calls if c[0] == "segments"]
    assert sorted(per_weight) == sorted(expected_groups * 2), fake.calls
    for lin, (rw, rb) in zip(layers, _reference(n)):
        assert torch.allclose(lin.weight.grad, rw, atol=1e-4) and torch.allclose(lin.bias.grad, rb, atol=1e-4)


def test_rows_that_the_segment_kernel_cannot_address_fall_back_to_single_products(fake):
    n, rows = 3, 96  # 3 stages of 32 rows per micro-batch: not a power of two
    layers = _layers()
    xs, gs = _data(n, rows)
    for i, (x, g) in enumerate(zip(xs, gs)):
        with ops.accumulation_window(i, n):
            _backward(layers, x, g)
    assert [c[1] for c in fake.calls] == [1] * (2 * n)
    for lin, (rw, rb) in zip(layers, _reference(n, rows)):
        assert torch.allclose(lin.weight.grad, rw, atol=1e-4) and torch.allclose(lin.bias.grad, rb, atol=1e-4)


def test_budget_exhausted_means_immediate_products(fake, monkeypatch):
    """Room for the first layer's pair only: the second layer's products are formed per micro-batch and
    added to what the window's last backward writes."""
    n = 3
    layers = _layers()
    xs, gs = _data(n)
    pair_bytes = 64 * (8 + 16) * 4  # the last layer runs its backward first: dy [64, 8], x [64, 16]
    monkeypatch.setattr(L, "_free_bytes", lambda device: int(pair_bytes * (n - 1) / 0.8) + 8)
    for i, (x, g) in enumerate(zip(xs, gs)):
        with ops.accumulation_window(i, n):
            _backward(layers, x, g)
        if i < n - 1:
            assert ops.pending_weight_grads() == 1
    assert ("segments", 3) in fake.calls and fake.calls.count(("one", 1)) == n
    for lin, (rw, rb) in zip(layers, _reference(n)):
        assert torch.allclose(lin.weight.grad, rw, atol=1e-4) and torch.allclose(lin.bias.grad, rb, atol=1e-4)


def test_switch_off_and_single_micro_batch_windows_change_nothing(fake, monkeypatch):
    layers = _layers()
    xs, gs = _data(2)
    with ops.accumulation_window(0, 1):
        _backward(layers, xs[0], gs[0])
    assert fake.calls == [("one", 1)] * 2 and ops.pending_weight_grads() == 0
    monkeypatch.setenv("PDA_DEFER_WGRAD", "0")
    with ops.accumulation_window(0, 2):
        _backward(layers, xs[1], gs[1])
    assert len(fake.calls) == 4 and ops.pending_weight_grads() == 0


def test_a_layer_missing_from_the_last_micro_batch_is_an_error_not_a_lost_gradient(fake):
    layers = _layers()
    xs, gs = _data(2)
    with ops.accumulation_window(0, 2):
        _backward(layers, xs[0], gs[0])
    with pytest.raises(RuntimeError, match="took no part"):
        with ops.accumulation_window(1, 2):
            _backward(layers[:1], xs[1], torch.randn(64, 16))
    assert ops.pending_weight_grads() == 0


def test_an_abandoned_window_and_an_exception_leave_nothing_behind(fake):
    layers = _layers()
    xs, gs = _data(2)
    with ops.accumulation_window(0, 4):
        _backward(layers, xs[0], gs[0])
    assert ops.pending_weight_grads() == 2
    with ops.accumulation_window(0, 4):  # a new window starts: the old operands are dropped
        assert ops.pending_weight_grads() == 0
    with pytest.raises(ValueError):
        with ops.accumulation_window(1, 4):
            _backward(layers, xs[1], gs[1])
            raise ValueError("step failed")
    assert ops.pending_weight_grads() == 0


def test_gradient_sinks_receive_the_window_in_one_write(fake):
    """A wrapper's sink (flat fp32 buffer slice + accumulate flag) is asked once, at the last micro-batch."""
    layers = _layers()
    xs, gs = _data(4)
    asked = []
    flat = torch.zeros(16 * 24)

    def sink():
        asked.append(len(asked))
        return flat.view(16, 24), False

    layers[0].weight._pda_grad_sink = sink
    layers[0].weight._pda_grad_sink_dtype = torch.float32
    for i, (x, g) in enumerate(zip(xs, gs)):
        with ops.accumulation_window(i, 4):
            _backward(layers, x, g)
    assert asked == [0]
    assert layers[0].weight.grad is None  # written into the sink, None handed to autograd
    assert torch.allclose(flat.view(16, 24), _reference(4)[0][0], atol=1e-4)


def test_a_derived_weight_is_never_kept(fake):
    """fp32 storage runs every product on a bf16 COPY of the weight made anew in each forward: such a tensor has
    no identity to file kept operands under, so its products are formed per micro-batch (and nothing is left
    behind for the window's end to complain about); the leaf parameter next to it is still deferred."""
    n = 3
    layers = _layers()
    xs, gs = _data(n)
    for i, (x, g) in enumerate(zip(xs, gs)):
        with ops.accumulation_window(i, n):
            h = L._LinearFn.apply(x, layers[0].weight * 1.0, layers[0].bias * 1.0, None)  # derived: not a leaf
            h = L._LinearFn.apply(h, layers[1].weight, layers[1].bias, None)
            h.backward(g)
        assert ops.pending_weight_grads() == (1 if i < n - 1 else 0)
    assert sorted(c[0] for c in fake.calls) == ["one"] * n + ["segments"]
    for lin, (rw, rb) in zip(layers, _reference(n)):
        assert torch.allclose(lin.weight.grad, rw, atol=1e-4) and torch.allclose(lin.bias.grad, rb, atol=1e-4)
