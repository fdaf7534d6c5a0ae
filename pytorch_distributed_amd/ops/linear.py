"""Linear layer: planned library GEMMs for forward / input gradient, native kernel for the
weight gradient.

* Forward ``y = x W^T + b`` and input gradient ``dx = dy W`` contract over a dimension that is
  contiguous in both operands — plain library GEMMs. They go through the C++ GEMM planner
  (``csrc/runtime/gemm_planner.cpp``): hipBLASLt with the bias add fused into the epilogue
  and the algorithm picked by timing the library's candidates the first time each shape is
  seen (5-25 % faster than the default heuristic on the tall shapes of training).
* The weight gradient ``dW = dY^T . X`` contracts over the token dimension — the row index of
  both operands — and produces a small output; there the library reaches only
  0.4-0.85 PFLOP/s on GPT-2 shapes, so it is done by ``csrc/kernels/gemm_wgrad.hip``
  (split-K over tokens, transposed LDS operand reads), which also returns the bias gradient
  as a by-product instead of a separate reduction pass over ``dY``.

* The input gradient ``dx = dy W`` contracts over the ROW index of ``W [out, in]``; the library's
  kernels for that layout are 5-15 % slower than the ones for two contraction-contiguous
  operands, and transposing a weight (``csrc/kernels/transpose.hip``) costs microseconds against a
  GEMM of hundreds. ``input_grad_mm`` therefore times both routes the first time a shape is seen
  and keeps the faster one (``PDA_NO_DGRAD_TRANSPOSE=1`` pins the direct layout).

fp32 tensors on the GPU (``--dtype fp32``, the reference's storage precision) take the same path
with bf16 MFMA inputs: activations, weight and bias are rounded to bf16 for the products,
accumulation is fp32, results and gradients are widened back to fp32. Other GPU dtypes are an
error; there is no eager fallback on the GPU. Shapes the native kernels do not cover (dimensions
that are not multiples of 8, token counts that are not multiples of 32) fall to the library GEMM
inside the same function, and CPU tensors use ``torch`` matmuls.
``PDA_NO_NATIVE_WGRAD=1`` / ``PDA_NO_GEMM_PLANNER=1`` switch the two native paths off
(A/B measurements).

Gradient accumulation (``accumulation_window``): the weight gradient of a window of micro-batches
is the product over the CONCATENATION of their tokens, so inside a declared window the
``(dY, X)`` pairs of all but the last micro-batch are kept instead of multiplied, and the last
micro-batch's backward runs ONE long-K GEMM per weight over all of them
(``linear_wgrad_segments``: the kernel reads up to four separate buffers as one operand). That
replaces N short split-K GEMMs + N slice reductions + N read-modify-writes of the gradient by one
of each, accumulates the whole window in fp32, and leaves the non-final backward passes with only
their input-gradient GEMMs. The kept operands cost memory (about 0.33 GB per GPT-2-large block per
micro-batch of 8 x 1024 tokens), so keeping stops at a budget taken from the memory that is free
at the first backward of the window; ``PDA_DEFER_WGRAD=0`` switches it off.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.nn.functional as F

from .. import _ext


def _native_wgrad_ok(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16):
        return False
    if os.environ.get("PDA_NO_NATIVE_WGRAD") == "1":
        return False
    return _ext.require().wgrad_supported(dy2.shape[0], dy2.shape[1], x2.shape[1])


_PLAN_FILE_CHECKED = False


def _planner_ok(*mats: torch.Tensor) -> bool:
    global _PLAN_FILE_CHECKED
    if os.environ.get("PDA_NO_GEMM_PLANNER") == "1":
        return False
    for m in mats:
        if not (m.is_cuda and m.dtype == torch.bfloat16 and m.dim() == 2 and m.stride(1) == 1
                and m.stride(0) % 8 == 0 and m.stride(0) >= m.shape[1] and m.data_ptr() % 16 == 0
                and m.numel() > 0):
            return False
    if not _PLAN_FILE_CHECKED:  # searched-once kernel choices, adopted before the first planned GEMM
        _PLAN_FILE_CHECKED = True
        from ..tuning import gemm_plans
        gemm_plans.load_default_once()
    return True


def planned_mm(a: torch.Tensor, b: torch.Tensor, *, trans_b: bool = False, bias=None, out=None) -> torch.Tensor:
    """``a @ b`` (or ``a @ b.T`` with ``trans_b``) ``+ bias`` for 2-D row-major bf16 GPU matrices
    through the GEMM planner; plain ``torch`` otherwise."""
    mats = (a, b) if out is None else (a, b, out)
    if _planner_ok(*mats) and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous())):
        return _ext.require().lt_matmul(a, False, b, trans_b, out, "none" if bias is None else "bias", bias)
    bm = b.t() if trans_b else b
    if bias is not None:
        res = torch.addmm(bias, a, bm)
        return res if out is None else out.copy_(res)
    return torch.mm(a, bm, out=out) if out is not None else torch.mm(a, bm)


# (device, tokens, out, in) -> route of the input-gradient GEMM chosen by timing on this device:
#   0  direct layout (contraction over the weight's row index)
#   1  transpose the weight, then the contraction-contiguous GEMM
# (Within each route the planner also tries the library's bias-epilogue kernels with a zero bias.)
_DGRAD_TRANSPOSED: dict = {}
_DGRAD_MIN_TOKENS = 2048


def _time_us(fn, runs: int = 3) -> float:
    fn()
    fn()  # the first call plans the GEMM, the second warms the chosen kernel
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(runs):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) * 1e3 / runs


def input_grad_mm(dy2: torch.Tensor, weight: torch.Tensor, dy_padded: torch.Tensor = None) -> torch.Tensor:
    """``dy2 @ weight`` for ``dy2 [M, N]`` and ``weight [N, K]`` (the input gradient of a linear layer).

    ``dy_padded`` is ``dy2`` including zero-valued padding columns up to its row pitch (the LM
    head's logits gradient, N = 50257 padded to 50304); with it the transposed route contracts
    over the padded, aligned length against a zero-padded transposed weight."""
    M, N = dy2.shape
    K = weight.shape[1]
    direct = lambda: planned_mm(dy2, weight)  # noqa: E731
    if (M < _DGRAD_MIN_TOKENS or K % 8 != 0 or not _planner_ok(dy2, weight)
            or os.environ.get("PDA_NO_DGRAD_TRANSPOSE") == "1"):
        return direct()
    a = dy2 if dy_padded is None else dy_padded
    pad_to = 8 if dy_padded is None else a.shape[1]  # transposed weight rows: N rounded up to this
    if pad_to % 8 != 0 or a.shape[0] != M or a.shape[1] != -(-N // pad_to) * pad_to or not _planner_ok(a):
        return direct()
    ext = _ext.require()

    def weight_t():
        # The weights do not change inside an accumulation window: micro-batches after the first reuse
        # the transposed copy (values, not storage: a re-gathered sharded weight holds the same numbers).
        if not _ACTIVE:
            return ext.transpose_2d(weight, pad_to)
        key = (id(weight), pad_to)
        hit = _WINDOW.transposed.get(key)
        if hit is None or hit[0] is not weight:
            hit = _WINDOW.transposed[key] = (weight, ext.transpose_2d(weight, pad_to))
        return hit[1]

    transposed = lambda: planned_mm(a, weight_t(), trans_b=True)  # noqa: E731
    routes = (direct, transposed)
    key = (weight.device.index, M, N, K)
    choice = _DGRAD_TRANSPOSED.get(key)
    if choice is None:
        if torch.cuda.is_current_stream_capturing():
            return direct()  # cannot time inside a graph capture; decided by an eager step
        # two alternating measurements of each route; an extra kernel has to pay for itself clearly
        totals = [0.0, 0.0]
        for _ in range(2):
            for i, route in enumerate(routes):
                totals[i] += _time_us(route)
        choice = 1 if totals[1] < 0.99 * totals[0] else 0
        _DGRAD_TRANSPOSED[key] = choice
    return routes[int(choice)]()


def sink_is_wide(param) -> bool:
    """True when the owner's gradient sink hands out a buffer of another dtype than the parameter's
    (fp32 gradient accumulation of a bf16 parameter)."""
    wide = getattr(param, "_pda_grad_sink_dtype", None)
    return wide is not None and wide != param.dtype


def _grad_target(param, sink, wide_ok: bool = True):
    """Where a parameter's gradient kernel should write: ``(tensor, accumulate)`` from the owner's
    sink, else the existing contiguous ``.grad`` (accumulate), else None (fresh tensor to autograd).
    ``wide_ok=False``: the caller's kernel can only write the parameter's own dtype, so a wider
    sink is left alone (the owner's accumulator hook then adds the returned gradient itself)."""
    if param is None:
        return None
    if sink is not None and not wide_ok and sink_is_wide(param):
        sink = None
    target = sink() if sink is not None else None
    # (only a leaf has a .grad of its own: a derived tensor, e.g. the bf16 copy of an fp32 weight, gets a
    # fresh gradient that autograd carries on to its source)
    if target is None and param.is_leaf and param.grad is not None and param.grad.is_contiguous():
        target = (param.grad, True)
    return target


class _Window:
    """State of the gradient-accumulation window whose backward passes are running."""
    __slots__ = ("index", "total", "kept", "kept_bytes", "budget", "measured_budget", "transposed")

    def __init__(self):
        self.index, self.total = 0, 1
        self.kept: dict = {}  # id(weight) -> (weight, [(dy2, x2), ...])
        self.kept_bytes = 0   # of the micro-batch `index`
        self.budget = None    # bytes one micro-batch may keep; measured at the window's first keep
        self.measured_budget = None  # the last measurement (reused while a HIP graph is being captured)
        self.transposed: dict = {}  # (id(weight), row padding) -> (weight, transposed copy) for the input-gradient GEMMs

    def clear(self):
        self.kept.clear()
        self.transposed.clear()
        self.kept_bytes = 0
        self.budget = None


_WINDOW = _Window()
_ACTIVE = False  # inside accumulation_window()


def deferred_weight_grads_enabled() -> bool:
    return os.environ.get("PDA_DEFER_WGRAD", "1") != "0"


@contextlib.contextmanager
def accumulation_window(index: int, total: int):
    """Declares that the backward passes run inside belong to micro-batch ``index`` of a window of
    ``total`` micro-batches whose gradients are summed before the optimizer step, and that nobody
    reads the weight gradients before the window's last backward has run (no gradient
    synchronisation in between: the caller is inside ``no_sync()``, or the model is not wrapped).
    The last micro-batch (``index == total - 1``) computes every kept product."""
    global _ACTIVE
    w = _WINDOW
    if total < 2 or _ACTIVE or not deferred_weight_grads_enabled():
        yield
        return
    if index == 0 or index != w.index + 1 or total != w.total:
        w.clear()  # a new window (or an abandoned one: its operands are dropped, like its gradients)
    w.index, w.total, w.kept_bytes = index, total, 0
    _ACTIVE = True
    try:
        yield
    except BaseException:
        _ACTIVE = False
        w.clear()
        raise
    _ACTIVE = False
    if index == total - 1:
        left = len(w.kept)
        w.clear()
        if left:  # a layer that ran in an earlier micro-batch but not in the last one: its products were never formed
            raise RuntimeError(f"accumulation_window: {left} weight(s) had operands kept from earlier micro-batches "
                               "but took no part in the last micro-batch's backward; their gradients would be "
                               "lost. Set PDA_DEFER_WGRAD=0 for models whose layers run conditionally.")


def pending_weight_grads() -> int:
    """Number of weights with kept operands (0 outside a window and after its last backward)."""
    return len(_WINDOW.kept)


def _free_bytes(device) -> int:
    """Device memory a new allocation can still get: free at the driver plus cached by the allocator."""
    free, _ = torch.cuda.mem_get_info(device)
    return free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)


def _keep_for_later(dy2, x2, weight) -> bool:
    """Keeps one (dY, X) pair for the window's last backward; False when the budget is used up."""
    w = _WINDOW
    if w.budget is None:
        if w.measured_budget is not None and dy2.is_cuda and torch.cuda.is_current_stream_capturing():
            # no queries of the driver while a graph is being recorded: the warm-up steps before the capture
            # measured the same window
            w.budget = w.measured_budget
        else:
            # Measured where a backward starts, i.e. at the activation peak: what is free now is what the
            # later micro-batches' kept operands can grow into.
            w.budget = w.measured_budget = int(0.8 * _free_bytes(dy2.device)) // max(1, w.total - 1)
    size = dy2.numel() * dy2.element_size() + x2.numel() * x2.element_size()
    if w.kept_bytes + size > w.budget:
        return False
    w.kept_bytes += size
    entry = w.kept.get(id(weight))
    if entry is None:
        entry = w.kept[id(weight)] = (weight, [])
    entry[1].append((dy2, x2))
    return True


def _wgrad_over_pairs(pairs, need_b, w_out, w_acc, b_out, b_acc):
    """One kernel call per run of up to ``wgrad_max_segments()`` equally shaped pairs."""
    ext = _ext.require()
    cap = ext.wgrad_max_segments()
    dw = db = None
    i = 0
    while i < len(pairs):
        dy0, x0 = pairs[i]
        n = 1
        while (i + n < len(pairs) and n < cap and pairs[i + n][0].shape == dy0.shape
               and pairs[i + n][1].shape == x0.shape):
            n += 1
        while n > 1 and not ext.wgrad_segments_supported(dy0.shape[0], dy0.shape[1], x0.shape[1], n):
            n -= 1
        chunk = pairs[i:i + n]
        dw, db = ext.linear_wgrad_segments([p[0] for p in chunk], [p[1] for p in chunk], bool(need_b),
                                           w_out, w_acc, b_out, b_acc)
        w_out, w_acc = dw, True
        if need_b:
            b_out, b_acc = db, True
        i += n
    return dw, db


def weight_grads(dy2, x2, weight, bias, grad_sink, bias_sink, need_w: bool, need_b: bool):
    """``(dW, db)`` of ``y = x W^T + b`` from ``dy2 [tokens, out]`` and ``x2 [tokens, in]``.

    On the native path the kernel's slice-reduction pass writes (or adds) each gradient straight
    into its target — the owner's flat buffer (gradient sink), or an existing ``.grad`` when
    gradients are accumulated without a wrapper — and None is returned for that parameter: autograd
    then gets None, its accumulator node still runs, so the owner's hook on it fires and counts the
    parameter as ready."""
    dw = db = None
    if need_w and _native_wgrad_ok(dy2, x2):
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        x2 = x2.contiguous()
        pairs = None
        # Kept operands are filed under the weight's identity, so only a leaf parameter takes part: a derived
        # tensor (the bf16 copy of an fp32 weight, made anew in every forward) would never be found again.
        if _ACTIVE and weight.is_leaf:
            if _WINDOW.index < _WINDOW.total - 1:
                if _keep_for_later(dy2, x2, weight):
                    return None, None  # autograd's accumulator nodes still run (see above)
            else:
                entry = _WINDOW.kept.pop(id(weight), None)
                if entry is not None:
                    pairs = entry[1] + [(dy2, x2)]
        w_target = _grad_target(weight, grad_sink)
        b_target = _grad_target(bias, bias_sink) if need_b else None
        w_out, w_acc = w_target if w_target is not None else (None, False)
        b_out, b_acc = b_target if b_target is not None else (None, False)
        if pairs is not None:
            dw_native, db_native = _wgrad_over_pairs(pairs, need_b, w_out, w_acc, b_out, b_acc)
        else:
            dw_native, db_native = _ext.require().linear_wgrad(dy2, x2, bool(need_b), w_out, w_acc, b_out, b_acc)
        dw = dw_native if w_target is None else None
        if need_b and b_target is None:
            db = db_native
    else:
        if need_w:
            dw = torch.mm(dy2.t(), x2)
        if need_b:
            db = dy2.sum(0)
    return dw, db


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, regen=None):
        # With a recipe the input is rebuilt in backward from tensors its producer keeps anyway
        # (ops.regen); otherwise the input itself is kept.
        ctx.regen = None if regen is None else regen.recipe()
        ctx.x_shape = x.shape
        if regen is None:
            ctx.save_for_backward(weight, x)
        else:
            ctx.regen_none = tuple(t is None for t in regen.tensors)
            ctx.save_for_backward(weight, *[t for t in regen.tensors if t is not None])
        ctx.has_bias = bias is not None
        # Gradient sink: a sharding / reducing wrapper that owns a flat gradient buffer tags the
        # weight with a callable returning (slice_of_flat_buffer, accumulate).
        ctx.grad_sink = getattr(weight, "_pda_grad_sink", None)
        ctx.bias_sink = getattr(bias, "_pda_grad_sink", None)
        ctx.bias = bias  # only consulted for its .grad / sink; the values are not needed in backward
        x2 = x.reshape(-1, x.shape[-1])
        # The result is allocated in its final shape (not returned as a view of a 2-D tensor), so
        # a following op may legally work on it in place (the rotary kernel does).
        y = torch.empty(*x.shape[:-1], weight.shape[0], dtype=x.dtype, device=x.device)
        planned_mm(x2, weight, trans_b=True, bias=bias, out=y.view(-1, weight.shape[0]))
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, *kept = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = input_grad_mm(dy2, weight).view(ctx.x_shape)
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if need_w or need_b:
            if ctx.regen is None:
                x = kept[0]
            else:
                it = iter(kept)
                x = ctx.regen.rebuild(*[None if missing else next(it) for missing in ctx.regen_none])
            x2 = x.reshape(-1, x.shape[-1])
            dw, db = weight_grads(dy2, x2, weight, ctx.bias, ctx.grad_sink, ctx.bias_sink, need_w, need_b)
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None, regen=None) -> torch.Tensor:
    """``x @ weight.T + bias`` (``torch.nn.functional.linear`` semantics).

    ``regen`` (an :class:`ops.regen.Regen`) says how to rebuild ``x`` in backward from tensors that
    are kept anyway; with it ``x`` is not saved. It has to describe exactly ``x``."""
    if _ext.use_hip(x):
        if x.dtype == torch.bfloat16:
            return _LinearFn.apply(x, weight, bias, regen)
        if x.dtype == torch.float32:  # fp32 storage: bf16 MFMA inputs, fp32 accumulation, widened result
            lo = torch.bfloat16  # (the bf16 copy of the input is kept; a recipe would rebuild the fp32 one)
            return _LinearFn.apply(x.to(lo), weight.to(lo), None if bias is None else bias.to(lo)).to(torch.float32)
        raise NotImplementedError(f"linear on the GPU: dtype {x.dtype} has no kernel (bf16, or fp32 through bf16 MFMA inputs)")
    if regen is not None and torch.is_grad_enabled():
        return _LinearFn.apply(x, weight, bias, regen)  # same autograd node on the CPU, torch matmuls inside
    return F.linear(x, weight, bias)
