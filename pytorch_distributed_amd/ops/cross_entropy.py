"""LM head + softmax cross-entropy, fused.

``lm_head_cross_entropy(hidden, weight, targets)`` computes
``cross_entropy(hidden @ weight.T, targets)`` (mean over non-ignored targets)
— the reference's ``lm_head`` GEMM followed by ``F.cross_entropy``
(assignments/assignment1/model/my_gpt2.py:211-214,
assignments/assignment1/train/trainer.py:53-56) — as:

1. one GEMM into a bf16 logits buffer whose rows are padded to a multiple of
   128 columns, so every row is 16-byte aligned although V = 50257 is odd;
2. one HIP kernel that reads each logits row once, produces the per-row loss
   and overwrites the row with ``(softmax - onehot) / n_valid`` — the gradient
   with respect to the logits is already in place when backward starts;
3. in backward, the two GEMMs for ``dhidden`` and ``dweight``; when the (tied) weight belongs to
   a DDP bucket or an FSDP unit, ``dweight`` is produced straight into the owner's gradient
   buffer (gradient sink) and the embedding backward later adds its rows to the same buffer.

fp32 logits (1.65 GB at B=8, T=1024 for GPT-2) are never materialised.

Row chunks. The head can run in ROW CHUNKS through one reusable logits buffer (0.83 GB per 8192 rows
of GPT-2's vocabulary instead of 6.6 GB at 64 x 1024 tokens): per chunk the logits GEMM, the loss
kernel (which leaves the logits gradient in place) and — when gradients are wanted — that chunk's
two backward GEMMs follow at once, the weight gradient accumulating in one fp32 [V, d] buffer.
Backward then only applies the incoming scalar gradient: to the kept ``dhidden`` and while adding the
fp32 weight gradient to its target. Measured on GPT-2 Large at 64 x 1024 tokens per step (MI355X):
5.4 GB less peak memory for 1.9 % more step time (eight M = 8192 GEMM triples and eight read-modify-
write passes over the fp32 weight gradient instead of one of each), so chunking is a MEMORY policy:
``PDA_LM_HEAD_CHUNK_ROWS=<rows>`` forces it, ``0`` forbids it, and by default (``auto``) the head is
chunked into 8192-row pieces only when its one-piece logits buffer would take more than a sixteenth
of the device memory that is free at that moment.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import _ext
from .linear import _grad_target, input_grad_mm, planned_mm

_PAD = 128


_AUTO_CHUNK_ROWS = 8192


def _chunk_rows(rows: int, padded_cols: int, elem_size: int, device) -> int:
    """Rows per chunk; 0 = one piece (see the module docstring for the policy)."""
    import os
    setting = os.environ.get("PDA_LM_HEAD_CHUNK_ROWS", "auto")
    if setting != "auto":
        n = int(setting)
        return max(1024, n) if n > 0 else 0
    if rows <= _AUTO_CHUNK_ROWS:
        return 0
    free, _ = torch.cuda.mem_get_info(device)
    return _AUTO_CHUNK_ROWS if rows * padded_cols * elem_size > free // 16 else 0


def cross_entropy_reference(logits: torch.Tensor, targets: torch.Tensor, ignore_index: int = -100):
    return F.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), targets.reshape(-1),
                           ignore_index=ignore_index)


def lm_head_cross_entropy_reference(hidden, weight, targets, ignore_index=-100):
    logits = F.linear(hidden, weight)
    return cross_entropy_reference(logits, targets, ignore_index)


def _padded_cols(v: int) -> int:
    return (v + _PAD - 1) // _PAD * _PAD


class _CrossEntropyFn(torch.autograd.Function):
    """Cross-entropy over given logits; the logits buffer is consumed (overwritten)."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        ext = _ext.require()
        V = logits.shape[-1]
        rows = logits.numel() // V
        vp = _padded_cols(V)
        if logits.is_contiguous() and V == vp:
            buf = logits.view(rows, V)
        else:
            buf = torch.empty(rows, vp, dtype=logits.dtype, device=logits.device)[:, :V]
            buf.copy_(logits.reshape(rows, V))
        tgt = targets.reshape(-1).contiguous()
        n_valid = (tgt != ignore_index).sum().clamp_(min=1).to(torch.float32)
        scale = n_valid.reciprocal().reshape(1)
        row_loss = ext.cross_entropy_fwd_bwd(buf, tgt, V, scale, ignore_index)
        ctx.save_for_backward(buf)
        ctx.shape = logits.shape
        return row_loss.sum() * scale[0]

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        return (dlogits * dloss).reshape(ctx.shape), None, None


class _LMHeadCrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, weight, targets, ignore_index):
        ext = _ext.require()
        V, d = weight.shape
        h2 = hidden.reshape(-1, d)
        rows = h2.shape[0]
        vp = _padded_cols(V)
        logits = torch.empty(rows, vp, dtype=hidden.dtype, device=hidden.device)
        planned_mm(h2, weight, trans_b=True, out=logits[:, :V])
        tgt = targets.reshape(-1).contiguous()
        n_valid = (tgt != ignore_index).sum().clamp_(min=1).to(torch.float32)
        scale = n_valid.reciprocal().reshape(1)
        row_loss = ext.cross_entropy_fwd_bwd(logits[:, :V], tgt, V, scale, ignore_index)
        ctx.save_for_backward(logits, h2, weight)
        ctx.hidden_shape = hidden.shape
        ctx.grad_sink = getattr(weight, "_pda_grad_sink", None)
        return row_loss.sum() * scale[0]

    @staticmethod
    def backward(ctx, dloss):
        dlogits_p, h2, weight = ctx.saved_tensors
        V = weight.shape[0]
        dlogits = dlogits_p[:, :V]
        dhidden = dweight = None
        if ctx.needs_input_grad[0]:
            dhidden = input_grad_mm(dlogits, weight, dlogits_p).mul_(dloss.to(weight.dtype)).view(ctx.hidden_shape)
        if ctx.needs_input_grad[1]:
            target = _grad_target(weight, ctx.grad_sink)
            if _native_lm_head_wgrad_ok(dlogits_p, h2, V):
                # Native weight-gradient kernel over the padded buffer (its padding columns are zero), only
                # the V real rows folded — written or accumulated straight into the owner's slot; dloss is
                # applied to the small operand.
                out, acc = target if target is not None and target[0].is_contiguous() else (None, False)
                dw, _ = ext_wgrad(dlogits_p, h2 * dloss.to(weight.dtype), out, acc, V)
                dweight = dw if out is None else None
                if target is not None and out is None:
                    target[0].add_(dweight)
                    dweight = None
            elif target is not None and not target[1] and target[0].is_contiguous():
                # first writer of the owner's slot in this pass: the GEMM writes it, dloss is applied to
                # the small operand instead of a pass over the [V, d] result
                torch.mm(dlogits.t(), h2 * dloss.to(weight.dtype), out=target[0])
            else:
                dweight = torch.mm(dlogits.t(), h2).mul_(dloss.to(weight.dtype))
                if target is not None:
                    target[0].add_(dweight)
                    dweight = None
        return dhidden, dweight, None, None


class _ChunkedLMHeadCrossEntropyFn(torch.autograd.Function):
    """The same loss with the vocabulary-wide buffers cut into row chunks (see the module docstring)."""

    @staticmethod
    def forward(ctx, hidden, weight, targets, ignore_index, need_grad, chunk):
        ext = _ext.require()
        V, d = weight.shape
        h2 = hidden.reshape(-1, d)
        rows = h2.shape[0]
        vp = _padded_cols(V)
        tgt = targets.reshape(-1).contiguous()
        n_valid = (tgt != ignore_index).sum().clamp_(min=1).to(torch.float32)
        scale = n_valid.reciprocal().reshape(1)
        buf = torch.empty(min(chunk, rows), vp, dtype=hidden.dtype, device=hidden.device)
        loss = torch.zeros((), dtype=torch.float32, device=hidden.device)
        dhidden = torch.empty_like(h2) if need_grad and hidden.requires_grad else None
        want_w = need_grad and weight.requires_grad
        dw = None
        for lo in range(0, rows, chunk):
            n = min(chunk, rows - lo)
            logits_p = buf[:n]
            hc = h2[lo:lo + n]
            planned_mm(hc, weight, trans_b=True, out=logits_p[:, :V])
            loss += ext.cross_entropy_fwd_bwd(logits_p[:, :V], tgt[lo:lo + n], V, scale, ignore_index).sum()
            if dhidden is not None:  # logits_p now holds the gradient (zero in the padding columns)
                dhidden[lo:lo + n] = input_grad_mm(logits_p[:, :V], weight, logits_p)
            if want_w:
                if _native_lm_head_wgrad_ok(logits_p, hc, V):
                    if dw is None:
                        dw = torch.empty(V, d, dtype=torch.float32, device=hidden.device)
                        ext_wgrad(logits_p, hc, dw, False, V)
                    else:
                        ext_wgrad(logits_p, hc, dw, True, V)
                else:
                    part = torch.mm(logits_p[:, :V].t(), hc).float()
                    dw = part if dw is None else dw.add_(part)
        ctx.save_for_backward(*[t for t in (dhidden, dw) if t is not None])
        ctx.have = (dhidden is not None, dw is not None)
        ctx.hidden_shape = hidden.shape
        ctx.weight = weight  # consulted for .grad / its gradient sink only
        ctx.grad_sink = getattr(weight, "_pda_grad_sink", None)
        return loss * scale[0]

    @staticmethod
    def backward(ctx, dloss):
        if getattr(ctx, "consumed", False):
            raise RuntimeError("lm_head_cross_entropy (chunked): backward ran twice; its kept gradients are "
                               "scaled in place and cannot be reused")
        ctx.consumed = True
        kept = list(ctx.saved_tensors)
        dhidden = kept.pop(0) if ctx.have[0] else None
        dw = kept.pop(0) if ctx.have[1] else None
        weight = ctx.weight
        dweight = None
        if dhidden is not None and ctx.needs_input_grad[0]:
            dhidden = dhidden.mul_(dloss.to(dhidden.dtype)).view(ctx.hidden_shape)
        else:
            dhidden = None
        if dw is not None and ctx.needs_input_grad[1]:
            target = _grad_target(weight, ctx.grad_sink)
            s = dloss.to(torch.float32)
            if target is None:
                dweight = dw.mul_(s).to(weight.dtype)
            else:
                out, acc = target
                if acc and out.dtype == torch.float32:
                    out.addcmul_(dw, s)
                elif acc:
                    out.add_(dw.mul_(s).to(out.dtype))
                elif out.dtype == torch.float32:
                    torch.mul(dw, s, out=out)
                else:
                    out.copy_(dw.mul_(s))
        return dhidden, dweight, None, None, None, None


def _native_lm_head_wgrad_ok(dlogits_p, h2, V) -> bool:
    import os
    if os.environ.get("PDA_NO_NATIVE_WGRAD") == "1" or dlogits_p.dtype != torch.bfloat16:
        return False
    M, vp = dlogits_p.shape
    d = h2.shape[1]
    return (dlogits_p.is_contiguous() and h2.is_contiguous() and (V * d) % 4 == 0
            and _ext.require().wgrad_supported(M, vp, d))


def ext_wgrad(dlogits_p, h2, out, accumulate, rows):
    return _ext.require().linear_wgrad(dlogits_p, h2, False, out, accumulate, None, False, rows)


def cross_entropy(logits: torch.Tensor, targets: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """Mean cross-entropy of ``logits[..., V]`` against integer ``targets``."""
    if _ext.use_hip(logits):
        return _CrossEntropyFn.apply(logits, targets, ignore_index)
    return cross_entropy_reference(logits, targets, ignore_index)


def lm_head_cross_entropy(hidden: torch.Tensor, weight: torch.Tensor, targets: torch.Tensor,
                          ignore_index: int = -100) -> torch.Tensor:
    """``cross_entropy(hidden @ weight.T, targets)`` without fp32 / unpadded logits."""
    if _ext.use_hip(hidden):
        rows = hidden.numel() // hidden.shape[-1]
        chunk = _chunk_rows(rows, _padded_cols(weight.shape[0]), hidden.element_size(), hidden.device)
        if chunk and rows > chunk:
            need_grad = torch.is_grad_enabled() and (hidden.requires_grad or weight.requires_grad)
            return _ChunkedLMHeadCrossEntropyFn.apply(hidden, weight, targets, ignore_index, need_grad, chunk)
        return _LMHeadCrossEntropyFn.apply(hidden, weight, targets, ignore_index)
    return lm_head_cross_entropy_reference(hidden, weight, targets, ignore_index)
