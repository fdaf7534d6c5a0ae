"""Numerics of every HIP kernel against a plain PyTorch fp32 reference of the same op."""
import math

import pytest
import torch

from pytorch_distributed_amd import _ext, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ext_loaded():
    return _ext.require()


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


# --------------------------------------------------------------------------- norms
@pytest.mark.parametrize("d", [768, 1024, 1280, 1600, 4096])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_fwd_bwd(d, dtype, with_res):
    _ext_loaded()
    torch.manual_seed(0)
    rows = 517  # odd on purpose: exercises the row tail
    x = torch.randn(rows, d, device=DEV, dtype=dtype)
    res = torch.randn(rows, d, device=DEV, dtype=dtype) if with_res else None
    w = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(d, device=DEV)).to(dtype)
    dy = torch.randn(rows, d, device=DEV, dtype=dtype)
    dh = torch.randn(rows, d, device=DEV, dtype=dtype)

    def run(fn, xs):
        leaves = [t.detach().clone().requires_grad_(True) if t is not None else None for t in xs]
        y, h = fn(leaves[0], leaves[1], leaves[2], leaves[3], 1e-5)
        (y.float() * dy.float()).sum().add((h.float() * dh.float()).sum()).backward()
        return y, h, [t.grad if t is not None else None for t in leaves]

    y, h, g = run(ops.fused_add_layer_norm, [x, res, w, b])
    # fp32 oracle on the same (already rounded) inputs
    xs32 = [t.float() if t is not None else None for t in (x, res, w, b)]
    y_ref, h_ref, g_ref = run(ops.layer_norm_reference, xs32)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(y, y_ref) < tol
    assert rel_err(h, h_ref) < tol
    for a, r in zip(g, g_ref):
        if a is not None:
            assert rel_err(a, r) < tol, (d, dtype, with_res)


@pytest.mark.parametrize("d", [256, 2048, 4096])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rmsnorm_fwd_bwd(d, dtype):
    _ext_loaded()
    torch.manual_seed(1)
    rows = 300
    x = torch.randn(rows, d, device=DEV, dtype=dtype)
    res = torch.randn(rows, d, device=DEV, dtype=dtype)
    w = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype)
    dy = torch.randn(rows, d, device=DEV, dtype=dtype)

    def run(fn, xs):
        leaves = [t.detach().clone().requires_grad_(True) for t in xs]
        y, h = fn(leaves[0], leaves[1], leaves[2], 1e-6)
        (y.float() * dy.float()).sum().add(h.float().sum()).backward()
        return y, h, [t.grad for t in leaves]

    y, h, g = run(ops.fused_add_rms_norm, [x, res, w])
    y_ref, h_ref, g_ref = run(ops.rms_norm_reference, [x.float(), res.float(), w.float()])
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(y, y_ref) < tol and rel_err(h, h_ref) < tol
    for a, r in zip(g, g_ref):
        assert rel_err(a, r) < tol


# --------------------------------------------------------------------------- activations
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n", [8 * 1024 * 5120 // 64, 1000003])
def test_gelu(dtype, n):
    _ext_loaded()
    torch.manual_seed(2)
    u = (3 * torch.randn(n, device=DEV)).to(dtype)
    dy = torch.randn(n, device=DEV, dtype=dtype)
    a = u.clone().requires_grad_(True)
    y = ops.gelu_new(a)
    y.backward(dy)
    r = u.float().requires_grad_(True)
    y_ref = ops.gelu_new_reference(r)
    y_ref.backward(dy.float())
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-6
    assert rel_err(y, y_ref) < tol
    assert rel_err(a.grad, r.grad) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_swiglu(dtype):
    _ext_loaded()
    torch.manual_seed(3)
    x = (2 * torch.randn(333, 2 * 1376, device=DEV)).to(dtype)
    dy = torch.randn(333, 1376, device=DEV, dtype=dtype)
    a = x.clone().requires_grad_(True)
    y = ops.swiglu(a)
    y.backward(dy)
    r = x.float().requires_grad_(True)
    y_ref = ops.swiglu_reference(r)
    y_ref.backward(dy.float())
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-6
    assert rel_err(y, y_ref) < tol
    assert rel_err(a.grad, r.grad) < tol


# --------------------------------------------------------------------------- cross entropy
@pytest.mark.parametrize("vocab", [50257, 32000, 512, 131071])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cross_entropy_logits(vocab, dtype):
    _ext_loaded()
    torch.manual_seed(4)
    rows = 257
    logits = (4 * torch.randn(rows, vocab, device=DEV)).to(dtype)
    targets = torch.randint(0, vocab, (rows,), device=DEV)
    targets[::7] = -100
    a = logits.clone().requires_grad_(True)
    loss = ops.cross_entropy(a, targets)
    (loss * 0.5).backward()
    r = logits.float().requires_grad_(True)
    loss_ref = ops.cross_entropy_reference(r, targets)
    (loss_ref * 0.5).backward()
    assert abs(loss.item() - loss_ref.item()) < (2e-3 if dtype == torch.bfloat16 else 1e-5) * max(1.0, abs(loss_ref.item()))
    assert rel_err(a.grad, r.grad) < (1e-2 if dtype == torch.bfloat16 else 1e-5)


@pytest.mark.parametrize("vocab,d", [(50257, 768), (1000, 256)])
def test_lm_head_cross_entropy(vocab, d):
    _ext_loaded()
    torch.manual_seed(5)
    rows = 512
    h = torch.randn(rows, d, device=DEV, dtype=torch.bfloat16)
    w = (0.05 * torch.randn(vocab, d, device=DEV)).to(torch.bfloat16)
    targets = torch.randint(0, vocab, (rows,), device=DEV)
    ha, wa = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
    loss = ops.lm_head_cross_entropy(ha, wa, targets)
    loss.backward()
    hr, wr = h.float().requires_grad_(True), w.float().requires_grad_(True)
    loss_ref = ops.lm_head_cross_entropy_reference(hr, wr, targets)
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 5e-3 * abs(loss_ref.item())
    assert rel_err(ha.grad, hr.grad) < 2e-2
    assert rel_err(wa.grad, wr.grad) < 2e-2


@pytest.mark.parametrize("existing_grad", [None, torch.bfloat16])
def test_lm_head_cross_entropy_in_row_chunks(monkeypatch, existing_grad):
    """Above the chunk size the head runs through one reusable logits buffer: same loss and
    gradients as the one-piece path and as the fp32 reference (V = 50257, ragged last chunk,
    ignored targets, a scaled loss, and accumulation into an existing weight gradient)."""
    _ext_loaded()
    monkeypatch.setenv("PDA_LM_HEAD_CHUNK_ROWS", "1024")
    torch.manual_seed(15)
    vocab, d, rows = 50257, 256, 2 * 1024 + 452
    h = torch.randn(rows, d, device=DEV, dtype=torch.bfloat16)
    w = (0.05 * torch.randn(vocab, d, device=DEV)).to(torch.bfloat16)
    targets = torch.randint(0, vocab, (rows,), device=DEV)
    targets[::11] = -100
    prior = 0.01 * torch.randn(vocab, d, device=DEV)

    def run(chunked):
        monkeypatch.setenv("PDA_LM_HEAD_CHUNK_ROWS", "1024" if chunked else "0")
        ha, wa = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
        if existing_grad is not None:
            wa.grad = prior.to(existing_grad)
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = ops.lm_head_cross_entropy(ha, wa, targets)
        (loss * 0.25).backward()
        torch.cuda.synchronize()
        return loss.item(), ha.grad, wa.grad, torch.cuda.max_memory_allocated() - base

    loss_c, dh_c, dw_c, peak_c = run(True)
    loss_1, dh_1, dw_1, peak_1 = run(False)
    hr, wr = h.float().requires_grad_(True), w.float().requires_grad_(True)
    if existing_grad is not None:
        wr.grad = prior.to(existing_grad).float()
    loss_ref = ops.lm_head_cross_entropy_reference(hr, wr, targets)
    (loss_ref * 0.25).backward()
    assert abs(loss_c - loss_1) < 1e-4 * abs(loss_1) and abs(loss_c - loss_ref.item()) < 5e-3 * abs(loss_ref.item())
    assert rel_err(dh_c, dh_1) < 1e-3 and rel_err(dh_c, hr.grad) < 2e-2
    assert rel_err(dw_c, dw_1) < 1e-2 and rel_err(dw_c, wr.grad) < 2e-2
    del peak_c, peak_1  # (at this toy size the weight-gradient kernel's slice workspace outweighs the logits)


# --------------------------------------------------------------------------- AdamW
def test_fused_adamw_cached_lists_keep_step_counts_and_state_layout():
    """From the second step on the optimizer runs from cached argument lists with one step integer
    per group: the checkpoint layout still shows per-parameter step counts, a step in which one
    parameter has no gradient falls back (its count stays behind), and a reloaded state continues."""
    _ext_loaded()
    from pytorch_distributed_amd.optim import FusedAdamW

    torch.manual_seed(16)
    shapes = [(4096,), (33, 7), (1280,)]
    params = [torch.randn(s, device=DEV, dtype=torch.bfloat16).requires_grad_(True) for s in shapes]
    ref_params = [p.detach().float().clone().requires_grad_(True) for p in params]
    opt = FusedAdamW(params, lr=1e-3, weight_decay=0.1)
    ref = torch.optim.AdamW(ref_params, lr=1e-3, weight_decay=0.1)

    def one(skip=None):
        for i, (p, r) in enumerate(zip(params, ref_params)):
            if i == skip:
                p.grad = r.grad = None
                continue
            g = torch.randn_like(r)
            p.grad = g.to(torch.bfloat16)
            r.grad = p.grad.float()
        opt.step(zero_grad=True)
        ref.step()

    for _ in range(4):
        one()
    assert 0 in opt._fast and opt._fast[0]["pending"] == 3          # steps 2-4 ran from the cache
    sd = opt.state_dict()
    assert [float(sd["state"][i]["step"]) for i in range(3)] == [4.0, 4.0, 4.0]
    one(skip=1)                                                       # falls back to the general path
    one()
    sd = opt.state_dict()
    assert [float(sd["state"][i]["step"]) for i in range(3)] == [6.0, 5.0, 6.0]
    opt2 = FusedAdamW(params, lr=1e-3, weight_decay=0.1)
    opt2.load_state_dict(sd)
    opt, before = opt2, opt
    one()
    one()
    assert [float(opt.state_dict()["state"][i]["step"]) for i in range(3)] == [8.0, 7.0, 8.0]
    del before
    for p, r in zip(params, ref_params):
        assert rel_err(opt.state[p]["master"], r.detach()) < 2e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_adamw_matches_torch(dtype):
    _ext_loaded()
    from pytorch_distributed_amd.optim import FusedAdamW

    torch.manual_seed(6)
    shapes = [(1 << 21,), (1023, 7), (50257, 64), (5,), (1280,), (3, 3, 3)]
    ref_params = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    params = [p.detach().to(dtype).clone().requires_grad_(True) for p in ref_params]
    if dtype == torch.bfloat16:  # the fp32 reference starts from the rounded values too
        ref_params = [p.detach().float().clone().requires_grad_(True) for p in params]
    opt = FusedAdamW(params, lr=3e-4, weight_decay=0.1)
    ref = torch.optim.AdamW(ref_params, lr=3e-4, weight_decay=0.1)
    for step in range(5):
        for p, r in zip(params, ref_params):
            g = torch.randn_like(r)
            r.grad = g.clone()
            p.grad = g.to(dtype)
            if dtype == torch.bfloat16:
                r.grad = p.grad.float()
        # even steps clear the gradients: dropped on step 0, zeroed inside the kernel on 2 and 4
        opt.step(zero_grad=(step % 2 == 0), set_to_none=(step == 0))
        ref.step()
        if step == 0:
            assert all(p.grad is None for p in params)
        elif step % 2 == 0:
            assert all(float(p.grad.abs().max()) == 0.0 for p in params)
    for p, r in zip(params, ref_params):
        master = opt.state[p].get("master", p.data)
        assert rel_err(master, r) < 1e-6
        assert rel_err(opt.state[p]["exp_avg"], ref.state[r]["exp_avg"]) < 5e-5
        assert rel_err(opt.state[p]["exp_avg_sq"], ref.state[r]["exp_avg_sq"]) < 5e-5
        if dtype == torch.bfloat16:
            assert torch.equal(p.data, master.to(torch.bfloat16))


# --------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,T,H,D", [(2, 1024, 4, 64), (1, 333, 3, 64), (2, 128, 2, 128), (1, 640, 2, 128),
                                     (1, 64, 1, 64), (3, 1000, 5, 64)])
def test_attention_forward(B, T, H, D):
    ext = _ext_loaded()
    torch.manual_seed(7)
    qkv = torch.randn(B, T, 3 * H * D, device=DEV, dtype=torch.bfloat16)
    out, lse = ext.attention_forward(qkv, H, 1.0 / math.sqrt(D))
    ref = ops.attention_reference(qkv.float(), H)
    assert rel_err(out, ref) < 1e-2, rel_err(out, ref)
    q, k, _ = qkv.float().view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(D)
    s = s.masked_fill(~torch.ones(T, T, dtype=torch.bool, device=DEV).tril(), float("-inf"))
    assert (lse - torch.logsumexp(s, -1)).abs().max().item() < 2e-2


def test_attention_forward_structured():
    """Exact-integer check of the MFMA operand maps: V carries an asymmetric pattern and
    attention is made (near) one-hot, so a swapped row/column or a permuted k order cannot pass."""
    ext = _ext_loaded()
    B, T, H, D = 1, 256, 2, 64
    q = torch.zeros(B, T, H, D, device=DEV)
    k = torch.zeros(B, T, H, D, device=DEV)
    t = torch.arange(T, device=DEV)
    # query t attends (with a huge logit) to key max(t - 3, 0): near one-hot rows.
    tgt = (t - 3).clamp(min=0)
    # position p is encoded as a +-1 code of 8 bits in the first 8 dims; q carries the code of
    # its target key scaled up, so the matching key wins the softmax by a wide margin
    bits = ((t[:, None] >> torch.arange(8, device=DEV)) & 1).float() * 2 - 1  # [T, 8]
    k[0, :, :, :8] = bits[:, None, :]
    q[0, :, :, :8] = bits[tgt][:, None, :] * 24.0
    v = (t[:, None, None] * 1.0 + torch.arange(D, device=DEV)[None, None, :] * 0.25 +
         torch.arange(H, device=DEV)[None, :, None] * 1000.0).expand(T, H, D)[None]
    qkv = torch.stack([q, k, v.contiguous()], dim=2).reshape(B, T, 3 * H * D).to(torch.bfloat16)
    out, _ = ext.attention_forward(qkv, H, 1.0)
    ref = ops.attention_reference(qkv.float(), H, scale=1.0)
    assert (out.float() - ref).abs().max().item() <= 8.0  # bf16 resolution at magnitude ~1000-2000
    assert rel_err(out, ref) < 5e-3


@pytest.mark.parametrize("B,T,H,D", [(2, 1024, 4, 64), (1, 333, 3, 64), (2, 256, 2, 128), (1, 640, 2, 128),
                                     (1, 64, 1, 64), (2, 1000, 5, 64)])
def test_attention_backward(B, T, H, D):
    _ext_loaded()
    torch.manual_seed(8)
    qkv = torch.randn(B, T, 3 * H * D, device=DEV, dtype=torch.bfloat16)
    dout = torch.randn(B, T, H * D, device=DEV, dtype=torch.bfloat16)
    a = qkv.clone().requires_grad_(True)
    out = ops.causal_attention(a, H)
    out.backward(dout)
    r = qkv.float().requires_grad_(True)
    ref = ops.attention_reference(r, H)
    ref.backward(dout.float())
    assert rel_err(out, ref) < 1e-2
    d = 3 * H * D
    g, gr = a.grad.view(B, T, 3, H * D), r.grad.view(B, T, 3, H * D)
    for i, name in enumerate(("dq", "dk", "dv")):
        e = rel_err(g[:, :, i], gr[:, :, i])
        assert e < 2e-2, (name, e)


def test_attention_backward_deterministic():
    _ext_loaded()
    torch.manual_seed(9)
    B, T, H, D = 2, 512, 4, 64
    qkv = torch.randn(B, T, 3 * H * D, device=DEV, dtype=torch.bfloat16)
    dout = torch.randn(B, T, H * D, device=DEV, dtype=torch.bfloat16)
    grads = []
    for _ in range(2):
        a = qkv.clone().requires_grad_(True)
        ops.causal_attention(a, H).backward(dout)
        grads.append(a.grad.clone())
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("B,T,H,D,p", [(2, 512, 3, 64, 0.1), (1, 333, 2, 64, 0.25), (1, 256, 2, 128, 0.1)])
def test_attention_dropout_matches_reference_with_same_mask(B, T, H, D, p):
    """Forward and backward with in-kernel dropout against the eager op sequence using the mask the
    kernels generate (materialised by attention_dropout_mask from the same seed)."""
    ext = _ext_loaded()
    from pytorch_distributed_amd.ops.attention import dropout_quantised

    torch.manual_seed(10)
    seed = 123456789012345
    qkv = torch.randn(B, T, 3 * H * D, device=DEV, dtype=torch.bfloat16)
    dout = torch.randn(B, T, H * D, device=DEV, dtype=torch.bfloat16)
    keep = ext.attention_dropout_mask(B, H, T, p, seed, torch.device(DEV))
    p_q, keep_scale = dropout_quantised(p)
    causal = torch.ones(T, T, dtype=torch.bool, device=DEV).tril()
    frac_dropped = 1.0 - keep[:, :, causal].float().mean().item()
    assert abs(frac_dropped - p_q) < 0.01, (frac_dropped, p_q)

    a = qkv.clone().requires_grad_(True)
    out = ops.causal_attention(a, H, dropout_p=p, training=True, seed=seed)
    out.backward(dout)
    r = qkv.float().requires_grad_(True)
    ref = ops.attention_reference(r, H, keep_mask=keep, keep_scale=keep_scale)
    ref.backward(dout.float())
    assert rel_err(out, ref) < 1.5e-2, rel_err(out, ref)
    g, gr = a.grad.view(B, T, 3, H * D), r.grad.view(B, T, 3, H * D)
    for i, name in enumerate(("dq", "dk", "dv")):
        e = rel_err(g[:, :, i], gr[:, :, i])
        assert e < 2.5e-2, (name, e)
    # a different seed gives a different mask; eval mode ignores dropout
    other = ops.causal_attention(qkv, H, dropout_p=p, training=True, seed=seed + 1)
    assert not torch.equal(other, out.detach())
    assert torch.equal(ops.causal_attention(qkv, H, dropout_p=p, training=False), ops.causal_attention(qkv, H))


def test_attention_dropout_follows_torch_seed():
    _ext_loaded()
    qkv = torch.randn(1, 128, 3 * 2 * 64, device=DEV, dtype=torch.bfloat16)
    torch.manual_seed(3)
    a = ops.causal_attention(qkv, 2, dropout_p=0.2, training=True)
    b = ops.causal_attention(qkv, 2, dropout_p=0.2, training=True)
    torch.manual_seed(3)
    c = ops.causal_attention(qkv, 2, dropout_p=0.2, training=True)
    assert torch.equal(a, c) and not torch.equal(a, b)


@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 128), (torch.bfloat16, 64), (torch.float32, 64)])
def test_rope_kernel_matches_reference(dtype, D):
    """In-place rotary kernel on the packed QKV activation, forward and backward, against the
    PyTorch rotate-half formula in fp32."""
    _ext_loaded()
    from pytorch_distributed_amd.models.llama import _RopePacked, _rotate, rope_tables

    torch.manual_seed(11)
    B, T, H = 2, 96, 3
    cos, sin = rope_tables(128, D, 10000.0, DEV)
    base = torch.randn(B, T, 3 * H * D, device=DEV, dtype=dtype)
    dout = torch.randn(B, T, 3 * H * D, device=DEV, dtype=dtype)
    x = base.clone().requires_grad_(True)
    out = _RopePacked.apply(x * 1.0, cos, sin, H)  # x * 1.0: a non-leaf the op may rotate in place
    out.backward(dout)
    v = base.float().view(B, T, 3, H, D)
    ref = v.clone()
    ref[:, :, :2] = _rotate(v[:, :, :2], cos[:T], sin[:T], inverse=False)
    g = dout.float().view(B, T, 3, H, D)
    gref = g.clone()
    gref[:, :, :2] = _rotate(g[:, :, :2], cos[:T], sin[:T], inverse=True)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(out, ref.view(B, T, -1)) < tol
    assert rel_err(x.grad, gref.view(B, T, -1)) < tol
    assert torch.equal(out.view(B, T, 3, H, D)[:, :, 2], base.view(B, T, 3, H, D)[:, :, 2])


@pytest.mark.parametrize("rows,cols,pad", [(1280, 5120, 8), (3840, 1280, 8), (50257, 1280, 50304), (72, 264, 8), (100, 64, 128)])
def test_transpose_kernel(rows, cols, pad):
    ext = _ext.require()
    src = torch.randn(rows, cols, device=DEV, dtype=torch.bfloat16)
    out = ext.transpose_2d(src, pad)
    rp = -(-rows // pad) * pad
    assert out.shape == (cols, rp)
    assert torch.equal(out[:, :rows], src.t())
    assert not out[:, rows:].any()
    view = torch.randn(rows, cols + 16, device=DEV, dtype=torch.bfloat16)[:, 8:8 + cols]  # pitched source
    if view.data_ptr() % 16 == 0:
        assert torch.equal(ext.transpose_2d(view, pad)[:, :rows], view.t())


def test_input_grad_layout_choice_matches_direct(monkeypatch):
    """Whichever route the timing picks, the input gradient equals the direct product (the two
    routes run the same contraction; only the operand layout differs)."""
    import importlib

    L = importlib.import_module("pytorch_distributed_amd.ops.linear")

    dy = torch.randn(4096, 3840, device=DEV, dtype=torch.bfloat16)
    w = torch.randn(3840, 1280, device=DEV, dtype=torch.bfloat16) * 0.02
    ref = (dy.float() @ w.float())
    got = L.input_grad_mm(dy, w)
    assert (dy.device.index, 4096, 3840, 1280) in L._DGRAD_TRANSPOSED
    assert rel_err(got, ref) < 1e-2
    for forced in (0, 1):  # direct, transposed weight
        monkeypatch.setitem(L._DGRAD_TRANSPOSED, (dy.device.index, 4096, 3840, 1280), forced)
        assert rel_err(L.input_grad_mm(dy, w), ref) < 1e-2
    # padded route (LM head): N = 1001 valid columns in a 1024-column buffer whose padding is zero
    buf = torch.zeros(4096, 1024, device=DEV, dtype=torch.bfloat16)
    buf[:, :1001] = torch.randn(4096, 1001, device=DEV, dtype=torch.bfloat16)
    w2 = torch.randn(1001, 768, device=DEV, dtype=torch.bfloat16) * 0.02
    ref2 = buf[:, :1001].float() @ w2.float()
    for forced in (0, 1):
        monkeypatch.setitem(L._DGRAD_TRANSPOSED, (dy.device.index, 4096, 1001, 768), forced)
        assert rel_err(L.input_grad_mm(buf[:, :1001], w2, buf), ref2) < 1e-2


@pytest.mark.parametrize("has_res", [True, False])
@pytest.mark.parametrize("d", [768, 1280, 1600])
def test_layer_norm_fused_dropout(d, has_res):
    """LN(dropout(x) + residual): the mask is recovered from the emitted sum, then forward and
    backward are compared with PyTorch on that mask (the backward kernel regenerates it)."""
    from pytorch_distributed_amd.ops.attention import dropout_quantised

    torch.manual_seed(0)
    rows, p = 512, 0.1
    pq, scale = dropout_quantised(p)
    x = torch.randn(rows, d, device=DEV, dtype=torch.bfloat16).requires_grad_()
    res = torch.randn(rows, d, device=DEV, dtype=torch.bfloat16).requires_grad_() if has_res else None
    w = (1 + 0.1 * torch.randn(d, device=DEV)).to(torch.bfloat16).requires_grad_()
    b = (0.1 * torch.randn(d, device=DEV)).to(torch.bfloat16).requires_grad_()
    y, h = ops.fused_add_layer_norm(x, res, w, b, 1e-5, dropout_p=p, training=True, seed=1234)
    base = res.detach().float() if has_res else torch.zeros_like(h, dtype=torch.float32)
    # The mask depends on (seed, row, column) only: read it off a call without residual, where a
    # dropped element is exactly zero and a kept one is not.
    _, h0 = ops.fused_add_layer_norm(x.detach(), None, w.detach(), b.detach(), 1e-5, dropout_p=p, training=True, seed=1234)
    keep = h0 != 0
    kept = keep.float().mean().item()
    assert abs(kept - (1 - pq)) < 0.01, kept
    xr, wr, br = (t.detach().float().requires_grad_() for t in (x, w, b))
    rr = res.detach().float().requires_grad_() if has_res else None
    h_ref = xr * keep * scale + (rr if has_res else 0)
    h_ref_q = h_ref.detach().to(torch.bfloat16).float() + (h_ref - h_ref.detach())  # bf16-rounded value, identity gradient
    y_ref = torch.nn.functional.layer_norm(h_ref_q, (d,), wr, br, 1e-5)
    assert rel_err(h, h_ref) < 5e-3 and rel_err(y, y_ref) < 1e-2
    gy, gh = torch.randn_like(y), torch.randn_like(h)
    torch.autograd.backward([y, h], [gy, gh])
    torch.autograd.backward([y_ref, h_ref], [gy.float(), gh.float()])
    assert rel_err(x.grad, xr.grad) < 1e-2
    assert (x.grad[~keep] == 0).all()
    if has_res:
        assert rel_err(res.grad, rr.grad) < 1e-2
    assert rel_err(w.grad, wr.grad) < 2e-2 and rel_err(b.grad, br.grad) < 2e-2
    # same seed -> same mask, another seed -> another mask, eval -> no dropout
    _, h2 = ops.fused_add_layer_norm(x.detach(), res.detach() if has_res else None, w.detach(), b.detach(), 1e-5,
                                     dropout_p=p, training=True, seed=1234)
    _, h3 = ops.fused_add_layer_norm(x.detach(), res.detach() if has_res else None, w.detach(), b.detach(), 1e-5,
                                     dropout_p=p, training=True, seed=99)
    _, h4 = ops.fused_add_layer_norm(x.detach(), res.detach() if has_res else None, w.detach(), b.detach(), 1e-5,
                                     dropout_p=p, training=False)
    assert torch.equal(h2, h.detach()) and not torch.equal(h3, h.detach())
    assert rel_err(h4, x.detach().float() + base) < 5e-3


def test_gpt2_trains_with_dropout():
    """All three dropouts of the reference configuration active (p = 0.1): loss falls, and eval
    mode is deterministic."""
    from pytorch_distributed_amd.models import MyGPT2LMHeadModel, get_config
    from pytorch_distributed_amd.optim import FusedAdamW

    cfg = get_config("gpt2", n_layer=2)
    assert cfg.resid_pdrop == 0.1 and cfg.attn_pdrop == 0.1 and cfg.embd_pdrop == 0.1
    torch.manual_seed(3)
    model = MyGPT2LMHeadModel(cfg).to(device=DEV, dtype=torch.bfloat16)
    ids = torch.randint(0, cfg.vocab_size, (4, 257), device=DEV)
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
    losses = []
    for _ in range(10):
        loss = model(ids[:, :-1], ids[:, 1:])
        loss.backward()
        opt.step(zero_grad=True)
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 1.0, losses
    a, b = model(ids[:, :-1], ids[:, 1:]).item(), model(ids[:, :-1], ids[:, 1:]).item()
    assert a != b  # training mode: fresh masks every call
    model.eval()
    with torch.no_grad():
        assert model(ids[:, :-1], ids[:, 1:]).item() == model(ids[:, :-1], ids[:, 1:]).item()


# ------------------------------------------------------------------ embedding
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("V,d,B,T", [(50257, 768, 4, 1024), (512, 128, 3, 37), (50257, 1600, 2, 1024)])
def test_embedding_forward_backward_match_fp32_reference(dtype, V, d, B, T):
    """Fused token + position lookup and its deterministic scatter backward against an fp32 PyTorch
    reference on the same (rounded) inputs; skewed token frequencies so runs of equal tokens are long."""
    from pytorch_distributed_amd import ops

    g = torch.Generator(device="cpu").manual_seed(V + d)
    wte = (torch.randn(V, d, generator=g) * 0.02).to(dtype).to(DEV).requires_grad_(True)
    wpe = (torch.randn(1024, d, generator=g) * 0.01).to(dtype).to(DEV).requires_grad_(True)
    # a quarter of the positions use five very frequent tokens, the rest are uniform
    ids = torch.randint(0, V, (B, T), generator=g)
    hot = torch.rand(B, T, generator=g) < 0.25
    ids[hot] = torch.randint(0, 5, (int(hot.sum()),), generator=g)
    ids = ids.to(DEV)
    dout = torch.randn(B, T, d, generator=g).to(dtype).to(DEV)

    out = ops.token_position_embedding(ids, wte, wpe)
    ref_w, ref_p = wte.detach().float().requires_grad_(True), wpe.detach().float().requires_grad_(True)
    ref = torch.nn.functional.embedding(ids, ref_w) + ref_p[:T]
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-6
    assert (out.float() - ref).abs().max().item() <= tol * 0.1
    out.backward(dout)
    ref.backward(dout.float())
    assert rel_err(wte.grad, ref_w.grad) < (6e-3 if dtype == torch.bfloat16 else 1e-6)
    assert rel_err(wpe.grad, ref_p.grad) < (6e-3 if dtype == torch.bfloat16 else 1e-6)
    assert wpe.grad[T:].abs().max().item() == 0.0 if T < 1024 else True
    # untouched rows are exactly zero, and the result is bit-reproducible
    untouched = torch.ones(V, dtype=torch.bool, device=DEV)
    untouched[ids.reshape(-1)] = False
    assert wte.grad[untouched].abs().max().item() == 0.0
    first = wte.grad.clone()
    wte.grad = None
    wpe.grad = None
    ops.token_position_embedding(ids, wte, wpe).backward(dout)
    assert torch.equal(wte.grad, first)


def test_embedding_backward_accumulates_into_an_existing_gradient():
    """The tied LM-head weight already holds a gradient when the embedding backward runs: rows are
    added in place, everything else is left alone."""
    from pytorch_distributed_amd import ops

    V, d, B, T = 1000, 256, 2, 128
    g = torch.Generator(device="cpu").manual_seed(3)
    wte = torch.randn(V, d, generator=g).to(torch.bfloat16).to(DEV).requires_grad_(True)
    ids = torch.randint(0, V, (B, T), generator=g).to(DEV)
    dout = torch.randn(B, T, d, generator=g).to(torch.bfloat16).to(DEV)
    existing = torch.randn(V, d, generator=g).to(torch.bfloat16).to(DEV)
    wte.grad = existing.clone()
    ops.embedding(ids, wte).backward(dout)
    ref = existing.float().index_add(0, ids.reshape(-1), dout.reshape(-1, d).float())
    assert rel_err(wte.grad, ref) < 6e-3


# ------------------------------------------------------------------ fused-epilogue GEMM
def _gelu_new_f32(u):
    return 0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3)))


@pytest.mark.parametrize("M,N,K", [(512, 256, 64), (1000, 776, 96), (4096, 5120, 1280), (300, 3072, 768), (2048, 6400, 1600)])
def test_gemm_fused_epilogues_match_fp32_reference(M, N, K):
    """out = epilogue(x @ w.T) for the four epilogues, against fp32 PyTorch on the same bf16 inputs;
    shapes include partial tiles in both dimensions and every GPT-2 width."""
    ext = _ext.require()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV)
    ref = x.float() @ w.float().t()
    (plain,) = ext.gemm_fused(x, w, None, "none", None)
    assert rel_err(plain, ref) < 4e-3
    (biased,) = ext.gemm_fused(x, w, b, "bias", None)
    assert rel_err(biased, ref + b.float()) < 4e-3
    u, act = ext.gemm_fused(x, w, b, "bias_gelu", None)
    assert torch.equal(u, biased)
    assert rel_err(act, _gelu_new_f32(u.float())) < 4e-3  # the activation of the stored (rounded) pre-activation
    pre = (torch.randn(M, N, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    (dgel,) = ext.gemm_fused(x, w, None, "mul_gelu_grad", pre)
    pf = pre.float().requires_grad_(True)
    _gelu_new_f32(pf).sum().backward()
    assert rel_err(dgel, ref * pf.grad) < 5e-3
    # strided operands (a column slice of a wider matrix) and an aux with a row pitch
    wide = torch.randn(M, K + 64, generator=g).to(torch.bfloat16).to(DEV)
    xs = wide[:, 32:32 + K] if K % 8 == 0 else wide[:, :K]
    if xs.data_ptr() % 16 == 0:
        (strided,) = ext.gemm_fused(xs, w, None, "none", None)
        assert rel_err(strided, xs.float() @ w.float().t()) < 4e-3
