"""Native weight-gradient GEMM against torch.mm in fp32."""
import pytest
import torch

from pytorch_distributed_amd import _ext, ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,I,J", [(8192, 1280, 1280), (4096, 3840, 1280), (8192, 256, 5120), (64, 256, 256),
                                   (2048 + 64, 768, 2304),
                                   # partial edge tiles (GPT-2 XL d = 1600, odd small shapes)
                                   (4096, 1600, 1600), (2048, 4800, 1600), (1024, 1600, 6400), (512, 72, 264)])
@pytest.mark.parametrize("bias", [True, False])
def test_linear_wgrad_matches_fp32(M, I, J, bias):
    ext = _ext.require()
    torch.manual_seed(0)
    dy = torch.randn(M, I, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, J, device="cuda", dtype=torch.bfloat16)
    assert ext.wgrad_supported(M, I, J)
    dw, db = ext.linear_wgrad(dy, x, bias)
    ref = dy.float().t() @ x.float()
    err = ((dw.float() - ref).norm() / ref.norm()).item()
    assert err < 4e-3, err
    if bias:
        rb = dy.float().sum(0)
        assert ((db.float() - rb).norm() / rb.norm()).item() < 4e-3
    else:
        assert db is None


def test_linear_wgrad_asymmetric_integer_data():
    """Exact small-integer operands: a transposed or k-permuted operand map cannot pass."""
    ext = _ext.require()
    M, I, J = 128, 256, 256
    m = torch.arange(M, device="cuda")
    dy = ((m[:, None] * 3 + torch.arange(I, device="cuda")[None, :]) % 7 - 3).to(torch.bfloat16)
    x = ((m[:, None] * 5 + 2 * torch.arange(J, device="cuda")[None, :]) % 5 - 2).to(torch.bfloat16)
    dw, db = ext.linear_wgrad(dy, x, True)
    assert torch.equal(dw.float(), (dy.float().t() @ x.float()).bfloat16().float())
    assert torch.equal(db.float(), dy.float().sum(0).bfloat16().float())


def test_linear_op_grads():
    _ext.require()
    torch.manual_seed(1)
    x = torch.randn(4, 512, 768, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    w = (0.02 * torch.randn(2304, 768, device="cuda")).to(torch.bfloat16).requires_grad_(True)
    b = torch.zeros(2304, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(4, 512, 2304, device="cuda", dtype=torch.bfloat16)
    ops.linear(x, w, b).backward(dy)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.linear(xr, wr, br).backward(dy.float())
    for a, r in ((x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        assert ((a.float() - r).norm() / r.norm()).item() < 1e-2


@pytest.mark.parametrize("n,rows,I,J", [(2, 8192, 1280, 1280), (4, 8192, 1280, 5120), (3, 2048, 3840, 1280),
                                        (4, 256, 72, 264), (2, 32, 256, 256), (4, 1024, 1600, 1600)])
@pytest.mark.parametrize("bias", [True, False])
def test_linear_wgrad_segments_matches_fp32_over_the_concatenation(n, rows, I, J, bias):
    """Up to four (dy, x) pairs in separate buffers multiplied as ONE long product: equal to the fp32
    product over their row-wise concatenation, and to the sum of the per-pair kernel calls."""
    ext = _ext.require()
    torch.manual_seed(n)
    # separate allocations, deliberately not in address order
    dys = [torch.randn(rows, I, device="cuda", dtype=torch.bfloat16) for _ in range(n)][::-1]
    pad = torch.empty(12345, device="cuda")  # noqa: F841 - shifts the following allocations
    xs = [torch.randn(rows, J, device="cuda", dtype=torch.bfloat16) for _ in range(n)]
    assert ext.wgrad_segments_supported(rows, I, J, n) and ext.wgrad_max_segments() >= 4
    dw, db = ext.linear_wgrad_segments(dys, xs, bias)
    ref = torch.cat(dys).float().t() @ torch.cat(xs).float()
    assert ((dw.float() - ref).norm() / ref.norm()).item() < 4e-3
    if bias:
        rb = torch.cat(dys).float().sum(0)
        assert ((db.float() - rb).norm() / rb.norm()).item() < 4e-3
    else:
        assert db is None
    # fp32 target, accumulated onto existing contents
    out = torch.ones(I, J, device="cuda", dtype=torch.float32)
    bout = torch.ones(I, device="cuda", dtype=torch.float32) if bias else None
    ext.linear_wgrad_segments(dys, xs, bias, out, True, bout, bias)
    assert ((out - 1 - ref).norm() / ref.norm()).item() < 1e-4  # fp32 accumulation end to end
    if bias:
        assert ((bout - 1 - rb).norm() / rb.norm()).item() < 1e-4


def test_linear_wgrad_segments_exact_on_integer_data():
    """Small-integer operands that differ per pair: a stage read from the wrong pair cannot pass."""
    ext = _ext.require()
    rows, I, J, n = 128, 256, 264, 3
    m = torch.arange(rows, device="cuda")
    dys = [((m[:, None] * (3 + k) + torch.arange(I, device="cuda")[None, :]) % 7 - 3).to(torch.bfloat16) for k in range(n)]
    xs = [((m[:, None] * (5 + k) + 2 * torch.arange(J, device="cuda")[None, :]) % 5 - 2).to(torch.bfloat16) for k in range(n)]
    out = torch.zeros(I, J, device="cuda", dtype=torch.float32)
    bout = torch.zeros(I, device="cuda", dtype=torch.float32)
    ext.linear_wgrad_segments(dys, xs, True, out, False, bout, False)
    ref = sum(a.float().t() @ b.float() for a, b in zip(dys, xs))
    assert torch.equal(out, ref)
    assert torch.equal(bout, sum(a.float().sum(0) for a in dys))


def test_linear_wgrad_segments_rejects_what_the_kernel_cannot_address():
    ext = _ext.require()
    assert not ext.wgrad_segments_supported(96, 256, 256, 2)   # 3 stages per pair: not a power of two
    assert not ext.wgrad_segments_supported(64, 256, 256, 5)   # more pairs than the kernel takes
    dys = [torch.zeros(64, 256, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    xs = [torch.zeros(64, 256, device="cuda", dtype=torch.bfloat16), torch.zeros(64, 264, device="cuda", dtype=torch.bfloat16)]
    with pytest.raises(RuntimeError):
        ext.linear_wgrad_segments(dys, xs, False)
