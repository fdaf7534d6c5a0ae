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
