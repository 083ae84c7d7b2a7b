"""GPU parity: fused softmax statistics (lse, log-probs of chosen ids, backward into logits) vs torch fp32 on CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,V,n_ids", [(6, 50304, 2), (5, 50257, 17), (3, 97, 1), (33, 1031, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stats_forward_backward(rows, V, n_ids, dtype):
    from branchgan import ops
    gen = torch.Generator().manual_seed(rows + V)
    x = (3.0 * torch.randn(rows, V, generator=gen)).to(dtype)
    ids = torch.randint(0, V, (rows, n_ids), generator=gen)
    ids[0, 0] = -1                                   # "no id" marker
    if n_ids > 1:
        ids[1, 1] = ids[1, 0]                        # duplicate id: both gradients land on the same logit
    g_lse, g_logp = torch.randn(rows, generator=gen), torch.randn(rows, n_ids, generator=gen)

    xr = x.float().clone().requires_grad_(True)
    lse_ref = torch.logsumexp(xr, -1)
    logp_ref = torch.gather(torch.log_softmax(xr, -1), 1, ids.clamp(min=0)) * (ids >= 0)
    (lse_ref * g_lse).sum().backward(retain_graph=True)
    (logp_ref * g_logp).sum().backward()

    xd = x.cuda().requires_grad_(True)
    lse, logp = ops.softmax_stats(xd, ids.cuda())
    ((lse * g_lse.cuda()).sum() + (logp * g_logp.cuda()).sum()).backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-2   # fp32: fast exp + summation order; bf16: output rounding of dlogits
    np.testing.assert_allclose(lse.detach().cpu().numpy(), lse_ref.detach().numpy(), atol=2e-5, rtol=1e-6)
    np.testing.assert_allclose(logp.detach().cpu().numpy(), logp_ref.detach().numpy(), atol=2e-5, rtol=1e-6)
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.numpy(), atol=tol, rtol=tol)
    assert xd.grad.dtype == dtype


def test_3d_input_and_unaligned_rows():
    from branchgan import ops
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 50257, generator=gen)      # odd V: every second row starts 4 bytes off a 16-byte boundary
    ids = torch.randint(0, 50257, (2, 3, 2), generator=gen)
    lse, logp = ops.softmax_stats(x.cuda(), ids.cuda())
    assert lse.shape == (2, 3) and logp.shape == (2, 3, 2)
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(x, -1).numpy(), atol=2e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), torch.gather(torch.log_softmax(x, -1), 2, ids).numpy(), atol=2e-5)
