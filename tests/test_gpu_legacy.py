"""GPU parity for the two call-surface pieces the reference keeps beside the tree path: `head_mask` in the branch attention
functions (NeoXNonResidualAttention.py:221-222, NonResidualAttention.py:161-162) and the legacy single-step `discForward`
(Discriminator.py:28-79,112-127), both against the oracle."""
import math

import numpy as np
import pytest
import torch

from oracle import attention as A
from oracle import models as OM

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family", ["gpt_neox", "gpt2"])
def test_head_mask_in_the_branch_attention_function(family):
    """One factor per head on the softmax weights (what HuggingFace's get_head_mask hands a layer): output = (w * mask) @ V.
    The oracle forms the weights and masks them; the fused path scales each head's output and the weights it hands out."""
    from branchgan import configs, models
    H, Dh, K, S, B = 4, 32, 24, 40, 2
    cfg = configs.LMConfig(family, 97, H * Dh, 1, H, 64)
    att = (models.NeoXNonResidualAttention(cfg) if family == "gpt_neox" else models.GPT2NonResidualAttention(cfg, layer_idx=0)).cuda().eval()
    gen = torch.Generator().manual_seed(4)
    q, k, v = [torch.randn(B, H, K, Dh, generator=gen) for _ in range(3)]
    kctx, vctx = torch.randn(B, H, S, Dh, generator=gen), torch.randn(B, H, S, Dh, generator=gen)
    head_mask = torch.tensor([1.0, 0.0, 0.5, 2.0]).view(1, H, 1, 1)
    fn = att._nonResidualTextAttention if family == "gpt_neox" else att.nonResidualTextAttention
    out, w = fn(q.cuda(), k.cuda(), v.cuda(), kctx.cuda(), vctx.cuda(), head_mask=head_mask.cuda(), output_attentions=True)
    fam = A.NEOX if family == "gpt_neox" else A.GPT2
    _, w_ref = A.branch_attention(fam, q, k, v, kctx, vctx)          # default mask
    w_ref = w_ref * head_mask
    o_ref = w_ref[..., :-1] @ vctx + w_ref[..., -1:] * v
    np.testing.assert_allclose(out.cpu().numpy(), o_ref.numpy(), atol=2e-5)
    np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), atol=2e-5)
    out1, none = fn(q.cuda(), k.cuda(), v.cuda(), kctx.cuda(), vctx.cuda(), head_mask=head_mask.view(H).cuda())
    assert none is None and torch.equal(out1, out)
    with pytest.raises(NotImplementedError):
        fn(q.cuda(), k.cuda(), v.cuda(), kctx.cuda(), vctx.cuda(), head_mask=torch.ones(1, H, K, 1).cuda())


@pytest.mark.parametrize("family", ["gpt_neox", "gpt2"])
def test_legacy_disc_forward_matches_the_oracle(family):
    from branchgan import configs, gan
    torch.manual_seed(2)
    mk = lambda hidden, heads: configs.LMConfig(family, 131, hidden, 2, heads, 64)
    setup = gan.GannSetup(mk(64, 2), mk(64, 2))
    with torch.no_grad():
        for p in setup.parameters():
            p.mul_(3.0)
    sd = {k_: v_.detach().clone() for k_, v_ in setup.state_dict().items()}
    disc = setup.discriminator.cuda()
    B, n, S = 2, 3, 20
    text = torch.randint(0, 131, (B, S))
    alts = torch.randint(0, 131, (B, n, S))
    txt, alt, heads = disc({'textualIDs': text.cuda(), 'alternationIDs': alts.cuda()})            # doDiscForward is the default
    ocfg = OM.LMConfig("neox" if family == "gpt_neox" else "gpt2", 2, 2, 64, 131, 64)
    t_ref, a_ref, h_ref = OM.discriminator_forward_legacy(sd, ocfg, text, alts)
    assert alt.shape == (B, n, S)
    np.testing.assert_allclose(txt.detach().cpu().numpy(), t_ref.numpy(), atol=2e-5)
    np.testing.assert_allclose(alt.detach().cpu().numpy(), a_ref.numpy(), atol=2e-5)
    for key in ("txtInputPreds", "txtAlterationsPreds", "altInputPreds", "altAlterationsPreds"):
        np.testing.assert_allclose(heads[key].detach().cpu().numpy().reshape(h_ref[key].shape), h_ref[key].numpy(), atol=2e-5, err_msg=key)
    # gradients flow to the discriminator through both passes
    (alt.mean() + txt.mean()).backward()
    assert all(torch.isfinite(p.grad).all() for p in disc.parameters() if p.grad is not None)
    assert disc.nonResidualModel.valueHead.weight.grad.abs().sum() > 0


def test_neox_sequential_residual_block_on_the_fused_path():
    """use_parallel_residual=False (no Pythia config uses it): the attention projection's bias gradient comes from the LayerNorm
    backward behind the residual add, and only from there - round 3 found it counted twice.  Every gradient of the fused path
    against the module-by-module path."""
    import dataclasses
    from branchgan import configs, fused, models
    torch.manual_seed(0)
    cfg = dataclasses.replace(configs.LMConfig("gpt_neox", 101, 64, 2, 2, 48), use_parallel_residual=False)
    lm = models.make_lm(cfg).cuda()
    with torch.no_grad():
        for p in lm.parameters():
            p.mul_(3.0)
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    ids = torch.randint(0, 101, (2, 24)).cuda()
    grads = {}
    try:
        for use in (True, False):
            fused.ENABLED = use
            lm.zero_grad()
            lm(input_ids=ids, returnLmLogits=True).logits.float().square().mean().backward()
            grads[use] = {n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None}
    finally:
        fused.ENABLED = True
    assert grads[True].keys() == grads[False].keys()
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-7, n
