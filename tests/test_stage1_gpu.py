"""GPU parity: HIP stage-1 matching (through the C ABI) vs the CPU oracle."""
import pytest
import torch

from oracle import matching as om

gpu = pytest.mark.gpu


def _inputs(B, N, C, seed, mask="bernoulli"):
    g = torch.Generator().manual_seed(seed)
    bank = torch.randn(B, N, C, 16, 16, generator=g)
    query = torch.randn(B, C, 16, 16, generator=g)
    if mask == "bernoulli":
        m = (torch.rand(B, 224, 224, generator=g) < 0.7).float()
    elif mask == "disk":
        yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
        m = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()[None].repeat(B, 1, 1)
    elif mask == "ones":
        m = torch.ones(B, 224, 224)
    else:
        m = torch.zeros(B, 224, 224)
    return bank, query, m


def _check(B, N, C, seed, mode, mask="bernoulli", k=None, atol=None):
    from picopose_amd.utils import matching as hm

    # stated tolerance on sim_avg: exact mode differs from the oracle only by fp32
    # summation order; fast mode rounds the operands to fp16 (error ~ 2^-11/sqrt(C) per score)
    atol = atol or (2e-6 if mode == "exact" else 1e-5)
    bank, query, m = _inputs(B, N, C, seed, mask)
    ref = om.template_scores(bank, query, m)
    margin = om.decision_margins(bank, query, m)
    got = hm.template_scores(bank.cuda(), query.cuda(), m.cuda(), mode=mode).cpu()
    safe = margin > 1e-5  # decisions farther than fp32 reassociation error from a tie
    assert safe.float().mean() > 0.9
    err = (got - ref).abs()
    assert err[safe].max().item() <= atol, (mode, err[safe].max().item())
    # an unsafe template may legitimately flip one decision: bounded by one row's weight
    assert err.max().item() <= 1.0 / 256 + atol
    k = k or min(5, N)
    rs, ri = torch.topk(ref, k, dim=1)
    gs, gi = hm.topk_templates(got.cuda(), k)
    ts, ti = torch.topk(got, k, dim=1)
    assert torch.equal(gi.cpu(), ti) or torch.equal(gs.cpu(), ts)  # HIP top-k == torch top-k on the same scores
    return ref, got


@gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("B,N,C", [(1, 4, 384), (2, 6, 64), (8, 42, 384), (9, 5, 768), (3, 7, 1024)])
def test_scores_match_oracle(B, N, C, mode):
    _check(B, N, C, seed=B * 100 + N, mode=mode)


@gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("mask", ["disk", "ones", "zeros"])
def test_mask_variants(mode, mask):
    ref, got = _check(2, 6, 384, seed=5, mode=mode, mask=mask)
    if mask == "zeros":
        assert torch.count_nonzero(got) == 0
