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


@gpu
@pytest.mark.parametrize("mask", ["bernoulli", "disk"])
def test_fast_mode_resolves_near_ties_like_exact_mode(mask):
    """Plant near-ties with patch 0 in rows and columns: the fp16 pass cannot decide them,
    the fix-up pass must, and then fast == exact far below one decision's weight (~4e-4)."""
    from picopose_amd.utils import matching as hm

    B, N, C = 2, 12, 384
    bank, query, m = _inputs(B, N, C, 77, mask)
    g = torch.Generator().manual_seed(78)
    bank = bank.reshape(B, N, C, 256).clone()
    query = query.reshape(B, C, 256).clone()
    # template patches 5, 40, 200 become near-copies of template patch 0 (row decisions) ...
    for s in (5, 40, 200):
        bank[:, :, :, s] = bank[:, :, :, 0] * (1.0 + 2e-5 * torch.randn(B, N, C, generator=g))
    # ... and query patches 7, 130 near-copies of query patch 0 (column decisions)
    for t in (7, 130):
        query[:, :, t] = query[:, :, 0] * (1.0 + 2e-5 * torch.randn(B, C, generator=g))
    m[:, 0, 0] = 1.0  # keep query patch 0 unmasked so its row takes part in the column arg-max
    m[:, 14 * 0, 14 * 7] = 1.0
    m[:, 14 * 8, 14 * 2] = 1.0
    bank = bank.reshape(B, N, C, 16, 16).cuda()
    query = query.reshape(B, C, 16, 16).cuda()
    m = m.cuda()
    exact = hm.template_scores(bank, query, m, mode="exact")
    fast, stats = hm.template_scores(bank, query, m, mode="fast", return_stats=True)
    stats = stats.tolist()
    assert stats[0] + stats[1] > 0, stats  # row fix-ups ran
    assert stats[2] > 0, stats  # column fix-ups ran
    assert (exact - fast).abs().max().item() <= 1e-5
    # and without the fix-ups (eps ~ 0) the fp16 pass does get decisions wrong
    raw = hm.template_scores(bank, query, m, mode="fast", eps=1e-12)
    assert (exact - raw).abs().max().item() > 1e-4


@gpu
def test_matching_templates_api_matches_oracle_topk():
    from picopose_amd.utils import matching as hm

    bank, query, m = _inputs(4, 42, 384, 11, "disk")
    rs, ri = om.matching_templates(bank, query, None, m, topk=5)
    for mode in ("exact", "fast"):
        gs, gi = hm.matching_templates(bank.cuda(), query.cuda(), None, m.cuda(), topk=5, mode=mode)
        assert gi.dtype == torch.int64 and gs.dtype == torch.float32
        assert torch.equal(gi.cpu(), ri), mode  # gaps between adjacent scores here are >> 1e-5
        assert (gs.cpu() - rs).abs().max().item() <= 1e-5


@gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_golden_fixtures_from_the_reference(mode, golden_dir):
    """The reference's own outputs (tests/golden, generated from /root/reference)."""
    import os

    import numpy as np

    from picopose_amd.utils import matching as hm

    z = np.load(os.path.join(golden_dir, "stage1_matching_templates.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    for name in names:
        c = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
        bank, query, mask = (torch.from_numpy(c[k]) for k in ("bank", "query", "mask"))
        k = int(c["topk"])
        gs, gi = hm.matching_templates(bank.cuda(), query.cuda(), None, mask.cuda(), topk=k, mode=mode)
        gs, gi = gs.cpu().numpy(), gi.cpu().numpy()
        tol = 2e-6 if mode == "exact" else 2e-5
        assert np.abs(gs - c["score"]).max() <= tol, (name, np.abs(gs - c["score"]).max())
        if name == "all_masked":
            assert np.count_nonzero(gs) == 0
            continue  # all scores tie at 0: torch's tie order is unspecified, ours is lowest-id-first
        assert np.array_equal(gi, c["index"]), name


@gpu
def test_full_size_b32_n162_c768_properties():
    """BASELINE configs[2] stage-1 shape (4.1 GB bank, 10368 work items over 512 persistent workgroups):
    oracle on a sample of crops, fast == exact decisions, and two size-independent properties —
    template permutation equivariance and invariance to positive per-column scaling of the bank
    (matching.py:43 normalises every template patch)."""
    from picopose_amd.utils import matching as hm

    B, N, C = 32, 162, 768
    g = torch.Generator(device="cuda").manual_seed(5)
    bank = torch.randn(B, N, C, 16, 16, device="cuda", generator=g)
    query = torch.randn(B, C, 16, 16, device="cuda", generator=g)
    yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
    disk = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()
    m = disk[None].repeat(B, 1, 1).cuda()

    fast = hm.template_scores(bank, query, m, mode="fast")
    exact = hm.template_scores(bank, query, m, mode="exact")
    # exact mode vs the CPU oracle on three crops (first, middle, last: every region of the item walk)
    for b in (0, 17, 31):
        ref = om.template_scores(bank[b:b + 1].cpu(), query[b:b + 1].cpu(), m[b:b + 1].cpu())
        margin = om.decision_margins(bank[b:b + 1].cpu(), query[b:b + 1].cpu(), m[b:b + 1].cpu())
        safe = margin > 1e-5
        assert safe.float().mean() > 0.9
        assert (exact[b:b + 1].cpu() - ref).abs()[safe].max().item() <= 2e-6
        assert (fast[b:b + 1].cpu() - ref).abs()[safe].max().item() <= 1e-5
    # fast mode takes the same discrete decisions as exact mode: scores differ by rounding only
    assert (fast - exact).abs().max().item() <= 1e-5
    # (on random features the 162 scores of a crop lie within ~1e-3 of each other, so the top-5 ORDER may differ
    # between the modes where two scores are within the rounding error; the top-5 score values may not)
    assert (hm.topk_templates(fast, 5)[0] - hm.topk_templates(exact, 5)[0]).abs().max().item() <= 1e-5

    # permutation of the template axis permutes the scores (bit-exact: same per-template arithmetic)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(6)).cuda()
    fast_p = hm.template_scores(bank[:, perm].contiguous(), query, m, mode="fast")
    assert torch.equal(fast_p, fast[:, perm])
    del fast_p
    # positive per-patch scaling by powers of two leaves every normalised column unchanged
    scale = torch.tensor([0.5, 1.0, 2.0, 4.0], device="cuda")[torch.randint(0, 4, (B, N, 1, 16, 16), device="cuda", generator=g)]
    bank.mul_(scale)
    exact_s = hm.template_scores(bank, query, m, mode="exact")
    assert (exact_s - exact).abs().max().item() <= 2e-6


@gpu
def test_max_size_b64_n512_c1024_indexing():
    """BASELINE configs[4] stage-1 shape on one GPU (34 GB bank, 65536 work items): byte offsets beyond 32 bits.
    The score of a (crop, template) pair depends on that pair only, so the oracle runs on a sample of pairs
    taken from the corners of the bank."""
    from picopose_amd.utils import matching as hm

    B, N, C = 64, 512, 1024
    free, _ = torch.cuda.mem_get_info()
    if free < 48 << 30:
        pytest.skip("needs 48 GB of free HBM")
    g = torch.Generator(device="cuda").manual_seed(9)
    bank = torch.empty(B, N, C, 16, 16, device="cuda")
    for b in range(B):  # fill crop by crop: a one-shot randn of 34 GB would need a second 34 GB of temporaries
        bank[b].normal_(generator=g)
    query = torch.randn(B, C, 16, 16, device="cuda", generator=g)
    m = (torch.rand(B, 224, 224, device="cuda", generator=g) < 0.7).float()
    fast = hm.template_scores(bank, query, m, mode="fast")
    assert torch.isfinite(fast).all()
    for b in (0, 33, 63):
        for n0 in (0, 255, 508):
            sl = slice(n0, n0 + 4)
            args = (bank[b:b + 1, sl].cpu(), query[b:b + 1].cpu(), m[b:b + 1].cpu())
            ref = om.template_scores(*args)
            safe = om.decision_margins(*args) > 1e-5
            err = (fast[b:b + 1, sl].cpu() - ref).abs()
            assert err[safe].max().item() <= 1e-5 if safe.any() else True
            assert err.max().item() <= 1.0 / 256 + 1e-5
    score, index = hm.topk_templates(fast, 5)
    ts, ti = torch.topk(fast, 5, dim=1)
    assert torch.equal(score, ts)


@gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_four_wave_workgroup_shape(mode, monkeypatch):
    """PP_S1_WAVES=4: the half-template work items (two 256-thread workgroups per CU) give the same scores as the
    default 8-wave workgroups — bit for bit, the per-wave arithmetic is identical — and match the oracle."""
    from picopose_amd.utils import matching as hm

    for B, N, C in ((9, 40, 768), (3, 7, 1024)):    # 720 half-items > 512 workgroups: the item walk rolls over
        bank, query, m = _inputs(B, N, C, 31)
        ref8 = hm.template_scores(bank.cuda(), query.cuda(), m.cuda(), mode=mode)
        monkeypatch.setenv("PP_S1_WAVES", "4")
        got4 = hm.template_scores(bank.cuda(), query.cuda(), m.cuda(), mode=mode)
        monkeypatch.delenv("PP_S1_WAVES")
        assert torch.equal(got4, ref8)
    _check(2, 6, 64, seed=5, mode=mode)


@gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("B,N,C", [(1, 4, 384), (2, 6, 64), (8, 42, 384), (5, 9, 768), (3, 7, 1024)])
def test_half_precision_bank_equals_fp32_path_on_the_rounded_values(B, N, C, mode):
    """BASELINE configs[4] stores the template bank as fp16 (2 B/elem in HBM).  The fp16-bank path must return what the
    reference arithmetic gives on those stored values: (a) the CPU oracle on bank.half().float() within the same
    tolerances as the fp32 path, (b) in exact mode BIT-equal to the HIP fp32-bank path fed with the widened values
    (same kernel arithmetic, only the load differs), (c) top-k ids equal."""
    from picopose_amd.utils import matching as hm

    bank, query, m = _inputs(B, N, C, seed=7 * B + N)
    bank16 = bank.half()
    wide = bank16.float()
    ref = om.template_scores(wide, query, m)
    margin = om.decision_margins(wide, query, m)
    got = hm.template_scores(bank16.cuda(), query.cuda(), m.cuda(), mode=mode)
    same = hm.template_scores(wide.cuda(), query.cuda(), m.cuda(), mode=mode)
    if mode == "exact":
        assert torch.equal(got, same)
    safe = margin > 1e-5
    err = (got.cpu() - ref).abs()
    assert err[safe].max().item() <= (2e-6 if mode == "exact" else 1e-5)
    assert err.max().item() <= 1.0 / 256 + 1e-5
    k = min(5, N)
    gs, gi = hm.matching_templates(bank16.cuda(), query.cuda(), None, m.cuda(), topk=k, mode=mode)
    ss, si = hm.matching_templates(wide.cuda(), query.cuda(), None, m.cuda(), topk=k, mode=mode)
    assert torch.equal(gi, si) or torch.equal(gs, ss)


@gpu
def test_half_precision_bank_full_size_and_golden(golden_dir):
    """The configs[2] stage-1 shape with an fp16 bank (2.04 GB instead of 4.08 GB): fast == exact decisions, and the
    reference-generated golden cases with their banks rounded to fp16 against the oracle on the rounded values."""
    import os

    import numpy as np

    from picopose_amd.utils import matching as hm

    g = torch.Generator(device="cuda").manual_seed(3)
    bank = torch.randn(32, 162, 768, 16, 16, device="cuda", generator=g).half()
    query = torch.randn(32, 768, 16, 16, device="cuda", generator=g)
    yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
    m = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()[None].repeat(32, 1, 1).cuda()
    fast = hm.template_scores(bank, query, m, mode="fast")
    exact = hm.template_scores(bank, query, m, mode="exact")
    assert (fast - exact).abs().max().item() <= 1e-5
    assert torch.equal(torch.topk(fast, 5, dim=1).indices, torch.topk(exact, 5, dim=1).indices) or (fast - exact).abs().max().item() < 2e-6
    z = np.load(os.path.join(golden_dir, "stage1_matching_templates.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        bank = torch.from_numpy(z[f"{name}/bank"]).half()
        if not torch.isfinite(bank).all():
            continue
        query, mask = torch.from_numpy(z[f"{name}/query"]), torch.from_numpy(z[f"{name}/mask"])
        ref = om.template_scores(bank.float(), query, mask)
        got = hm.template_scores(bank.cuda(), query.cuda(), mask.cuda(), mode="fast").cpu()
        margin = om.decision_margins(bank.float(), query, mask)
        safe = margin > 1e-5
        assert (got - ref).abs()[safe].max().item() <= 1e-5 if safe.any() else True


@gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("B,N,C,half", [(8, 42, 384, False), (1, 4, 384, False), (5, 9, 768, False), (32, 162, 768, False), (3, 70, 1024, True)])
def test_one_pass_match_equals_scores_then_topk(B, N, C, half, mode):
    """matching_templates runs as ONE ABI call (pp_stage1_match_ex; with fewer than two items per CU on the 4-wave workgroup
    shape).  The ids and scores must equal template_scores followed by topk_templates bit for bit, on every repeat, with the
    workspace reused between calls."""
    from picopose_amd.utils import matching as hm

    bank, query, m = _inputs(B, N, C, 77 + B)
    bank, query, m = (bank.cuda().half() if half else bank.cuda()), query.cuda(), m.cuda()
    k = min(5, N)
    sim = hm.template_scores(bank, query, m, mode=mode)
    rs, ri = hm.topk_templates(sim, k)
    for rep in range(8):
        s, i = hm.matching_templates(bank, query, None, m, topk=k, mode=mode)
        assert torch.equal(i, ri) and torch.equal(s, rs), (rep, i, ri)
    torch.cuda.synchronize()


@gpu
def test_one_pass_match_with_the_optional_launch_fusions(monkeypatch):
    """PP_S1_FUSE_TOPK=1 PP_S1_QPREP=1 PP_S1_TOPK_SMALL=1 (all off by default: measured slower, profiles/r04/stage1_small.txt): the
    arrival-counter top-k, the one-launch query pre-pack and the one-wave top-k give the same ids and scores, repeat after repeat."""
    import subprocess
    import sys

    code = ("import torch; from picopose_amd.utils import matching as hm; g = torch.Generator().manual_seed(5); "
            "bank = torch.randn(8, 42, 384, 16, 16, generator=g).cuda(); q = torch.randn(8, 384, 16, 16, generator=g).cuda(); "
            "m = (torch.rand(8, 224, 224, generator=g) < 0.7).float().cuda(); sim = hm.template_scores(bank, q, m); rs, ri = hm.topk_templates(sim, 5); "
            "ok = all(torch.equal(hm.matching_templates(bank, q, None, m, topk=5)[1], ri) for _ in range(20)); print('FUSED OK' if ok else 'FUSED DIFFERS')")
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, PP_S1_FUSE_TOPK="1", PP_S1_QPREP="1", PP_S1_TOPK_SMALL="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and "FUSED OK" in r.stdout.decode(), r.stdout.decode()[-1500:]


@gpu
def test_matching_graph_replay_equals_the_eager_call():
    """picopose_amd.utils.matching.MatchingGraph: the five launches of a call captured in ONE HIP graph; a replay returns the eager
    call's scores and ids bit for bit, also after the query and the mask were refreshed IN PLACE (BASELINE configs[1] shape)."""
    import torch

    from picopose_amd.utils import matching as hm

    B, N, C = 8, 42, 384
    g = torch.Generator(device="cuda").manual_seed(5)
    bank = torch.randn(B, N, C, 16, 16, device="cuda", generator=g)
    query = torch.randn(B, C, 16, 16, device="cuda", generator=g)
    mask = (torch.rand(B, 224, 224, device="cuda", generator=g) < 0.7).float()
    mg = hm.MatchingGraph(bank, query, mask, topk=5)
    s, i = mg()
    ws, wi = hm.matching_templates(bank, query, None, mask, topk=5)
    assert torch.equal(i, wi) and torch.equal(s, ws)
    query.copy_(torch.randn(B, C, 16, 16, device="cuda", generator=g))
    mask.copy_((torch.rand(B, 224, 224, device="cuda", generator=g) < 0.5).float())
    s, i = mg()
    ws, wi = hm.matching_templates(bank, query, None, mask, topk=5)
    assert torch.equal(i, wi) and torch.equal(s, ws)
