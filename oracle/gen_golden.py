"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference); the fixtures it writes
are data (inputs + the reference's outputs), committed under tests/golden/.
    python oracle/gen_golden.py [--only stage1]
Nothing under tests/, bench.py or the package imports this module.
"""
import argparse
import os
import sys

import numpy as np
import torch

REF = os.environ.get("PICOPOSE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)


def disk_mask(B, size=224, frac=0.4):
    yy, xx = torch.meshgrid(torch.arange(float(size)), torch.arange(float(size)), indexing="ij")
    c = (size - 1) / 2.0
    return (((yy - c) ** 2 + (xx - c) ** 2) < (frac * size) ** 2).float()[None].repeat(B, 1, 1)


def gen_stage1():
    _ref()
    from utils.matching import matching_features_similarity, matching_templates

    cases = {}

    def add(name, bank, query, mask, topk):
        score, idx = matching_templates(bank.clone(), query.clone(), None, mask.clone(), topk=topk)
        cases[name] = dict(bank=bank.numpy(), query=query.numpy(), mask=mask.numpy(),
                           topk=np.int64(topk), score=score.numpy(), index=idx.numpy())

    g = torch.Generator().manual_seed(1234)
    # random, Bernoulli mask (query patch 0 sometimes unmasked -> column decisions live)
    B, N, C = 2, 6, 64
    add("random_bernoulli", torch.randn(B, N, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g),
        (torch.rand(B, 224, 224, generator=g) < 0.7).float(), 3)
    # disk mask (patch 0 is background: exercises the idx != 0 logic with sim[0,:] == 0)
    B, N, C = 2, 5, 128
    add("random_disk", torch.randn(B, N, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g),
        disk_mask(B), 5)
    # fully masked query -> every sim_avg is 0, top-k order is torch's tie order
    B, N, C = 1, 4, 64
    add("all_masked", torch.randn(B, N, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g),
        torch.zeros(B, 224, 224), 2)
    # patch-0 winners: template patch 0 is a copy of many query patches' direction, and query
    # patch 0 of many template patches' direction, so idx == 0 decisions occur often
    B, N, C = 2, 4, 64
    bank = torch.randn(B, N, C, 256, generator=g)
    query = torch.randn(B, C, 256, generator=g)
    for t in range(0, 256, 5):
        query[:, :, t] = bank[:, 1, :, 0] + 0.3 * torch.randn(B, C, generator=g)
    for s in range(0, 256, 7):
        bank[:, 2, :, s] = query[:, :, 0] + 0.3 * torch.randn(B, C, generator=g)
    add("patch0_winners", bank.reshape(B, N, C, 16, 16), query.reshape(B, C, 16, 16),
        torch.ones(B, 224, 224), 4)
    # all-negative similarities in some rows (negative scores enter sim_avg), non-square mask size
    B, N, C = 1, 3, 64
    query = torch.randn(B, C, 16, 16, generator=g)
    bank = -query[:, None].repeat(1, N, 1, 1, 1) + 0.5 * torch.randn(B, N, C, 16, 16, generator=g)
    add("negative_scores_mask100", bank, query, (torch.rand(B, 100, 100, generator=g) < 0.8).float(), 3)
    # un-normalised, badly scaled features (the reference normalises; so must we)
    B, N, C = 1, 4, 64
    add("scaled_features", 37.0 * torch.randn(B, N, C, 16, 16, generator=g),
        0.01 * torch.randn(B, C, 16, 16, generator=g), disk_mask(B), 4)

    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "stage1_matching_templates.npz"), **flat)

    # stage-2 similarity volume (matching.py:6-26)
    sims = {}
    B, C = 2, 64
    src, tar = torch.randn(B, C, 16, 16, generator=g), torch.randn(B, C, 16, 16, generator=g)
    sm = (torch.rand(B, 224, 224, generator=g) < 0.6).float()
    out = matching_features_similarity(src.clone(), tar.clone(), sm.clone(), None)
    sims.update({"random/src": src.numpy(), "random/tar": tar.numpy(), "random/src_mask": sm.numpy(),
                 "random/out": out.numpy()})
    np.savez_compressed(os.path.join(OUT, "stage2_similarity.npz"), **sims)
    print("stage1/stage2-similarity fixtures written")


GENERATORS = {"stage1": gen_stage1}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    for name, fn in GENERATORS.items():
        if a.only is None or a.only == name:
            fn()
