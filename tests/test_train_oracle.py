"""CPU: the training-forward oracle (oracle/train.py) against the REFERENCE's own `Net.forward_train` + `Loss`
(tests/golden/train_forward.npz, oracle/gen_golden.py gen_train_forward): key-point sampler bit-exact, every loss, the
noise draws of aug_M_noise, BatchNorm running buffers after the step."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import HEADS, TAKE, make_train_end_points, small_cfg, train_case  # noqa: E402

from oracle import train as ot  # noqa: E402
from oracle.weights import apply_head_calibration, seeded_state_dict  # noqa: E402

CASES = ["train_forward", "train_forward_edge"]   # the second: a pair without any correspondence (tests/netcfg.train_case)
LOSS_KEYS = ["loss_info", "loss_2d_trans", "loss_scale", "loss_inplane"] + [f"loss_{k}{i}" for i in range(3) for k in ("flow", "certainty")]


def load_train_fixture(golden_dir, name="train_forward"):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    B, seed, wseed, np_seed, torch_seed = (int(v) for v in z["meta"])
    assert (B, seed) == train_case(name)[:2]
    cal = {"flow": [tuple(r) for r in z["cal_flow"]], "cert": [tuple(r) for r in z["cal_cert"]], "proj_bn": float(z["cal_proj_bn"]),
           "affine": {h: (float(z[f"cal_affine_{h}"][0]), tuple(z[f"cal_affine_{h}"][1:])) for h in ("translation", "scale", "inplane")}}
    ep = train_case(name)[2](make_train_end_points(B, seed, poses=(torch.from_numpy(z["real_pose"]), torch.from_numpy(z["tem_pose"]))))
    weights = lambda template: apply_head_calibration(seeded_state_dict(template, wseed), cal)  # noqa: E731
    return z, ep, weights, (np_seed, torch_seed)


def patch_coords(px):
    """The fixture stores key-points as integer pixels; the functions return pixels / 3.5 (or -1)."""
    p = torch.from_numpy(px.astype(np.float32))
    return torch.where(p == -1, p, p / 3.5)


@pytest.mark.parametrize("name", CASES)
def test_keypoint_sampler_is_bit_exact(golden_dir, name):
    z, ep, _, _ = load_train_fixture(golden_dir, name)
    kp = ot.keypoint_data({k: v.clone() for k, v in ep.items()})
    for k in ("src_pts", "tar_pts"):
        assert torch.equal(kp[k], patch_coords(z[f"kp_{k}_px"])), k
    assert int((kp["src_pts"][..., 0] != -1).sum()) > 1000        # the fixture exercises the sampler
    if name == "train_forward_edge":
        assert bool((kp["src_pts"][1] == -1).all())                # the pair without a real mask has no correspondence


@pytest.mark.parametrize("name", CASES)
def test_noise_draws_reproduce_the_reference_affines(golden_dir, name):
    z, ep, _, (np_seed, torch_seed) = load_train_fixture(golden_dir, name)
    np.random.seed(np_seed)
    torch.manual_seed(torch_seed)
    M = ot.noisy_M(ot.relative_M(ep), *ot.draw_noise(ep["real_rgb"].shape[0]))
    assert np.abs(M.numpy() - z["pred_Ms"]).max() < 2e-4 * np.abs(z["pred_Ms"]).max()


@pytest.mark.parametrize("name", CASES)
def test_training_forward_losses_and_batchnorm_buffers(golden_dir, name):
    from picopose_amd.picopose import Net

    torch.set_num_threads(8)
    z, ep, weights, _ = load_train_fixture(golden_dir, name)
    sd = {k: v.clone() for k, v in weights(Net(small_cfg()).state_dict()).items()}
    with torch.no_grad():
        losses, aux = ot.net_forward_train(sd, {k: v.clone() for k, v in ep.items()}, HEADS, TAKE, torch.from_numpy(z["pred_Ms"]))
    for k in LOSS_KEYS:
        assert abs(float(losses[k]) - float(z[k])) <= 2e-4 * max(1.0, abs(float(z[k]))), (k, float(losses[k]), float(z[k]))
    assert abs(float(ot.total_loss(losses)) - float(z["total_loss"])) <= 2e-4 * float(z["total_loss"])
    for key in z.files:
        if key.startswith("bn/"):
            got, ref = sd[key[3:]].numpy(), z[key]
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), key
            if key.endswith("num_batches_tracked"):
                assert int(got) == int(ref) and int(ref) in (1, 2)     # the DPT head sees two batches per step
