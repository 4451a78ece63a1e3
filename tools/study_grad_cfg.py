"""STUDY: do the engine's tile configurations change the BITS of the training gradients?  Runs the ViT-S training step of
tests/dist_worker_train_gpu.py (deterministic scatter adjoints) in this process under the current PP_GEMM_* environment and writes every
gradient to <out>.pt; `compare a.pt b.pt` prints the tensors that differ.  usage: study_grad_cfg.py run <out> | compare <a> <b>"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if sys.argv[1] == "run":
    import numpy as np
    from netcfg import make_train_end_points, small_cfg

    from picopose_amd import autograd
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss
    from picopose_amd.utils.seeding import calibrated_state_dict

    autograd.DETERMINISTIC = True
    net = Net(small_cfg())
    net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vits14"))
    net = net.cuda().train()
    ep = {k: v.cuda() for k, v in make_train_end_points(2, 100).items()}
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}

    def step():
        net.load_state_dict(sd0)                 # (training-mode BatchNorm moved the running buffers)
        net.zero_grad(set_to_none=True)
        np.random.seed(700)
        torch.manual_seed(900)
        loss = Loss()(net(ep))["loss"]
        loss.backward()
        torch.cuda.synchronize()
        return {"loss": loss.detach().cpu(), **{n: p.grad.cpu().clone() for n, p in net.named_parameters() if p.grad is not None}}

    first = step()                               # (alone or not: also the autotuner's pass)
    if len(sys.argv) > 3:                        # run <out> <flag-file> [repeats]: wait for the flag, then repeat the step side by side with the peers
        import time

        open(sys.argv[3] + f".ready{os.getpid()}", "w").close()
        while not os.path.exists(sys.argv[3]):
            time.sleep(0.01)
        bad = 0
        for i in range(int(sys.argv[4]) if len(sys.argv) > 4 else 5):
            g = step()
            diff = [k for k in first if not torch.equal(first[k], g[k])]
            bad += bool(diff)
            if diff:
                errs = sorted((float((first[k] - g[k]).abs().max()) / max(float(first[k].abs().max()), 1e-30), k) for k in diff)
                big = [k for k in first if k != "loss" and float(first[k].abs().max()) > 1e-3]
                eb = sorted((float((first[k] - g[k]).abs().max()) / float(first[k].abs().max()), k) for k in big)
                same = [k for k in first if k not in diff and k != "loss"]
                if os.environ.get("PP_STUDY_VERBOSE") == "1":
                    print(f"pid {os.getpid()} repeat {i}: EQUAL tensors: {[k.replace('offset_regressor.', 'OR.').replace('feature_extractor.dinov2.', 'vit.') for k in same][:45]}", flush=True)
                print(f"pid {os.getpid()} repeat {i}: loss equal: {torch.equal(first['loss'], g['loss'])}", flush=True)
                print(f"pid {os.getpid()} repeat {i}: {len(diff)} tensors differ from this process's first step; median {errs[len(errs) // 2][0]:.1e}; "
                      f"among the {len(big)} tensors with max|g| > 1e-3: worst {eb[-1][0]:.1e} {eb[-1][1]}, median {eb[len(eb) // 2][0]:.1e}", flush=True)
        print(f"pid {os.getpid()}: {bad} repeats with different bits", flush=True)
        torch.save(g, sys.argv[2] + ".last")
    torch.save(first, sys.argv[2])
    print("loss %.9f" % float(first["loss"]))
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    rows = []
    for k in a:
        if k == "loss":
            print("loss", float(a[k]), float(b[k]), "equal" if torch.equal(a[k], b[k]) else "DIFFERENT")
            continue
        if not torch.equal(a[k], b[k]):
            rows.append((float((a[k] - b[k]).abs().max()) / max(float(a[k].abs().max()), 1e-30), k))
    rows.sort(reverse=True)
    print(len(rows), "of", len(a) - 1, "gradient tensors differ; worst:", [(f"{e:.1e}", n) for e, n in rows[:6]])
