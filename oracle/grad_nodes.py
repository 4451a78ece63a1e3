"""STUDY TOOL (test infrastructure, runs only where /root/reference exists): the gradient of the reference's training loss at INTERIOR
nodes of the DPT head / flow decoder, evaluated in float64 and in float32, for locating where a build's backward first departs from
float64 (profiles/r05/grad_f64.txt).  Same step as gen_golden.gen_train_grads_f64.  Writes, for every hooked node and call,
`sum/<node>/<call>` = per-channel sums over batch and pixels (C,), `max/<node>/<call>` = max |g|, and `map/<node>/<call>` = the whole map
(NCHW) for nodes of at most 32 x 32 pixels.   usage: python oracle/grad_nodes.py [dir = _dbg]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import gen_golden as G  # noqa: E402


def run(double):
    G._ref()
    sys.path.insert(0, os.path.join(G.REF, "model"))
    from oracle import ref_shims

    ref_shims.install()
    import picopose as ref_picopose

    from oracle.weights import AFFINE_CALIBRATION, HEAD_CALIBRATION, PROJ_BN_GAIN, apply_head_calibration, seeded_state_dict

    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    from netcfg import make_train_end_points, train_case, train_kwargs

    f32 = np.load(os.path.join(HERE, "..", "tests", "golden", "train_grads.npz"))
    B, seed, edit = train_case("train_forward")
    vit, wseed = "dinov2_vits14", 4
    net = ref_picopose.Net(G._cfg(vit)).train()
    cal = dict(HEAD_CALIBRATION[vit], affine=AFFINE_CALIBRATION, proj_bn=PROJ_BN_GAIN)
    net.load_state_dict(apply_head_calibration(seeded_state_dict(net.state_dict(), wseed), cal))
    ep = edit(make_train_end_points(B, seed, **train_kwargs("train_grads")))
    orig_aug, orig_float, orig_dtype = ref_picopose.aug_gtM_noise, torch.Tensor.float, torch.get_default_dtype()
    if double:
        net = net.double()
        ep = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in ep.items()}
        torch.Tensor.float = lambda self, *a, **k: self.double()
        torch.set_default_dtype(torch.float64)
    pm = torch.from_numpy(f32["pred_Ms"])
    ref_picopose.aug_gtM_noise = lambda end_points: pm.double() if double else pm.clone()
    out, calls = {}, {}

    def keep(name):
        def hook(g):
            g = g.detach().double()
            out[f"sum/{name}"] = g.sum(dim=(0, 2, 3)).numpy()
            out[f"max/{name}"] = np.float64(g.abs().max())
            if g.shape[-1] <= 32:
                out[f"map/{name}"] = g.numpy()
        return hook

    def watch(mod, node):
        def fwd(m, inp, o):
            c = calls.get(node, 0)
            calls[node] = c + 1
            if o.requires_grad:
                o.register_hook(keep(f"{node}.out/{c}"))
            if inp[0].requires_grad:
                inp[0].register_hook(keep(f"{node}.in/{c}"))
            if ".rcu" in node and node.count(".") == 1 or node.endswith(".bn1"):
                # values next to a ReLU kink: the unit's input (ReLU in front of conv1) and bn1's output (ReLU in front of conv2)
                v = (inp[0] if ".bn1" not in node else o).detach().double().permute(0, 2, 3, 1).reshape(-1)      # NHWC order
                idx = torch.nonzero(v.abs() < 1e-4).reshape(-1)
                out[f"near0idx/{node}/{c}"] = idx.numpy()
                if node.startswith("refinenet2.rcu2") and c == 1:
                    out[f"fwd/{node}/{c}"] = v.numpy()
                out[f"near0val/{node}/{c}"] = v[idx].numpy()
        mod.register_forward_hook(fwd)

    orr = net.offset_regressor
    s = orr.dpt_head.scratch
    for i in (2, 3, 4):
        f = getattr(s, f"refinenet{i}")
        watch(f, f"refinenet{i}")
        watch(f.out_conv, f"refinenet{i}.out_conv")
        watch(f.resConfUnit2, f"refinenet{i}.rcu2")
        for u, un in ((f.resConfUnit1, "rcu1"), (f.resConfUnit2, "rcu2")):
            if i == 4 and un == "rcu1":
                continue
            watch(u.conv1, f"refinenet{i}.{un}.conv1")
            watch(u.bn1, f"refinenet{i}.{un}.bn1")
            watch(u.conv2, f"refinenet{i}.{un}.conv2")
            watch(u.bn2, f"refinenet{i}.{un}.bn2")
        if i != 4:
            watch(f.resConfUnit1, f"refinenet{i}.rcu1")
    for l in range(len(orr.flow_decoder.proj)):
        watch(orr.flow_decoder.proj[l][0], f"proj{l}.conv")
        watch(orr.flow_decoder.proj[l][1], f"proj{l}.bn")
    try:
        np.random.seed(1000 + seed)
        torch.manual_seed(2000 + seed)
        res = net(ep)
        from utils.loss_utils import Loss

        total = Loss()(res)["loss"]
        total.backward()
    finally:
        ref_picopose.aug_gtM_noise, torch.Tensor.float = orig_aug, orig_float
        torch.set_default_dtype(orig_dtype)
    print("total", float(total), total.dtype, len(out), "entries")
    return out


MAPS = ["refinenet2.rcu2.conv1.in/1", "refinenet2.rcu2.bn1.in/1", "refinenet2.rcu2.conv2.in/1", "refinenet2.rcu2.bn2.in/1", "refinenet2.rcu2.out/1",
        "refinenet2.rcu2.in/1"]

if __name__ == "__main__":
    # -> <dir>/sums_f64.npz, sums_f32.npz (every node), maps_f64.npz, maps_f32.npz (whole gradient maps of MAPS), fwd.npz (forward values at
    # refinenet2.resConfUnit2's two ReLUs, real crops): what tools/grad_nodes_hip.py reads from _dbg/ on the GPU box (scratch, not committed)
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "_dbg")
    os.makedirs(d, exist_ok=True)
    r = {"f64": run(True), "f32": run(False)}
    fwd = {}
    for t, o in r.items():
        np.savez(os.path.join(d, f"sums_{t}.npz"), **{k: v for k, v in o.items() if not k.startswith(("map/", "fwd/"))})
        np.savez(os.path.join(d, f"maps_{t}.npz"), **{k: (o["map/" + k] if t == "f64" else o["map/" + k].astype(np.float32)) for k in MAPS})
        fwd.update({f"{t}/{k}": (v if t == "f64" else v.astype(np.float32)) for k, v in o.items() if k.startswith("fwd/")})
    np.savez(os.path.join(d, "fwd.npz"), **fwd)
