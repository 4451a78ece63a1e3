import os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_train_gpu import _load_grad_fixture, _cuda
from netcfg import small_cfg
from picopose_amd.picopose import Net
from picopose_amd.utils.loss_utils import Loss
z, ep, weights = _load_grad_fixture("tests/golden")
ep = _cuda(ep)
for lr in (1e-3, 3e-4, 1e-4, 3e-5):
    net = Net(small_cfg()); net.load_state_dict(weights(net.state_dict())); net = net.cuda().train()
    opt = None; rows = []
    for step in range(5):
        res = net(dict(ep)); tot = Loss()(res)
        rows.append([float(res[k]) for k in ("loss_info", "loss_2d_trans", "loss_scale", "loss_inplane")])
        tot["loss"].backward()
        tr = [p for p in net.parameters() if p.grad is not None]
        if opt is None: opt = torch.optim.SGD(tr, lr=lr)
        opt.step(); opt.zero_grad(set_to_none=True)
    print(lr, [[round(v, 5) for v in r] for r in rows], flush=True)
