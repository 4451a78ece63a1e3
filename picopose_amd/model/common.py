"""Parameter containers with the reference's state_dict names, and the pack-once cache.

The modules below hold exactly the tensors the reference's nn.Modules hold (same names, shapes and
buffers, so the authors' Lightning checkpoint loads unchanged — SURVEY.md §8b) but no torch compute:
`forward` runs on the HIP engine (picopose_amd/ops.py).  Weights are re-laid-out once per load
("packing": conv filters to (Cout, KH*KW*Cin), eval BatchNorm folded into the preceding conv, ...)."""
import torch
import torch.nn as nn


class Holder(nn.Module):
    """A bare named container (children / parameters are attached by the builders below)."""


def linear_p(cin, cout, bias=True):
    m = Holder()
    m.weight = nn.Parameter(torch.zeros(cout, cin))
    if bias:
        m.bias = nn.Parameter(torch.zeros(cout))
    return m


def conv_p(cin, cout, k, bias=True):
    m = Holder()
    m.weight = nn.Parameter(torch.zeros(cout, cin, k, k))
    if bias:
        m.bias = nn.Parameter(torch.zeros(cout))
    return m


def convT_p(cin, cout, k):
    m = Holder()
    m.weight = nn.Parameter(torch.zeros(cin, cout, k, k))
    m.bias = nn.Parameter(torch.zeros(cout))
    return m


def norm_p(c):
    m = Holder()
    m.weight = nn.Parameter(torch.ones(c))
    m.bias = nn.Parameter(torch.zeros(c))
    return m


def bn_p(c):
    m = norm_p(c)
    m.register_buffer("running_mean", torch.zeros(c))
    m.register_buffer("running_var", torch.ones(c))
    m.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
    return m


def seq(*mods, skip=()):
    """nn.Sequential-like numbering with holes (activation slots own no tensors): returns a ModuleDict-backed
    container whose children are named by their index in the reference's Sequential/ModuleList."""
    m = Holder()
    for i, sub in mods:
        m.add_module(str(i), sub)
    return m


def fold_bn(w, b, bn, eps=1e-5):
    """Conv (weight (Cout,...), bias or None) followed by eval BatchNorm2d -> equivalent conv."""
    s = bn.weight / torch.sqrt(bn.running_var + eps)
    w2 = w * s.reshape(-1, *([1] * (w.dim() - 1)))
    b0 = b if b is not None else torch.zeros_like(bn.running_mean)
    return w2, (b0 - bn.running_mean) * s + bn.bias


class Packed(nn.Module):
    """Mixin: `self.packed()` returns the cached device-side re-layout of the weights."""

    def __init__(self):
        super().__init__()
        self._pack_cache = None
        self._pack_train_cache = None   # training forward: convolutions whose BatchNorm is NOT folded (it runs on batch statistics)
        self._bn_stale = False          # a training step moved the running buffers the eval packing has folded

    def _apply(self, fn, *a, **k):
        self._pack_cache = self._pack_train_cache = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        # nn.Module.load_state_dict recurses through `_load_from_state_dict` of EVERY module of the tree — this hook
        # fires when a checkpoint is loaded through any ancestor (Net.load_state_dict, Lite.load_from_checkpoint),
        # whereas an override of load_state_dict only sees loads that start at this very module
        self._pack_cache = self._pack_train_cache = None
        return super()._load_from_state_dict(*a, **k)

    def _signatures(self):
        """(address, version) of every parameter / of every buffer the packings are derived from: an in-place update of the
        tensor ITSELF between two calls (`with torch.no_grad(): p.copy_(..)` / `p.mul_(..)`, a torch optimizer step) changes the
        version, a re-allocation the address.  NOT seen: writes through `p.data` (`p.data.copy_`, `p.data.mul_`: typical EMA /
        weight-surgery code) — `.data` is a view with its OWN version counter, so neither number moves.  Code that writes
        through `.data` must call `invalidate_packed()` (on the module, or `Net.invalidate_packed()` for all of them)
        afterwards; tests/test_abi.py pins both behaviours."""
        sig = lambda ts: tuple((t.data_ptr(), t._version) for t in ts)  # noqa: E731
        return sig(self.parameters()), sig(self.buffers())

    def _check_sources(self, buffers_too):
        psig, bsig = self._signatures()
        if psig != getattr(self, "_pack_psig", None):            # a weight moved: every derived copy is stale
            self._pack_cache = self._pack_train_cache = None
            self._pack_psig = psig
        if buffers_too and bsig != getattr(self, "_pack_bsig", None):   # a BatchNorm buffer moved: the eval packing folds them
            self._pack_cache = None
            self._pack_bsig = bsig

    def invalidate_packed(self):
        """Drop the derived copies of the weights explicitly (they are also dropped by .to() / load_state_dict and whenever a
        parameter's or buffer's version counter or address has changed since they were built — which a write through `.data`
        does not do: call this after one)."""
        from .. import ops

        self._pack_cache = self._pack_train_cache = None
        self._pack_psig = self._pack_bsig = None
        ops.drop_split_cache()          # the pre-split operand copies are keyed the same way

    def packed(self, for_training=False):
        """The eval packing.  for_training: the caller reads only entries that do not fold a BatchNorm, so a packing whose
        folds a training step has made stale is still good (and is not rebuilt every step)."""
        self._check_sources(buffers_too=not for_training)
        if self._pack_cache is None or (self._bn_stale and not for_training):
            with torch.no_grad():
                self._pack_cache = self._pack()
            self._bn_stale = False
            self._pack_psig, self._pack_bsig = self._signatures()
        return self._pack_cache

    def packed_train(self):
        """Weights of the layers that differ in training mode (the module's `_pack_train`).  A training step moves the
        BatchNorm running buffers: the eval packing, which folds them, is marked stale and re-folded by the next eval call."""
        self._check_sources(buffers_too=False)
        if self._pack_train_cache is None:
            with torch.no_grad():
                self._pack_train_cache = self._pack_train()
        return self._pack_train_cache

    def bn_moved(self):
        self._bn_stale = True
