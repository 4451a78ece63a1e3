"""Host side of stage-1/stage-2 matching — mirrors reference utils/matching.py.

`matching_templates` keeps the reference signature and return values
(utils/matching.py:29-69) but runs as ONE fused HIP pass over the bank
(picopose_amd/csrc/pp_stage1.hip) through the C ABI; nothing here computes on
the CPU and there is no torch fallback.
"""
import ctypes

import torch

from .. import _lib

# arithmetic of the 256x256xC contraction: "fast" (fp16 MFMA + exact fp32
# re-evaluation of near-tie decisions) or "exact" (fp32 MFMA, an fma chain).
DEFAULT_MODE = "fast"
_MODES = {"exact": _lib.PP_MATCH_EXACT, "fast": _lib.PP_MATCH_FAST}

_ws_cache = {}


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(B, N, C, device):
    need = ctypes.c_size_t()
    _lib.check(_lib.lib().pp_stage1_workspace_bytes(B, N, C, ctypes.byref(need)), "pp_stage1_workspace_bytes")
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need.value:
        ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws, need.value


def _check_inputs(src_feats, tar_feat, tar_mask):
    if not (src_feats.is_cuda and tar_feat.is_cuda and tar_mask.is_cuda):
        raise _lib.PicoPoseHipError("picopose_amd runs on the GPU only: inputs must be CUDA(HIP) tensors")
    B, N, C, H, W = src_feats.shape
    assert H == W  # reference utils/matching.py:35
    if H != 16:
        raise _lib.PicoPoseHipError("the HIP stage-1 kernel is built for 16x16 patch grids")
    assert tar_feat.shape == (B, C, H, W), (tar_feat.shape, src_feats.shape)
    assert tar_mask.dim() == 3 and tar_mask.shape[0] == B
    return B, N, C


def template_scores(src_feats, tar_feat, tar_mask, mode=None, eps=0.0, return_stats=False):
    """sim_avg (B,N) of utils/matching.py:38-66 for a (B,N,C,16,16) bank.  A bank handed over as torch.float16 is read
    as stored (2 bytes per element in HBM — BASELINE configs[4]); the result is what the fp32 path returns on those
    values widened to float."""
    B, N, C = _check_inputs(src_feats, tar_feat, tar_mask)
    mode_id = _MODES[mode or DEFAULT_MODE]
    half = src_feats.dtype == torch.float16
    bank = src_feats.contiguous() if half else src_feats.contiguous().float()
    query = tar_feat.contiguous().float()
    mask = tar_mask.contiguous().float()
    ws, nbytes = _workspace(B, N, C, bank.device)
    sim_avg = torch.empty(B, N, dtype=torch.float32, device=bank.device)
    stats = torch.zeros(4, dtype=torch.int32, device=bank.device) if return_stats else None
    rc = _lib.lib().pp_stage1_scores_ex(
        bank.data_ptr(), _lib.PP_BANK_F16 if half else _lib.PP_BANK_F32, query.data_ptr(), mask.data_ptr(), mask.shape[1],
        mask.shape[2], B, N, C, mode_id, float(eps), ws.data_ptr(), nbytes, sim_avg.data_ptr(),
        stats.data_ptr() if stats is not None else None, _stream_ptr())
    _lib.check(rc, "pp_stage1_scores_ex")
    return (sim_avg, stats) if return_stats else sim_avg


def topk_templates(sim_avg, topk):
    """torch.topk(sim_avg, topk, dim=1) (utils/matching.py:68), ties -> lower id."""
    B, N = sim_avg.shape
    if topk > N:
        raise RuntimeError(f"selected index k out of range: topk={topk} > N={N}")
    sim_avg = sim_avg.contiguous().float()
    score = torch.empty(B, topk, dtype=torch.float32, device=sim_avg.device)
    index = torch.empty(B, topk, dtype=torch.int64, device=sim_avg.device)
    rc = _lib.lib().pp_topk(sim_avg.data_ptr(), B, N, topk, score.data_ptr(), index.data_ptr(), _stream_ptr())
    _lib.check(rc, "pp_topk")
    return score, index


def matching_templates(src_feats, tar_feat, src_masks, tar_mask, topk=5, mode=None):
    """Drop-in for reference utils/matching.py:29-69.

    src_feats (B,N,C,16,16), tar_feat (B,C,16,16), tar_mask (B,H,W);
    `src_masks` is accepted and unused, exactly as in the reference.
    Returns (pred_score_src (B,topk) f32, pred_id_src (B,topk) i64).
    """
    B, N, C = _check_inputs(src_feats, tar_feat, tar_mask)
    if topk > N:
        raise RuntimeError(f"selected index k out of range: topk={topk} > N={N}")
    half = src_feats.dtype == torch.float16
    bank = src_feats.contiguous() if half else src_feats.contiguous().float()
    query, mask = tar_feat.contiguous().float(), tar_mask.contiguous().float()
    ws, nbytes = _workspace(B, N, C, bank.device)
    sim_avg = torch.empty(B, N, dtype=torch.float32, device=bank.device)
    score = torch.empty(B, topk, dtype=torch.float32, device=bank.device)
    index = torch.empty(B, topk, dtype=torch.int64, device=bank.device)
    # scores + top-k as one ABI call (pp_stage1_match_ex: the last resolve workgroup of a crop ranks its scores)
    rc = _lib.lib().pp_stage1_match_ex(
        bank.data_ptr(), _lib.PP_BANK_F16 if half else _lib.PP_BANK_F32, query.data_ptr(), mask.data_ptr(), mask.shape[1],
        mask.shape[2], B, N, C, int(topk), _MODES[mode or DEFAULT_MODE], 0.0, ws.data_ptr(), nbytes, sim_avg.data_ptr(),
        score.data_ptr(), index.data_ptr(), None, _stream_ptr())
    _lib.check(rc, "pp_stage1_match_ex")
    return score, index


class MatchingGraph:
    """`matching_templates` on FIXED buffers as ONE HIP graph: the five launches of a call (query norms, query pre-pack, the fused
    similarity kernel, the near-tie resolve, top-k) are captured once and replayed — at BASELINE configs[1] (batch 8, 42 templates,
    ViT-S) a call is five dependent 5-40 us kernels and the host's launch gaps are a fifth of it (profiles/r05/stage1_small.txt).
    A serving loop with a resident bank has fixed shapes and addresses; the caller refreshes `tar_feat` / `tar_mask` / the bank IN
    PLACE between replays (the graph reads the same addresses).  `__call__()` -> (pred_score_src, pred_id_src), the tensors the
    capture wrote (overwritten by the next replay).  Same kernels, same arguments: bit-identical to `matching_templates`."""

    def __init__(self, src_feats, tar_feat, tar_mask, topk=5, mode=None):
        _check_inputs(src_feats, tar_feat, tar_mask)
        for t in (src_feats, tar_feat, tar_mask):
            if not t.is_contiguous():
                raise _lib.PicoPoseHipError("MatchingGraph reads its inputs in place: they must be contiguous")
        if src_feats.dtype not in (torch.float16, torch.float32) or tar_feat.dtype != torch.float32 or tar_mask.dtype != torch.float32:
            raise _lib.PicoPoseHipError("MatchingGraph reads its inputs in place: fp32 query / mask, fp32 or fp16 bank")
        self.inputs = (src_feats, tar_feat, tar_mask)
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # (first use outside the capture: per-device kernel attributes, the allocator's pool)
            matching_templates(src_feats, tar_feat, None, tar_mask, topk=topk, mode=mode)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(self.graph):
            self.score, self.index = matching_templates(src_feats, tar_feat, None, tar_mask, topk=topk, mode=mode)

    def __call__(self):
        self.graph.replay()
        return self.score, self.index


def matching_features_similarity(src_feat, tar_feat, src_mask, tar_mask):
    """Drop-in for reference utils/matching.py:6-26 (`tar_mask` is unused there as well).

    src_feat/tar_feat (B,C,16,16), src_mask (B,H,W) -> (B,256,16,16)."""
    B, C, H, W = src_feat.shape
    assert H == W  # reference utils/matching.py:10
    if H != 16:
        raise _lib.PicoPoseHipError("the HIP similarity-volume kernel is built for 16x16 patch grids")
    src, tar, mask = _lib.dev_f32(src_feat, tar_feat, src_mask)
    out = torch.empty(B, H * W, H, W, dtype=torch.float32, device=src.device)
    rc = _lib.lib().pp_similarity_volume(src.data_ptr(), tar.data_ptr(), mask.data_ptr(), mask.shape[1],
                                         mask.shape[2], B, C, out.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "pp_similarity_volume")
    return out
