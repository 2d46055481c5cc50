"""Nearest-code search of the VecTree vector quantiser on the MI355X matrix cores (second half).

Replaces the two lines of the reference's EuclideanCodebook.forward that hold all of its arithmetic weight
(vectree/vq.py:265-266):
    dist = -torch.cdist(flatten, embed, p = 2)
    embed_ind = gumbel_sample(dist, dim = -1, temperature = self.sample_codebook_temp)      # temperature 0 -> dist.argmax(-1)
as driven by vectree/vectree.py:87-101 (chunks of 8192 feature rows against the 8192-entry codebook, 27 or 48 dimensions)
and :176-186 (the EMA k-means iterations on 80 000 sampled rows).  The kernel (csrc/lg_vq.h) evaluates |c|^2 - 2 x.c with
exact-f32 MFMA (v_mfma_f32_32x32x2_f32) and keeps the running argmin in registers; ties go to the lowest code index.
"""
import ctypes as C

import torch

from . import _lib


def nearest_code(flatten, embed):
    """embed_ind = (-torch.cdist(flatten, embed, p=2)).argmax(-1) for flatten [h, n, d] (or [n, d]) and embed [h, K, d] (or
    [K, d]): int64 indices [h, n] (or [n]).  HIP tensors only -- no CPU / torch fallback."""
    squeeze = flatten.dim() == 2
    x = flatten.unsqueeze(0) if squeeze else flatten
    cb = embed.unsqueeze(0) if embed.dim() == 2 else embed
    if x.dim() != 3 or cb.dim() != 3 or x.shape[0] != cb.shape[0] or x.shape[2] != cb.shape[2]:
        raise ValueError(f"nearest_code: incompatible shapes {tuple(flatten.shape)} / {tuple(embed.shape)}")
    if not (x.is_cuda and cb.is_cuda):
        raise RuntimeError("nearest_code runs on the MI355X HIP library only (no CPU fallback)")
    lib = _lib.load()
    h, n, d = x.shape
    K = cb.shape[1]
    nbytes = lib.lg_vq_scratch_bytes(K, d)
    if nbytes == 0:
        raise Exception(f"nearest_code: unsupported shape (K={K}, d={d}; need K >= 1, 1 <= d <= 63)")
    dev = x.device
    out = torch.empty((h, n), dtype=torch.int32, device=dev)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for i in range(h):
        xi = x[i].detach().contiguous().float()
        ci = cb[i].detach().contiguous().float()
        _lib.check(lib.lg_vq_nearest(n, d, K, xi.data_ptr(), ci.data_ptr(), out[i].data_ptr(), scratch.data_ptr(), 0, stream))
    out = out.long()
    return out[0] if squeeze else out


def quantize(flatten, embed):
    """(quantized rows, indices): the codebook lookup that follows the search (vectree/vq.py:269 batched_embedding)."""
    ind = nearest_code(flatten, embed)
    if embed.dim() == 2:
        return embed[ind], ind
    return torch.stack([embed[i][ind[i]] for i in range(embed.shape[0])]), ind
