#!/usr/bin/env python3
"""Which device operation order reproduces torch's getters bit for bit?  (the fused-getter path evaluates
exp / sigmoid / normalize inside K1; a last-bit difference against torch.exp / torch.sigmoid / F.normalize can move a radius by
one.)  Prints, per candidate, the number of elements whose bits differ from torch's result on this GPU."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
n = 3_000_000
g = torch.Generator().manual_seed(0)
s = (torch.randn(n, generator=g) * 0.5 - 5.5).to(dev)
r = torch.randn(n, 4, generator=g).to(dev)
o = (torch.randn(n, generator=g) * 1.5 - 1.0).to(dev)
out_s = torch.empty(n, device=dev); out_r = torch.empty(4, n, 4, device=dev); out_o = torch.empty(2, n, device=dev)
_lib.check(lib.lg_debug_activations(n, s.data_ptr(), r.data_ptr(), o.data_ptr(), out_s.data_ptr(), out_r.data_ptr(), out_o.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
diff = lambda a, b: int((a.view(torch.int32) != b.view(torch.int32)).sum())
print("exp      expf:", diff(out_s, torch.exp(s)), "of", n)
print("sigmoid  1/(1+expf(-x)):", diff(out_o[0], torch.sigmoid(o)), " 1/(1+__expf(-x)):", diff(out_o[1], torch.sigmoid(o)))
ref = torch.nn.functional.normalize(r)
for v, name in enumerate(["((a2+b2)+c2)+d2", "(a2+b2)+(c2+d2)", "fma chain", "(a2+c2)+(b2+d2)"]):
    print(f"normalize {name}: {diff(out_r[v], ref)} of {4 * n}")
nrm = r.norm(dim=1, keepdim=True)
print("torch's own alternatives: x / x.norm():", diff(r / nrm.clamp_min(1e-12), ref), " x * (1/norm):", diff(r * (1.0 / nrm.clamp_min(1e-12)), ref))
