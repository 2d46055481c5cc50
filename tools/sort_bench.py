#!/usr/bin/env python3
"""Micro-benchmark of K4 alone (lg_debug_sort_keys): n random 64-bit keys, bits [begin, end).  Usage: sort_bench.py [n] [begin] [end]
(LIGHTGAUSSIAN_HIP_LIB selects the library build)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_141_089
begin = int(sys.argv[2]) if len(sys.argv) > 2 else 29
end = int(sys.argv[3]) if len(sys.argv) > 3 else 61
lib = _lib.load()
dev = torch.device("cuda:0")
keys = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, device=dev)
out = torch.empty_like(keys)
temp = torch.empty(lib.lg_debug_sort_temp_bytes(n), dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    _lib.check(lib.lg_debug_sort_keys(n, keys.data_ptr(), out.data_ptr(), begin, end, temp.data_ptr(), st))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
a.record()
for _ in range(reps):
    lib.lg_debug_sort_keys(n, keys.data_ptr(), out.data_ptr(), begin, end, temp.data_ptr(), st)
b.record()
torch.cuda.synchronize()
ok = bool(torch.equal(out, keys[torch.sort((keys >> begin) & ((1 << (end - begin)) - 1), stable=True).indices]))
print(f"{os.environ.get('LIGHTGAUSSIAN_HIP_LIB', 'default'):60s} n={n} bits=[{begin},{end}) {a.elapsed_time(b) / reps * 1e3:8.1f} us per sort "
      f"(memset + histogram + {(end - begin + 7) // 8} passes)  correct={ok}")
