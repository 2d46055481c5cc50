#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_knn.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_knn.log
python - <<'PY'
import time, torch, numpy as np
from simple_knn._C import distCUDA2
for n in (100000, 1000000, 3000000):
    for kind in ("uniform", "clustered"):
        g = torch.Generator().manual_seed(0)
        p = torch.rand(n, 3, generator=g) if kind == "uniform" else (torch.randn(n, 3, generator=g) * 0.02 + torch.randint(0, 20, (n, 1), generator=g).float() * torch.tensor([[1.0, 0.37, 2.1]]))
        p = p.cuda(); distCUDA2(p); torch.cuda.synchronize()
        t0 = time.perf_counter(); d = distCUDA2(p); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"knn {kind} P={n}: {dt*1e3:.2f} ms  mean d2 {float(d.mean()):.3e}")
PY
echo "PYTEST: $(grep -E 'passed|failed' gpurun_out/pytest_knn.log | tail -1)"; grep -E "Error|assert" gpurun_out/pytest_knn.log | head -5
