#!/bin/bash
# round 3, call A: the whole -m gpu suite on the stateless ABI v5 library, smoke, the default bench line, and the measurements that
# decide the K7 / K6 work: segment length sweep (more, shorter work items for the backward blend), long-tile mode cost on the
# uniform scene, packed-f32 / dependent-chain micro-benchmark
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -40 > gpurun_out/r3a_pytest.log; grep -E "passed|failed|error" gpurun_out/r3a_pytest.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r3a_pytest.log | head -20
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout -s KILL 900 python bench.py ) > gpurun_out/r3a_bench_default.log 2>&1; tail -4 gpurun_out/r3a_bench_default.log | cut -c1-3000
run() { timeout -s KILL 400 python bench.py --no-cpu-baseline --no-literal "$@" 2>&1 | tail -1 > gpurun_out/r3a_tmp.json; python - "$*" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r3a_tmp.json"))
    print(sys.argv[1], "->", d["value"], "views/s", d["ms_per_step"], "ms", "contract", (d.get("contract_region") or {}).get("views_per_s"), "kernels", d.get("kernels_ms"), flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/r3a_tmp.json").read()[-1500:])
PY
}
run --steps 100
run --steps 100 --long-tiles serial
run --steps 100 --segment-length 512
run --steps 100 --segment-length 256
run --steps 100 --segment-length 128
run --steps 60 --scene heavy
run --steps 60 --scene heavy --long-tiles serial
run --steps 60 --mode count
./tools/ubench/valu_rate4 2>&1 | tail -30
