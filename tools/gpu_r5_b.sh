#!/bin/bash
# Round 5, second GPU call: the parallel count walk, the K7 quad reduction and the K6 unroll as A/B variants.
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r5b.log; : > $L
say() { echo "$@" | tee -a $L; }
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/r5b.err | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); c=d['config']; print('$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', 'burst', (d.get('contract_region') or {}).get('views_per_s'), 'R', c['tile_instances'], d.get('kernels_ms'))" | cut -c1-700 | tee -a $L; }
say "== new tests"
timeout -s KILL 900 python -m pytest tests/test_gpu_round5.py -q --tb=short 2>&1 | tail -30 | tee -a $L
say "== whole suite"
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_gpu_round5.py 2>&1 | tail -25 > gpurun_out/r5b_pytest.log; grep -E "passed|failed" gpurun_out/r5b_pytest.log | tail -1 | tee -a $L; grep -E "^FAILED|^E  " gpurun_out/r5b_pytest.log | cut -c1-300 | head -20 | tee -a $L
say "== variants"
for rep in 1 2; do
  for lib in - lightgaussian_amd/variants/lib_k7_quad.so lightgaussian_amd/variants/lib_k6_unroll2.so; do
    if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
    say "lib $lib"; run --mode fwdbwd --steps 100 --no-literal
    if [ "$lib" != "lightgaussian_amd/variants/lib_k7_quad.so" ]; then run --mode count --steps 100; fi
  done
done
unset LIGHTGAUSSIAN_HIP_LIB
say "== parity of the quad reduction (variant library through the backward tests)"
LIGHTGAUSSIAN_HIP_LIB=$PWD/lightgaussian_amd/variants/lib_k7_quad.so timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -q -k "backward_parity or determin" --tb=short 2>&1 | tail -3 | tee -a $L
say "== significance pass: parallel long-tile walk (default rule) vs serial"
for rep in 1 2; do
  run --mode count --steps 100 --scene heavy
  run --mode count --steps 100 --scene heavy --long-tiles serial
  run --mode count --steps 100
  run --mode count --steps 100 --long-tiles serial
done
run --mode count --steps 100 --scale 0.012
run --mode count --steps 100 --scale 0.012 --long-tiles serial
timeout -s KILL 300 python tools/gpu_fuzz.py 60 2>&1 | tail -2 | cut -c1-300 | tee -a $L
