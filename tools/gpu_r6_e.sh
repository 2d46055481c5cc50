#!/bin/bash
# round 6, call E: lg_score_kernel as a grid-stride loop over N with G persistent workgroups (G = 512 .. 4096) against one thread per Gaussian
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout -s KILL 600 python bench.py --no-cpu-baseline --n-gaussians 3000000 --mode count --steps 100 "$@" 2>&1 | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip()); print('${LIGHTGAUSSIAN_HIP_LIB##*/}', '$*', '->', d['value'], 'views/s', d['ms_per_step'], 'ms', d.get('kernels_ms'))" | cut -c1-600; }
for rep in 1 2; do
for lib in - triv now; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/lightgaussian_amd/variants/lib_$lib.so; fi
  run
done
done
