export TMPDIR=/tmp
cp lightgaussian_amd/liblightgaussian_hip.so /tmp/orig.so
for k in 0 1 3 5; do cp lightgaussian_amd/liblg_run$k.so lightgaussian_amd/liblightgaussian_hip.so; timeout 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run=2^$k', d['value'], d['kernels_ms']['blend_bwd'], d['kernels_ms']['blend_fwd'])"; done
cp /tmp/orig.so lightgaussian_amd/liblightgaussian_hip.so; timeout 300 python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run=2^2', d['value'], d['kernels_ms']['blend_bwd'], d['kernels_ms']['blend_fwd'])"
