for rep in 1 2; do
for mode in on off; do
  if [ $mode = off ]; then export LG_NO_SH_JACOBIAN=1; else unset LG_NO_SH_JACOBIAN; fi
  timeout -s KILL 300 python bench.py --mode fwdbwd --no-cpu-baseline --no-literal 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels_ms', {})
print('jacobian $mode', 'value', d['value'], 'ms', d['ms_per_step'], 'k1', k.get('preprocess'), 'k9', k.get('preprocess_bwd'), 'bwd', k.get('blend_bwd'))"
done; done
