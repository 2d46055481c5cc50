for lib in - lightgaussian_amd/variants/lib_onestage.so; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
  for extra in "" "--scene heavy" "--scale 0.012"; do
  timeout -s KILL 300 python bench.py --mode fwdbwd --no-cpu-baseline --no-literal $extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms',{}); print('$lib','$extra', d['value'], {a:k[a] for a in k if a in ('sort','tile_ranges','tile_sort','duplicate','scan')})"
  done
done
