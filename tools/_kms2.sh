for rep in 1 2; do
for lib in - lightgaussian_amd/variants/lib_onestage.so; do
  if [ "$lib" = "-" ]; then unset LIGHTGAUSSIAN_HIP_LIB; else export LIGHTGAUSSIAN_HIP_LIB=$PWD/$lib; fi
  for extra in "--mode count" "--mode count --scene heavy" "--mode fwd" "--mode fwdbwd --n-gaussians 6000000 --width 1600 --height 1060 --sh-degree 2 --steps 50"; do
  timeout -s KILL 300 python bench.py --no-cpu-baseline --no-literal --no-roofline $extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib','$extra', d['value'])"
  done
done
done
