for i in 1 2; do for T in base new; do
  if [ $T = new ]; then L=$PWD/polar_amd/libpolar_amd.so; else L=$PWD/polar_amd/libpolar_amd_base.so; fi
  POLAR_AMD_LIB=$L python bench.py --steps 4 --warmup 1 --cpu-sample 0 --mc-trials 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$T headline', round(d['value']), ' '.join('%s=%d'%(v['config'],round(v['value'])) for v in d.get('other_configs',[])))"
done; done
