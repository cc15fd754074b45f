# A/B of a tagged library on the whole-GPU window: tools/ab/ab_grid_lib.sh <tag>   (device generator, SVM, 48 steps)
cd /root/repo
L=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
for lib in libpfgrad.so libpfgrad_$1.so; do
  echo "== $lib"
  PFGRAD_LIB=$L/$lib GRID_TIME_REPLAY=0 GRID_TIME_MODELS=svm timeout -k 10 200 python tools/grid_time.py ${GRID_AB_NS:-600000 1000000} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if r['model']=='svm': print(r['N'], r['B'], '%.2f us/step frac %.3f'%(r['us_per_step'], r['frac_of_8TBps']), r['grad'][:2])"
  PFGRAD_LIB=$L/$lib timeout -k 10 100 python bench.py --config g1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g1', r['value'], r['roofline']['frac'], r['config']['kernel_variant'])"
done
