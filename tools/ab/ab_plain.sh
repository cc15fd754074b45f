cd /root/repo
L=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
for rep in 1 2; do
for c in ${AB_CFGS:-c4 c2 c3}; do
for lib in libpfgrad.so libpfgrad_plain.so; do
  PFGRAD_LIB=$L/$lib timeout -k 10 120 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-single-chain 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $lib rep$rep', r['ms_per_step'], r['value'])"
done; done; done
