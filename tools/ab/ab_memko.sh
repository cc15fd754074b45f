cd /root/repo
L=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
for rep in 1 2; do for lib in libpfgrad.so libpfgrad_memko.so; do for c in c4 c5; do
  PFGRAD_LIB=$L/$lib timeout -k 10 150 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ra=r.get('replay_arithmetic') or {}
print('$c $lib rep$rep replay leg', ra.get('kernel_ms'), ra.get('value'), ra.get('kernel_variant'))"
done; done; done
