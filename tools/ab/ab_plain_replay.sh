# the REPLAY-arithmetic legs (seed-compatible kernels on device-resident streams) with the whole library specialised to the
# Poyiadjis score (-DPFG_EXP_PLAIN=1) against the general build
cd /root/repo
L=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
for rep in 1 2; do
for c in c2 c1 c3; do
for lib in libpfgrad.so libpfgrad_plain.so; do
  PFGRAD_LIB=$L/$lib timeout -k 10 150 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ra=r.get('replay_arithmetic') or {}
print('$c $lib rep$rep replay leg', ra.get('kernel_ms'), ra.get('value'), ra.get('kernel_variant'))"
done; done; done
