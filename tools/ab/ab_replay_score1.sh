# A/B on one box: the REPLAY score-only twins (PFGRAD_NO_SCORE1=1: the general kernels) on bench.py's replay_arithmetic legs
cd /root/repo
for rep in 1 2; do for off in 0 1; do
for c in ${AB_CFGS:-c2 c1 c3}; do
  PFGRAD_NO_SCORE1=$off timeout -k 10 150 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ra=r.get('replay_arithmetic') or {}
print('$c no_score1=$off rep$rep replay leg', ra.get('kernel_ms'), ra.get('value'), ra.get('kernel_variant'))"
done; done; done
