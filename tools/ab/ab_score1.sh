# A/B of the PFG_SMOOTHER_POYIADJIS_N twin of the 1024 x 4 unit on one box: bench.py --config c4 with / without it
cd /root/repo
for rep in 1 2 3; do
for off in 0 1; do
  PFGRAD_NO_SCORE1=$off timeout -k 10 120 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline --no-single-chain 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 no_score1=$off rep$rep', r['ms_per_step'], r['value'], r['config']['kernel_variant'])"
done; done
