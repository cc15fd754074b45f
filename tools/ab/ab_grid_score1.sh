# A/B of the score-only twin of the whole-GPU step kernel on one box (PFGRAD_NO_SCORE1=1: the general kernel)
cd /root/repo
for rep in 1 2; do
for off in 0 1; do
  PFGRAD_NO_SCORE1=$off timeout -k 10 120 python bench.py --config g1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g1 no_score1=$off rep$rep', r['value'], r['roofline']['frac'], r['config']['kernel_variant'])"
  PFGRAD_NO_SCORE1=$off GRID_TIME_REPLAY=0 timeout -k 10 200 python tools/grid_time.py 100000 400000 1000000 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    print('   no_score1=$off', r['model'], r['N'], r['B'], '%.2f us/step frac %.3f'%(r['us_per_step'], r['frac_of_8TBps']))"
done; done
