"""Start / end times of every workgroup of one PF launch (s_memrealtime stamps, 100 MHz): where does the
launch's fixed cost (kernel time = 1.0 ms + 4.83 ms per 1024 chains) sit -- start-up ramp, tail, or both?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
import bench
from sgmcmc_ssm_amd.ensemble import ChainEnsemble
for C in (1024, 3072):
    p0, y, prior, cfg = bench.make_workload("svm")
    ens = ChainEnsemble("svm", y, p0, num_chains=C, N=1000, kernel="prior", epsilon=0.1, prior=prior, seed=3)
    ens.step(2); ens.synchronize(); ens.enable_stamps()
    st = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); ens.launch_pf(st); b.record(st); ens.synchronize()
    s = ens.stamps_dev.cpu().numpy().astype(np.int64)
    t0, t1 = s[:, 1], s[:, 3]
    base = t0.min()
    start, end = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
    dur = end - start
    print(f"C={C}: kernel {a.elapsed_time(b):.3f} ms; WG start: min {start.min():.1f} med {np.median(start):.1f} max {start.max():.1f} us;"
          f" WG end: min {end.min():.1f} med {np.median(end):.1f} max {end.max():.1f} us; WG duration: min {dur.min():.1f} med {np.median(dur):.1f} max {dur.max():.1f} us")
    q = np.percentile(dur, [1, 10, 50, 90, 99])
    print("   duration percentiles 1/10/50/90/99:", np.round(q, 1), " first-wave WGs (start < 50 us):", int((start < 50).sum()))
    if C == 3072:
        order = np.argsort(start)
        for lo, hi in ((0, 1024), (1024, 2048), (2048, 3072)):
            idx = order[lo:hi]
            print(f"   WGs {lo}-{hi} by start: start med {np.median(start[idx]):.0f} us, duration med {np.median(dur[idx]):.0f} us, end max {end[idx].max():.0f} us")
    if C == 1024:
        b = np.arange(C)
        print("   median duration by blockIdx % 8 (XCD):", [int(np.median(dur[b % 8 == k])) for k in range(8)])
        print("   median duration by (blockIdx // 8) % 4:", [int(np.median(dur[(b // 8) % 4 == k])) for k in range(4)])
        print("   median duration by blockIdx // 256:", [int(np.median(dur[b // 256 == k])) for k in range(4)])
        clk = (s[:, 2] - s[:, 0]) / np.maximum(s[:, 3] - s[:, 1], 1) * 0.1
        print("   in-kernel clock GHz by XCD:", [round(float(np.median(clk[b % 8 == k])), 3) for k in range(8)])
        print("   duration of first 32 blocks:", dur[:32].astype(int).tolist())
