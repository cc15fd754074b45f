"""Kernel time of the REPLAY-arithmetic instantiation on device-resident streams (bench.py's replay_arithmetic leg) for the
library PFGRAD_LIB selects.  usage: python tools/replay_leg.py [chains]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import bench
w = bench.config_workload("c2")
r = bench.replay_arithmetic_leg(w, 0, C=int(sys.argv[1]) if len(sys.argv) > 1 else 768, reps=5)
r["lib"] = os.path.basename(os.environ.get("PFGRAD_LIB", "libpfgrad.so"))
print(json.dumps(r))
