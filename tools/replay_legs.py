"""The REPLAY-arithmetic legs of every BASELINE config (tools; GPU box): python tools/replay_legs.py [c4 c5 ...]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
for cfg in sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5"]:
    w = bench.config_workload(cfg)
    out = bench.replay_arithmetic_leg(w, torch.cuda.current_device(), C=int(os.environ['REPLAY_LEG_C']) if os.environ.get('REPLAY_LEG_C') else None)
    out["config"] = cfg
    print(json.dumps(out), flush=True)
