"""Time the PF kernel per variant through the resident path (HIP events), one process."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["wg256x4", "wg512x2", "wg1024x1", "wg256x4s"]
chains = sys.argv[2].split(",") if len(sys.argv) > 2 else ["512", "1024"]
extra = sys.argv[3:] 
for v in variants:
    for c in chains:
        env = dict(os.environ, PFGRAD_VARIANT=v)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1",
                              "--no-cpu-baseline", "--chains-per-gpu", c] + extra, env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(v, "C=" + c, j["config"]["kernel_variant"], "kernel_ms %.3f" % j["roofline"]["kernel_ms"],
                  "steps/s %.0f" % j["value"], "frac %.3f" % j["roofline"]["frac"], flush=True)
        except Exception as e:
            print(v, c, "FAILED", out.stderr[-500:], flush=True)
