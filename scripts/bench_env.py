"""One bench configuration under several values of an environment switch of the library (GPU box):
    python scripts/bench_env.py VAR v1,v2,.. [config] [extra bench.py arguments]
e.g. python scripts/bench_env.py SMR_SYNC_RESET 0,1 space --envs-per-gpu 262144"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, vals = sys.argv[1], sys.argv[2].split(",")
cfg = sys.argv[3] if len(sys.argv) > 3 else "space"
for val in vals:
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "100", "--warmup", "20", "--no-cpu-baseline"] + sys.argv[4:],
                       env=dict(os.environ, **{var: val}), capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print("%s=%s  %.2f M env-steps/s" % (var, val, d["value"] / 1e6), {k: round(v, 4) for k, v in d["kernel_ms"].items()}, flush=True)
    except Exception:
        print("%s=%s failed: %s" % (var, val, p.stderr[-300:]), flush=True)
