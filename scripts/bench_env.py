"""One scene under several values of an environment switch of the library (GPU box): python scripts/bench_env.py VAR v1,v2,.. [config] [extra VAR=val ...]"""
import os, subprocess, sys
var = sys.argv[1]; vals = sys.argv[2].split(","); cfg = sys.argv[3] if len(sys.argv) > 3 else "space"
extra = dict(a.split("=") for a in sys.argv[4:])
code = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_ab.py")).read().split("code = r'''")[1].split("'''")[0]
for val in vals:
    env = dict(os.environ, **extra, **{var: val})
    print(var + "=" + val, subprocess.run([sys.executable, "-c", code, cfg], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
