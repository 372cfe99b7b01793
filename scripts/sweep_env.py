"""Runs bench.py once per value of an environment variable and prints one line each.
usage: sweep_env.py VAR v1 v2 ... [-- extra bench args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
var, values = args[0], args[1:]
for v in values:
    env = dict(os.environ)
    if v != "default":
        env[var] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--steps", "64"] + extra,
                         env=env, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("%s=%s: %.1f Mrays/s  %s" % (var, v, d["value"], d["roofline"]["frame_ms"]), flush=True)
    except Exception:
        print("%s=%s: failed: %s" % (var, v, out.stderr[-400:]), flush=True)
