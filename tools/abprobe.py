"""A/B of builds on one box: python tools/abprobe.py [--config N] libA.so libB.so ...  (each run in its own process, interleaved)"""
import subprocess, sys, os, re
args = sys.argv[1:]
config = "2"
if args and args[0] == "--config":
    config = args[1]; args = args[2:]
libs = args
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        env = dict(os.environ)
        lib, _, knobs = l.partition("@")          # lib.so@LATE=2,FOO=1 sets experiment env vars RANENV_LATE, RANENV_FOO
        for kv in filter(None, knobs.split(",")):
            k, _, v = kv.partition("=")
            env["RANENV_" + k] = v
        if lib != "default":
            env["RANENV_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "tools/benchprobe.py", config], env=env, capture_output=True, text=True)
        m = re.search(r"step\s+([\d.]+) us\s+kernel\s+([\d.]+)(?:\s+rollout x\d+\s+([\d.]+))?", out.stdout)
        if not m:
            print(l, "FAILED", out.stdout[-300:], out.stderr[-600:], flush=True)
            continue
        res[l].append(tuple(float(x) if x else 0.0 for x in m.groups()))
        print(l, res[l][-1], flush=True)
for l in libs:
    a = res[l]
    if a:
        print(f"{l:32s} step {sum(x[0] for x in a) / len(a):6.1f}  kernel {sum(x[1] for x in a) / len(a):6.1f}  rollout {sum(x[2] for x in a) / len(a):6.1f}   (min step {min(x[0] for x in a):.1f})")
