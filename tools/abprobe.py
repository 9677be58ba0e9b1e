"""A/B of two builds on one box: python tools/abprobe.py libA.so libB.so  (each run in its own process, interleaved)"""
import subprocess, sys, os, re
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rep in range(4):
    for l in libs:
        env = dict(os.environ, RANENV_LIB=l)
        out = subprocess.run([sys.executable, "tools/benchprobe.py"], env=env, capture_output=True, text=True).stdout
        m = re.search(r"step\s+([\d.]+) us\s+alloc\s+([\d.]+)\s+core\s+([\d.]+)", out)
        res[l].append(tuple(float(x) for x in m.groups()))
        print(l, res[l][-1], flush=True)
for l in libs:
    a = res[l]
    print(f"{l:28s} step {sum(x[0] for x in a) / len(a):6.1f}  alloc {sum(x[1] for x in a) / len(a):5.1f}  core {sum(x[2] for x in a) / len(a):5.1f}")
