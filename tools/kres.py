"""Per-kernel resource usage of csrc/ranenv.hip (hipcc -Rpass-analysis=kernel-resource-usage): python tools/kres.py [filter-regex] [extra hipcc flags...]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else "."
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", "--cuda-device-only", "-Wno-pass-failed",
       "-ffp-contract=off", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-mllvm", "-disable-machine-licm",
       "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(REPO, "include"),
       os.path.join(REPO, "intent_radio_sched_multi_slice_amd", "csrc", "ranenv.hip"), "-o", "/tmp/kres.o"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for l in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\w+) \[-Rpass", l)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
if not rows:
    print(err[-3000:])
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    if not re.search(flt, name):
        continue
    print(f"{name[:64]:64s} VGPR {v.get('VGPRs', '?'):>3} spill {v.get('VGPRs Spill', '?'):>3}  SGPR {v.get('TotalSGPRs', '?'):>3} spill {v.get('SGPRs Spill', '?'):>3}"
          f"  scratch {v.get('ScratchSize', '?'):>4}  occ {v.get('Occupancy', '?')}  lds {v.get('LDS Size', '?')}")
