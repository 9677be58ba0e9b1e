"""Static VALU / SALU / LDS / VMEM instruction counts per source file and line of one kernel (assembly from
hipcc --cuda-device-only -gline-tables-only -S):
    python tools/isa_lines.py <file.s> <kernel substr> <template substr> [bucket] [file-substr]
With a bucket size the lines are grouped (e.g. 10 = per 10 source lines).  Inlined callees count at their own lines."""
import collections, re, sys
src = open(sys.argv[1]).read()
want, targ = sys.argv[2], sys.argv[3]
bucket = int(sys.argv[4]) if len(sys.argv) > 4 else 1
only = sys.argv[5] if len(sys.argv) > 5 else ""
files = {int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', src)}
parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", src)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if want not in name or targ not in name:
        continue
    body = body.split(".Lfunc_end")[0]
    cur = (0, 0)
    cnt = collections.defaultdict(lambda: [0, 0, 0, 0])
    for line in body.split("\n"):
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (int(m.group(1)), int(m.group(2))); continue
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", line)
        if not m:
            continue
        op = m.group(1)
        k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3 if op.startswith(("global_", "buffer_", "scratch_")) else -1
        if k >= 0:
            cnt[(cur[0], cur[1] // bucket * bucket)][k] += 1
    print(name)
    tot = collections.defaultdict(lambda: [0, 0, 0, 0])
    for (f, ln) in sorted(cnt):
        v = cnt[(f, ln)]
        fn = files.get(f, str(f)).split("/")[-1]
        for k in range(4):
            tot[fn][k] += v[k]
        if sum(v) >= 8 and only in fn:
            print(f"  {fn:22s} line {ln:5d}: valu {v[0]:4d} salu {v[1]:4d} lds {v[2]:3d} vmem {v[3]:3d}")
    for fn, v in tot.items():
        print(f"  TOTAL {fn:22s}: valu {v[0]:5d} salu {v[1]:5d} lds {v[2]:4d} vmem {v[3]:4d}")
