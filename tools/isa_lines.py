"""Static VALU / SALU / LDS / VMEM instruction counts per source line of one kernel (assembly from
hipcc -gline-tables-only --save-temps):  python tools/isa_lines.py <file.s> <kernel substr> <template substr> [bucket]
With a bucket size the lines are grouped (e.g. 10 = per 10 source lines)."""
import collections, re, sys
src = open(sys.argv[1]).read()
want, targ = sys.argv[2], sys.argv[3]
bucket = int(sys.argv[4]) if len(sys.argv) > 4 else 1
parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", src)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if want not in name or targ not in name:
        continue
    body = body.split(".Lfunc_end")[0]
    cur = 0
    cnt = collections.defaultdict(lambda: [0, 0, 0, 0])
    for line in body.split("\n"):
        m = re.match(r"\s+\.loc\s+\d+\s+(\d+)", line)
        if m:
            cur = int(m.group(1)); continue
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", line)
        if not m:
            continue
        op = m.group(1)
        k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3 if op.startswith(("global_", "buffer_", "scratch_")) else -1
        if k >= 0:
            cnt[cur // bucket * bucket][k] += 1
    print(name)
    for ln in sorted(cnt):
        v = cnt[ln]
        if sum(v) >= 8:
            print(f"  line {ln:5d}: valu {v[0]:4d} salu {v[1]:4d} lds {v[2]:3d} vmem {v[3]:3d}")
    print("  total valu", sum(v[0] for v in cnt.values()))
