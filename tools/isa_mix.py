"""Static instruction mix of one kernel of the HIP library's gfx950 assembly (hipcc --save-temps):
    python tools/isa_mix.py <file.s> <kernel name substring> [<template arg, e.g. ILi0E>]"""
import collections, re, sys
src = open(sys.argv[1]).read()
want, targ = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", src)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if want not in name or targ not in name:
        continue
    body = body.split(".Lfunc_end")[0]
    ops = collections.Counter()
    for line in body.split("\n"):
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", line)
        if m and not m.group(1).startswith(("s_nop",)):
            ops[m.group(1)] += 1
    cls = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
    print(name, "total", sum(ops.values()), "| valu", cls("v_"), "salu", cls("s_"), "lds", cls("ds_"), "vmem", cls(("global_", "buffer_", "scratch_")))
    for o, c in ops.most_common(40):
        print(f"  {o:30s}{c}")
