"""Where the trainer-facing schedules lose time: from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K --no-gather --no-other-configs --no-cpu-baseline` (headline rollout, then the env.step() loop -- one launch of
ranenv_core_kernel_mixed<NP, false, false> per TTI on one stream --, then the learner loop -- ranenv_core_kernel<0, NP, false> per range and
TTI on the ranges' own streams, a torch policy kernel in between):
    python tools/steploop_timeline.py <p_kernel_trace.csv> [K]
Per schedule: kernel duration, GPU idle between consecutive launches of a queue (end -> next start inside a timed block), and for the learner
loop how much of the span has 0 / 1 / 2 step kernels in flight."""
import csv, re, sys, collections
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def launches(pattern):
    return sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if re.search(pattern, r["Kernel_Name"]))


def blocks_of(ks, pause_ns=30_000):
    out, cur = [], [ks[0]]
    for k in ks[1:]:
        if k[0] - max(e for _, e, _ in cur) > pause_ns:
            out.append(cur); cur = [k]
        else:
            cur.append(k)
    out.append(cur)
    return out


def med(x):
    return float(np.median(x)) if len(x) else float("nan")


step = launches(r"ranenv_core_kernel_mixed<\d+, false, false>")
if step:
    bl = [b for b in blocks_of(step) if len(b) == K][1:]
    dur = [(e - s) / 1e3 for b in bl for s, e, _ in b]
    gap = [(b[i + 1][0] - b[i][1]) / 1e3 for b in bl for i in range(len(b) - 1)]
    span = [(b[-1][1] - b[0][0]) / 1e3 / K for b in bl]
    print(f"env.step() loop (mixed one-TTI launches, one stream): {len(bl)} blocks of {K}; kernel {med(dur):.1f} us (p10 {np.percentile(dur, 10):.1f}, p90 {np.percentile(dur, 90):.1f}); "
          f"GPU idle between two TTIs {med(gap):.1f} us (p90 {np.percentile(gap, 90):.1f}); span per TTI {med(span):.1f} us -> the launch boundary is "
          f"{100 * med(gap) / med(span):.1f} % of a TTI")
rng = launches(r"ranenv_core_kernel<0, \d+, false>")
if rng:
    bl = [b for b in blocks_of(rng, 60_000) if len(b) >= 2 * K - 2]
    bl = bl[1:] if len(bl) > 1 else bl
    dur, gap, frac0, frac1, frac2, per_tti = [], [], [], [], [], []
    for b in bl:
        byq = collections.defaultdict(list)
        for s, e, q in b:
            byq[q].append((s, e))
        for q, v in byq.items():
            v.sort()
            dur += [(e - s) / 1e3 for s, e in v]
            gap += [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
        t0, t1 = min(s for s, _, _ in b), max(e for _, e, _ in b)
        ev = sorted([(s, 1) for s, _, _ in b] + [(e, -1) for _, e, _ in b])
        n, last, acc = 0, t0, [0, 0, 0]
        for t, d in ev:
            acc[min(n, 2)] += t - last; last = t; n += d
        tot = float(t1 - t0)
        frac0.append(acc[0] / tot); frac1.append(acc[1] / tot); frac2.append(acc[2] / tot)
        per_tti.append(tot / 1e3 / (len(b) / len(byq)))
    print(f"learner loop ({len(set(q for b in bl for _, _, q in b))} ranges, each an in-order chain TTI -> policy -> TTI on its own queue): {len(bl)} blocks; step kernel of a range "
          f"{med(dur):.1f} us; on a range's queue the next TTI starts {med(gap):.1f} us after the last ended (policy kernel + two launch boundaries; p90 {np.percentile(gap, 90):.1f}); "
          f"span per TTI of all ranges {med(per_tti):.1f} us; step kernels in flight: none {100 * med(frac0):.1f} %, one {100 * med(frac1):.1f} %, two or more {100 * med(frac2):.1f} % of the span")
