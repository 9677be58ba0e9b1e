"""What a block of K TTIs between two device syncs spends outside the steady state: from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K --no-gather --no-single-stream --no-other-configs --no-cpu-baseline` (the headline's rollout: launches over 3 partitions,
or -- rollouts of 16...64 TTIs, the auto rule -- one persistent launch per workgroup class).
    python tools/block_timeline.py <p_kernel_trace.csv> <K> [steady-state us per TTI]
Launches of the step kernel are grouped into blocks (a pause of > 30 us on every queue = the host's sync between two blocks); per
block: span from the first launch's start to the last launch's end, per queue the first launch's us per TTI against the later ones',
gaps between consecutive launches of a queue, and how long the other queues had finished before the last one did."""
import csv, re, sys, collections
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(r"ranenv_core_kernel\w*<0[,>]|ranenv_persist_kernel", r["Kernel_Name"])]
K = int(sys.argv[2])
steady = float(sys.argv[3]) if len(sys.argv) > 3 else None
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows)
blocks, cur = [], [ks[0]]
for k in ks[1:]:
    if k[0] - max(e for _, e, _ in cur) > 30_000:
        blocks.append(cur); cur = [k]
    else:
        cur.append(k)
blocks.append(cur)
sizes = collections.Counter(len(b) for b in blocks)
n_typ = sizes.most_common(1)[0][0]
blocks = [b for b in blocks if len(b) == n_typ][2:]          # the timed blocks (the commonest launch count), warm-up dropped
print(f"{len(blocks)} blocks of {n_typ} launches each")
span = np.array([max(e for _, e, _ in b) - min(s for s, _, _ in b) for b in blocks]) / 1e3
print(f"GPU span of a block (first launch start -> last launch end): median {np.median(span):.1f} us = {np.median(span) / K:.2f} us per TTI"
      + (f"; {K} TTIs at the steady state's {steady:.2f} us would be {K * steady:.1f} us: {np.median(span) - K * steady:+.1f} us" if steady else ""))
first, later, gaps, lead = [], [], [], []
for b in blocks:
    byq = collections.defaultdict(list)
    for s, e, q in b:
        byq[q].append((s, e))
    ends = []
    for q, v in byq.items():
        v.sort()
        first.append((v[0][1] - v[0][0]) / 1e3)
        later += [(e - s) / 1e3 for s, e in v[1:]]
        gaps += [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
        ends.append(v[-1][1])
    lead.append((max(ends) - min(ends)) / 1e3)
if later:
    print(f"first launch of a queue in a block: median {np.median(first):.1f} us (p90 {np.percentile(first, 90):.1f}); later launches: median {np.median(later):.1f} us")
    print(f"gap between consecutive launches of a queue: median {np.median(gaps):.2f} us, sum per block {np.sum(gaps) / len(blocks):.1f} us over {len(gaps) // len(blocks)} boundaries")
else:
    print(f"one launch per queue and block (persistent launches, one per workgroup class): median {np.median(first):.1f} us (p90 {np.percentile(first, 90):.1f})")
print(f"the queue that finishes first is done {np.median(lead):.1f} us (median) before the last one")
idle = [min(s for s, _, _ in blocks[i + 1]) - max(e for _, e, _ in blocks[i]) for i in range(len(blocks) - 1)]
idle = [x / 1e3 for x in idle if x < 500_000]
if idle:
    print(f"GPU idle between two blocks (last kernel's end -> the host notices, syncs, times, calls ranenv_rollout again -> first kernel's start): median {np.median(idle):.1f} us"
          " -- the part of it inside the timed region (launch latency at the start, the sync's wake-up at the end) is paid once per block whatever K")
