"""Timeline facts from a rocprofv3 --kernel-trace CSV: python tools/trace_overlap.py <kernel_trace.csv> [name filter]
For the step kernel: launches, mean duration, mean gap between consecutive launches on the same queue, and the
fraction of the span during which 0 / 1 / 2 / 3+ step kernels were running."""
import csv, sys, collections, re
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
flt = sys.argv[2] if len(sys.argv) > 2 else "ranenv_core_kernel"
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows if flt in r["Kernel_Name"] and re.search(r"<0[,>]", r["Kernel_Name"])]
ks.sort()
ks = ks[len(ks) // 5:]          # drop the warm-up fifth
dur = [e - s for s, e, _, _ in ks]
print(f"{len(ks)} launches, mean duration {sum(dur) / len(dur) / 1e3:.1f} us, queues {sorted(set(q for _, _, q, _ in ks))}")
byq = collections.defaultdict(list)
for s, e, q, _ in ks: byq[q].append((s, e))
for q, v in byq.items():
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print(f"  queue {q}: {len(v)} launches, mean gap to the next launch {sum(gaps) / max(1, len(gaps)) / 1e3:.1f} us")
ev = sorted([(s, 1) for s, e, _, _ in ks] + [(e, -1) for s, e, _, _ in ks])
t_prev, n, acc = ev[0][0], 0, collections.Counter()
for t, d in ev:
    acc[min(n, 3)] += t - t_prev; t_prev = t; n += d
span = ev[-1][0] - ev[0][0]
print("  concurrency:", {k: round(v / span, 3) for k, v in sorted(acc.items())}, f"span per launch {span / len(ks) / 1e3:.1f} us")
