"""Inter-kernel gaps from a rocprofv3 --kernel-trace CSV: how much of the GPU timeline of the timed steps is kernels, how much is the
space between them.   tools/trace_gaps.py <dir with *_kernel_trace.csv> [first_fraction]"""
import csv, glob, re, sys, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]            # the later part of the run: steady state
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
per = collections.defaultdict(list)
for (s, e, n), g in zip(rows, gaps):
    m = re.search(r"sph::(k_\w+)(<[^>]*>)?", n)
    per[(m.group(1) + (m.group(2) or "")) if m else n[:40]].append((e - s, g))
print("kernels %d, span %.3f ms, busy %.3f ms (%.1f %%), mean gap %.2f us, median gap %.2f us" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span,
      sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
for k, v in sorted(per.items(), key=lambda kv: -sum(d for d, _ in kv[1]))[:14]:
    print("  %-36s n=%-5d dur %.1f us  gap after %.2f us" % (k, len(v), sum(d for d, _ in v) / len(v) / 1e3, sum(g for _, g in v) / len(v) / 1e3))
