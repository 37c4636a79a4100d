"""Kernel statistics of the REPLAY window of a tools/loopback_replay.sh trace: the launches between the two loopback_marker_kernel dispatches
(one rank of a sharded run alone on the GPU).  python3 tools/replay_trace.py <rocprofv3 output dir>"""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "loopback_marker_kernel" in r["Kernel_Name"]]
assert len(marks) >= 2, "markers not found (%d)" % len(marks)
# windows: (replay) or (replay, one-GPU yardstick); argv[2] = which pair from the front (default 0 = the replay)
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
a, b = marks[2 * which], marks[2 * which + 1]
win = rows[a + 1:b]
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
agg = defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in win:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void sph::", "").replace("sph::", "")
    if "k_residual" in name or "k_correct" in name:          # the split launches of a sweep differ in their grids
        name += " grid %s" % (int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) // 256)
    agg[name][0] += 1
    agg[name][1] += d
    busy += d
# union of the busy intervals (kernels of the three streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
union, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
print("window: %.1f us, %d launches, sum of kernel durations %.1f us, GPU busy (union) %.1f us = %.1f %% of the window" % (span, len(win), busy, union / 1e3, 100.0 * union / 1e3 / span))
print("%-44s %8s %12s %10s %7s" % ("kernel", "launches", "total us", "avg us", "share"))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %8d %12.1f %10.2f %6.2f%%" % (name[:44], n, t, t / n, 100.0 * t / busy))

# idle gaps (no kernel of any stream running), grouped by what ran before and after
gaps = defaultdict(lambda: [0, 0.0])
evs = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sph::", "").replace("sph::", "")[:28]) for r in win))
cur_end, cur_name = None, None
for s, e, name in evs:
    if cur_end is not None and s > cur_end:
        g = (s - cur_end) / 1e3
        if g > 2.0:
            gaps[(cur_name, name)][0] += 1
            gaps[(cur_name, name)][1] += g
    if cur_end is None or e > cur_end:
        cur_end, cur_name = e, name
tot = sum(v[1] for v in gaps.values())
print("\nidle gaps > 2 us: %.1f us in total (%.1f %% of the window)" % (tot, 100.0 * tot / span))
print("%-30s %-30s %7s %10s %8s" % ("after", "before", "count", "total us", "avg us"))
for (a_, b_), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-30s %-30s %7d %10.1f %8.2f" % (a_, b_, n, t, t / n))

# a slice of the timeline: every launch of ~600 us in the middle of the window, with its queue (= stream)
# (argv[3] = offset in us, or "@kernel:k:length_us" = from the k-th launch of that kernel in the window)
if len(sys.argv) > 3:
    length = 600000
    if sys.argv[3].startswith("@"):
        kname, kth, lus = sys.argv[3][1:].split(":")
        hits = [r for r in win if kname in r["Kernel_Name"]]
        t0 = int(hits[int(kth)]["Start_Timestamp"])
        length = int(float(lus) * 1e3)
    else:
        t0 = int(rows[a]["End_Timestamp"]) + int(float(sys.argv[3]) * 1e3)
    qs = {}
    print("\ntimeline slice from +%s us (columns: start, end, duration [us]; queue; kernel)" % sys.argv[3])
    for r in win:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s_ < t0 or s_ > t0 + length:
            continue
        q = r.get("Queue_Id", "?")
        qs.setdefault(q, len(qs))
        print("%9.1f %9.1f %7.1f  q%d  %s" % ((s_ - t0) / 1e3, (e_ - t0) / 1e3, (e_ - s_) / 1e3, qs[q], r["Kernel_Name"].split("(")[0].replace("void sph::", "").replace("sph::", "")[:40]))
