"""Per-handle averages of the counters tools/placement_pmc.sh collected (residual sweeps only; handles identified by stream)."""
import collections
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select e.name, e.dispatch_id, e.counter_name, e.counter_value, e.duration, d.stream_id, d.queue_id "
                       "from pmc_events e join rocpd_kernel_dispatch d on d.dispatch_id = e.dispatch_id where e.name like '%k_residual%'").fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for name, did, cn, cv, du, sid, qid in rows:
        per[(sid, qid)][cn].append(cv)
        dur[(sid, qid)].append(du)
    print(path)
    for key in sorted(per):
        tail = {cn: v[-20:] for cn, v in per[key].items()}
        print("  stream/queue", key, "dur %.0f" % (sum(dur[key][-20:]) / len(dur[key][-20:]) / 1e3),
              " ".join("%s=%.4g" % (cn.replace("_sum", ""), sum(v) / len(v)) for cn, v in sorted(tail.items())))
