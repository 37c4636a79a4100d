"""Per-kernel means of the counters tools/pmc_sweep_anatomy.sh collected (rocpd databases): tools/pmc_sweep_report.py gpurun_out/anat_*/a_results.db"""
import collections
import os
import re
import sqlite3
import sys

for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, counter_name, counter_value, duration from pmc_events").fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, cn, cv, du in rows:
        m = re.search(r"sph::(k_\w+)(<[^>]*>)?", name)
        if m:
            per[m.group(1) + (m.group(2) or "")][cn].append(cv)
    print(path)
    for k in sorted(per, key=lambda k: -sum(len(v) for v in per[k].values())):
        if not re.match(os.environ.get("ANAT_KERNELS", r"k_(residual|correct|density|dfsph_ext|build_nl)"), k):
            continue
        # launches of a gated sweep that exit at once count almost nothing: keep launches above half of the maximum of the first counter
        first = sorted(per[k])[0]
        top = max(per[k][first]) or 1
        keep = [i for i, v in enumerate(per[k][first]) if v > 0.5 * top]
        print("  %-28s n=%-4d" % (k, len(keep)), " ".join("%s=%.4g" % (cn.replace("_sum", ""), sum(per[k][cn][i] for i in keep) / len(keep)) for cn in sorted(per[k])))
