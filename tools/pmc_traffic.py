#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md's
HBM section prescribes) into profiles/pmc_traffic.json: HBM bytes per launch for every kernel.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write N_particles [out.json] [gpurun_out/pmc_sq]

With a third pass directory (--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES) every kernel also gets its wave-level
instruction counts per launch: bench.py prices the dominant kernel against the VALU issue peak with them.

Units and gfx950 corrections (MI355X_MICROARCH.md section HBM): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the
read side is doubled.  The correction is calibrated in situ on dfsph_integrate / order_gather, whose
byte counts are known exactly (pure float4 streams), and the calibration is stored in the output.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"sph::(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0]


KERNEL_TO_PROFILE_NAME = {     # first template argument decides the bench.py name; the RIGID flag does not
    "k_residual<false": "dfsph_div_residual", "k_residual<true": "dfsph_dens_residual",
    "k_correct<0": "dfsph_warm_start", "k_correct<1": "dfsph_div_correct", "k_correct<2": "dfsph_dens_correct",
    "k_density<true": "dfsph_density_alpha", "k_density<false": "wcsph_density", "k_wcsph_force": "wcsph_force",
    "k_dfsph_ext": "dfsph_ext_force", "k_dfsph_integrate": "dfsph_integrate", "k_build_nl": "build_nl",
    "k_hash_count": "hash_count", "k_order_gather": "order_gather", "k_scatter": "scatter",
}


def profile_name(k):
    head = k.split(",")[0].rstrip(">")
    for key, name in KERNEL_TO_PROFILE_NAME.items():
        if head == key or head.split("<")[0] == key:
            return name
    return k


def collect(directory, counter):
    vals = defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return vals


def live(vals):
    """Launches of a gated sweep that exit at their first instruction (the loop had ended) count almost nothing: keep the real ones.
    (2 % of the largest launch, not 50 %: with change propagation in the density loop a launch in which most tiles return at once is
    a real launch and belongs in the mean.)"""
    if not vals:
        return vals
    top = max(vals)
    return [v for v in vals if v >= 0.02 * top] if top > 0 else vals


def main():
    fetch_dir, write_dir, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    # in-situ calibration on a kernel with exactly known traffic: dfsph_integrate reads 2 float4 and writes 2 float4 per particle
    calib = {}
    if "k_dfsph_integrate" in fetch:
        known = 32.0 * n
        f = sum(fetch["k_dfsph_integrate"]) / len(fetch["k_dfsph_integrate"]) * 1024
        w = sum(write["k_dfsph_integrate"]) / len(write["k_dfsph_integrate"]) * 1024
        calib = {"kernel": "k_dfsph_integrate", "known_read_bytes": known, "known_write_bytes": known,
                 "raw_fetch_bytes": f, "raw_write_bytes": w, "fetch_factor_measured": known / f if f else None,
                 "write_factor_measured": known / w if w else None}
    fetch_factor = 2.0   # MI355X_MICROARCH.md: FETCH_SIZE = 1/2 of wide coalesced reads on gfx950
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fl, wl = live(fetch.get(k, [])), live(write.get(k, []))
        fr = sum(fl) / max(len(fl), 1) * 1024
        wr = sum(wl) / max(len(wl), 1) * 1024
        kernels[profile_name(k)] = {
            "kernel": k, "launches_sampled": len(fetch.get(k, [])),
            "fetch_bytes_raw_per_launch": fr, "write_bytes_per_launch": wr,
            "hbm_bytes_per_launch": fr * fetch_factor + wr,
            "hbm_bytes_largest_launch": (max(fetch.get(k, [0])) * fetch_factor + max(write.get(k, [0]))) * 1024,
            "hbm_bytes_per_particle": (fr * fetch_factor + wr) / n,
        }
    if len(sys.argv) > 5:
        sq = {c: collect(sys.argv[5], c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_BUSY_CYCLES")}
        for k in sorted(sq["SQ_INSTS_VALU"]):
            if not k.startswith("k_"):
                continue
            entry = kernels.setdefault(profile_name(k), {"kernel": k})
            for c, vals in sq.items():
                if vals.get(k):
                    lv = live(vals[k])
                    entry[c.lower() + "_per_launch"] = sum(lv) / len(lv)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cfd_taichi_amd import build as hip_build
    # the sources these counters were taken on: bench.py compares with the sources it runs and labels roofline.traffic "stale" on a mismatch
    out = {"csrc_sha256": hip_build.sources_sha256(), "n_particles": n, "counter_unit": "KiB", "fetch_correction": fetch_factor, "calibration": calib, "kernels": kernels,
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (two separate passes)"}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=2)
        f.write("\n")
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main()
