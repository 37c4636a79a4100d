#!/usr/bin/env python3
"""bench.py -- million particle-steps/s of the DFSPH dam break on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one dfsph_solver.step() (grid rebuild, density/alpha, divergence solve, external
forces, adaptive dt, density solve, integration) over the whole particle set, which is resident in
HBM before the timed region starts.  Rank 0 prints ONE JSON line.

The timed window: `--preroll` untimed steps first (default 50, reported as config.preroll_steps) so that whatever --warmup / --steps
the caller passes land in the collapsing phase SURVEY.md 8d prescribes (50 warm-up + 200 timed), not in the first steps from rest
where the density loop runs its minimum of 2 iterations; the pre-roll itself is timed and reported as `early_phase`.

Extra objects on the line:
  roofline      dominant kernel: algorithmic bytes per launch / its mean launch
                duration, measured live with HIP events on the library's stream (a separate,
                profiled replay of the warm-up + timed steps); `traffic` and `valu.wave_insts_per_launch` are NOT measured in
                this run: they are read from the rocprofv3 --pmc passes committed under profiles/ (`traffic_source`), null if absent.
  cpu_baseline  the CPU oracle ("port": restatement of the ti.cpu path, not Taichi) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1 only): for dfsph / wcsph scenes it continues
                from the device's state at the start of the timed window (same phase of the collapse as `value`).
  relaxed       N = 1, dfsph scenes: the same timed steps from the same state with SphConfig.arith = SPH_ARITH_RELAXED (tolerance-grade sweeps),
                its throughput, dominant kernel and measured deviation from the exact run; the headline `value` is the exact arithmetic.
  config.slab_protocol_probe   N > 1, dfsph: the last 2 x 6 steps of the pre-roll time the two slab protocols -- halo and reductions on their own streams
                under the sweeps, or everything in order on one stream (the same bits) -- and the run keeps the faster on all ranks: with a free link
                the in-order form wins, on a slow one the overlap; --no-overlap-probe keeps the overlapped one unmeasured.
  strong_scaling_base   N = 1, default workload only: the N > 1 workload (dfsph_10m) on this one GPU with the same flags, so that
                the strong-scaling series the driver assembles from N = 2, 4, 8 has its one-GPU point.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec

# algorithmic bytes per particle per launch
# pcisph / iisph sweeps, same accounting (own state read + own results written, neighbour data counted once per particle):
#   pcisph_predict_rho: PP 16 + PB in/out 32 + rho_predict 4 = 52; pcisph_press_force: PB 16 + V 16 + EF 16 + rho 4 + PF 16 + PP 16 = 84
#   iisph_d_ij: PB 16 + rho 4 + DIJ 16 = 36; iisph_update_p: PB in/out 32 + DII 16 + DIJ 16 + rho/rho_adv/a_ii 12 = 76
ALGO_BYTES = {
    "pcisph_ext_force": 96, "pcisph_predict_rho": 52, "pcisph_press_force": 84, "pcisph_integrate": 96,
    "iisph_advect": 64, "iisph_rho_adv": 76, "iisph_d_ij": 36, "iisph_update_p": 76, "iisph_integrate": 116,
    "dfsph_density_alpha": 24, "dfsph_warm_start": 48, "dfsph_div_residual": 32, "dfsph_div_correct": 56,
    "dfsph_ext_force": 40, "dfsph_dens_residual": 32, "dfsph_dens_correct": 48, "dfsph_integrate": 48,
    "wcsph_density": 16, "wcsph_force": 52, "hash_count": 16, "order_gather": 64, "build_nl": 0,
    "pbf_lambda": 36, "pbf_delta_pos": 96, "pbf_xsph": 80,      # (pos 16 -> rho 4, lambda 4, (pos, lambda) 16) / (2 x 16 in, 3 x 16 out + lists) / (3 x 16 in, 2 x 16 out)
}


# kernels that skip tiles whose inputs did not change (change propagation in the dfsph density loop; zero-pressure tiles of pcisph / iisph)
TILE_SKIPPING = ("dfsph_dens_residual", "dfsph_dens_correct", "pcisph_press_force", "iisph_d_ij")

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / device-buffer sharing across processes needs dmabuf IPC on this driver


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, help="scene name from cfd_taichi_amd.scenes (default dfsph_1m)")
    ap.add_argument("--profile-steps", type=int, default=1, help="0 = skip the HIP-event profiled replay (roofline leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll", type=int, default=int(os.environ.get("SPH_BENCH_PREROLL", "50")),
                    help="untimed steps before the warm-up (reported as config.preroll_steps and timed as early_phase)")
    ap.add_argument("--no-scaling-base", action="store_true", help="N=1: skip the dfsph_10m one-GPU point")
    ap.add_argument("--no-relaxed", action="store_true", help="N=1 dfsph: skip the tolerance-grade arithmetic leg (the `relaxed` object)")
    ap.add_argument("--allow-overrides", action="store_true", default=os.environ.get("SPH_BENCH_ALLOW_OVERRIDES") == "1",
                    help="print a line although development overrides (SPH_DEV=1 + SPH_* knobs) are in force; they are named in config.overrides")
    ap.add_argument("--no-overlap-probe", action="store_true", help="N > 1, dfsph: keep the overlapped slab protocol without timing it against the in-order one")
    ap.add_argument("--rebalance", type=int, default=int(os.environ.get("SPH_REBALANCE_EVERY", "50")),
                    help="N>1: re-cut the x-slabs from the current particle distribution every M steps (0 = static cuts)")
    ap.add_argument("--first-contact", action="store_true",
                    help="(internal) N>1: run ONLY the transport probe and the discipline self-check and report each stage on stdout -- the short child job "
                         "first_contact_probe() starts under a wall-clock limit before the measuring ranks touch RCCL")
    return ap.parse_args()


def host_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota (16 on a 1-GPU box)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


CPU_BASELINE_SECONDS = 20.0     # bounded sample of the same workload on the host cores


def cpu_baseline(scene_name, solver_kind, state=None, first_step=1):
    """Oracle (kind 'port') on the host cores, bounded sample of the same workload.  `state` = (pos, vel, warm_start_k, dt) of the
    device at the start of the timed window (dfsph / wcsph / pbf without a body: pos, vel and for dfsph warm_start_k, dt are the whole
    state a step reads): the oracle continues from there, so the sample is the
    same phase of the collapse `value` is measured in; otherwise whole steps from rest."""
    from cfd_taichi_amd import scenes
    from oracle import oracle as orc
    cores = host_cores()
    cfg = scenes.get(scene_name)
    rigid = None
    if cfg.get("solid"):
        from cfd_taichi_amd import mesh
        rigid = mesh.rigid_from_config(cfg)
    o = orc.Oracle(cfg, num_threads=cores, rigid=rigid)
    rigid_active = bool(rigid and rigid.get("active"))
    if solver_kind == "pcisph":
        step = lambda: o.step_pcisph(1)
    elif solver_kind == "iisph":
        step = lambda: o.step_iisph(1)
    elif solver_kind == "dfsph":
        step = lambda: o.step_dfsph(1, 100)
    elif solver_kind == "pbf":
        step = lambda: o.step_pbf(1)
    else:
        step = lambda: o.step_wcsph(1)

    def one():
        step()
        if rigid_active:
            o.rigid_step()                # main.py:169-171
    if state is not None:
        pos, vel, warm, dt = state
        o.set(orc.F_POS, pos); o.set(orc.F_VEL, vel)
        if warm is not None:
            o.set(orc.F_WARM_K, warm); o.set_dt(dt)
        start = first_step
    else:
        one()        # one untimed step (from rest the first step is atypical: zero divergence residual)
        start = 2
    timed, t0 = 0, time.perf_counter()
    while True:
        one()
        timed += 1
        dt_s = time.perf_counter() - t0
        if dt_s >= CPU_BASELINE_SECONDS or timed >= 2000:
            break
    if solver_kind == "dfsph":
        detail = ", n_div=%d, n_dens=%d in the last step" % (o.last_stats.n_div, o.last_stats.n_dens)
    elif solver_kind in ("pcisph", "iisph"):
        detail = ", %d pressure iterations in the last step" % o.last_stats.n_dens
    else:
        detail = ""
    how = "continuing from the device state at the start of the timed window" if state is not None else "after 1 untimed step from rest"
    sample = "steps %d-%d of %s (N=%d%s) %s" % (start, start + timed - 1, scene_name, o.N, detail, how)
    value = o.N * timed / dt_s / 1e6
    o.close()
    return {"value": value, "unit": "Mparticle-steps/s", "cores": cores, "kind": "port",
            "sample": sample + "; OpenMP restatement of the ti.cpu path (oracle/), not Taichi", "seconds": dt_s}


def relaxed_leg(nat, scenes, scene_name, device, args, state, final_exact, exact_stats, fence, solver_kind="dfsph"):
    """SphConfig.arith = SPH_ARITH_RELAXED on the same workload: a second handle receives the state the exact handle had at the start of
    its timed window (pos, vel, warm_start_k, delta_time), runs the same --steps steps between two fences, and is compared with where
    the exact handle ended: per-particle deviation quantiles (relative to max|x| resp. max|v|), iteration counts next to each other.
    The headline `value` stays the exact arithmetic; this is the measured price list of the 1e-5 tolerance north_star states."""
    import numpy as np
    cfg = scenes.get(scene_name)
    sim = nat.Simulation(nat.config_from_dict(cfg, device=device, arith=nat.ARITH_RELAXED))
    pos, vel, warm, dt = state

    wcsph = solver_kind == "wcsph"

    def restore():
        sim.upload(nat.F_POS, pos); sim.upload(nat.F_VEL, vel)
        if not wcsph:
            sim.upload(nat.F_WARM_K, warm); sim.set_dt(dt)

    restore()
    if wcsph:
        sim.step_wcsph(4)                     # untimed: list build, graph capture, clocks
    else:
        sim.step_dfsph(1)
    active = sim.scalar(nat.S_ARITH_RELAXED) == 1.0
    restore()
    builds0 = sim.scalar(nat.S_VERLET_BUILDS) if wcsph else 0
    fence(sim)
    t0 = time.perf_counter()
    stats = []
    if wcsph:
        sim.step_wcsph(args.steps)
    else:
        for _ in range(args.steps):
            st = sim.step_dfsph(1)
            stats.append((st.n_div, st.n_dens))          # (the stats object is reused by the binding: copy the numbers)
    fence(sim)
    elapsed = time.perf_counter() - t0
    n = sim.n_fluid

    def quant(a, b):
        e = np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(1)) / max(float(np.abs(b).max()), 1e-30)
        return {"q50": float(np.quantile(e, 0.5)), "q99": float(np.quantile(e, 0.99)), "max": float(e.max())}

    out = {"arith": "SPH_ARITH_RELAXED (csrc/sph_relaxed_kernels.h: v_rsq_f32, FMAs, grad W as one scalar, per-step wall sums)", "active": active,
           "value": n * args.steps / elapsed / 1e6, "unit": "Mparticle-steps/s", "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps}
    if wcsph:
        out["verlet_list_builds_in_the_timed_steps"] = sim.scalar(nat.S_VERLET_BUILDS) - builds0
        out["arith"] += "; Verlet lists (skin 0.05 h, rebuilt when a particle has moved skin / 2)"
    else:
        out.update({"n_dens_mean": sum(x[1] for x in stats) / len(stats), "n_div_mean": sum(x[0] for x in stats) / len(stats),
                    "exact_n_dens_mean": sum(x[1] for x in exact_stats) / len(exact_stats)})
    out.update({
           "deviation_from_exact_after_the_timed_steps": {"pos": quant(sim.download(nat.F_POS), final_exact[0]), "vel": quant(sim.download(nat.F_VEL), final_exact[1]),
                                                          "norm": "per-particle |a - b| / max|b|; same start state (the exact handle's at the start of its timed window)"},
           "envelope": ("wcsph: the reference's own envelope is 4e-8 after 200 steps; tests/test_relaxed_gpu.py holds this path to 1e-5 of the oracle directly" if wcsph else
                        "two legal executions of the REFERENCE differ by more than this after as many steps (profiles/r03/envelope_*.json, tools/envelope.py)")})
    if args.profile_steps > 0:
        restore()
        sim.profile_enable(True)
        if wcsph:
            sim.step_wcsph(args.steps)
        else:
            for _ in range(args.steps):
                sim.step_dfsph(1)
        sim.synchronize()
        prof = sim.profile()
        sim.profile_enable(False)
        tot = sum(ms for ms, _ in prof.values())
        sweeps = {k: v for k, v in prof.items() if ALGO_BYTES.get(k, 0) > 0 and k not in TILE_SKIPPING}
        dom = max(sweeps, key=lambda k: sweeps[k][0])
        ms, cnt = prof[dom]
        avg_s = ms / cnt / 1e3
        rx_names = {"dfsph_div_residual": "k_residual_rx<false>", "dfsph_dens_residual": "k_residual_rx<true>", "dfsph_warm_start": "k_correct_rx<0>",
                    "dfsph_div_correct": "k_correct_rx<1>", "dfsph_dens_correct": "k_correct_rx<2>", "wcsph_density": "k_wcsph_density_rx", "wcsph_force": "k_wcsph_force_rx"}
        traffic, traffic_status = (load_traffic(rx_names.get(dom), path=os.path.join(ROOT, "profiles", "pmc_traffic_relaxed.json"))
                                   if scene_name == "dfsph_1m" else (None, None))
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ALGO_BYTES[dom] * n / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ALGO_BYTES[dom] * n / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_status": traffic_status,
                           "traffic_source": "profiles/pmc_traffic_relaxed.json (committed rocprofv3 --pmc passes with SPH_ARITH=relaxed; NOT measured in this run)" if traffic else None,
                           "avg_launch_us": avg_s * 1e6, "launches": cnt, "share_of_gpu_time": ms / tot if tot else None}
        out["kernel_breakdown_us"] = {k: {"avg_us": v[0] / v[1] * 1e3, "launches_per_step": v[1] / args.steps, "share": v[0] / tot}
                                      for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    sim.close()
    return out


def committed_status(data):
    """"current" when a committed measurement (profiles/pmc_traffic.json, valu_mix.json) was taken on the kernel sources this run executes -- the
    writers store cfd_taichi_amd.build.sources_sha256() -- "stale" when the sources have changed since, "unknown" for a file without a digest."""
    from cfd_taichi_amd import build as hip_build
    sha = data.get("csrc_sha256") if isinstance(data, dict) else None
    if not sha:
        return "unknown"
    return "current" if sha == hip_build.sources_sha256() else "stale"


def load_traffic(kernel, key="hbm_bytes_per_launch", path=None):
    """(HBM bytes per launch from the committed --pmc passes, status) -- (None, None) if absent.  ONE file: profiles/pmc_traffic.json."""
    path = path or os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            data = json.load(f)
        v = data.get("kernels", {}).get(kernel, {}).get(key)
        return v, (committed_status(data) if v is not None else None)
    except Exception:
        return None, None


def load_valu_mix(kernel_profile_name):
    """Per-launch VALU class counts and their issue floor for the dominant kernels (tools/valu_mix.py; None if absent)."""
    # staged, non-rigid instantiations (the third template argument is the sweep mode: 1 = staged; "true" in profiles taken before it became one)
    names = {"dfsph_div_residual": "k_residual<false, false, %s>", "dfsph_dens_residual": "k_residual<true, false, %s>",
             "dfsph_div_correct": "k_correct<1, false, %s>", "dfsph_dens_correct": "k_correct<2, false, %s>"}
    try:
        with open(os.path.join(ROOT, "profiles", "valu_mix.json")) as f:
            data = json.load(f)
        kernels = data["kernels"]
        pattern = names.get(kernel_profile_name)
        if pattern is None:
            return None
        # (the fourth template argument, RX = the relaxed kernel functions on unstaged handles, arrived in round 4: profiles taken since carry it)
        mix = kernels.get((pattern % "1")[:-1] + ", false>") or kernels.get(pattern % "1") or kernels.get(pattern % "true")
        if mix is not None:
            mix = dict(mix, status=committed_status(data))
        return mix
    except Exception:
        return None


def load_ceiling():
    """G wave64-instructions/s that tools/valu_issue.hip sustained on an MI355X (committed under profiles/; None if absent)."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_issue.json")) as f:
            return json.load(f).get("valu_issue_ginst_measured")
    except Exception:
        return None


# VALU issue on gfx950, measured (tools/valu_issue.hip -> profiles/valu_issue.json): a SIMD takes a plain wave64 f32 instruction every ~2.3
# cycles once two waves alternate (~1050 G wave-instructions/s on the chip, 135 TFLOP/s as FMAs), a packed v_pk_fma_f32 every ~4.2
# (146 TFLOP/s: the guide's 157.3 TFLOP/s FP32 vector peak), an instruction with an SGPR operand ~4, a transcendental 8.  roofline.valu
# prices the dominant kernel's own instruction mix at those costs.
FP32_VECTOR_PEAK_TFLOPS = 157.3


def make_sim(nat, scenes, scene_name, world, rank, local_rank, transport_group, rebalance=0, discipline=None):
    """Single-GPU handle, or this rank's slab of the sharded simulation."""
    cfg = scenes.get(scene_name)
    if world == 1:
        rigid = None
        if cfg.get("solid"):
            from cfd_taichi_amd import mesh
            rigid = mesh.rigid_from_config(cfg)
        return nat.Simulation(nat.config_from_dict(cfg, device=local_rank), rigid=rigid), None
    from cfd_taichi_amd.slab import SlabSimulation, TorchComm
    slab = SlabSimulation.__new__(SlabSimulation)
    c = nat.config_from_dict(cfg, device=local_rank, slab_rank=rank, slab_count=world, slab_rebalance_every=rebalance)
    slab.rank, slab.world = rank, world
    slab.solver = cfg["solver"]["name"]
    slab.sim = nat.Simulation(c)
    if discipline == "native":          # the library's own RCCL communicator; the id travels over the gloo side group
        from cfd_taichi_amd.slab import attach_native
        slab.comm = None
        attach_native(slab.sim, rank, 128 << 20, transport_group)
    else:
        slab.comm = TorchComm(rank, world, device=local_rank, capacity_bytes=128 << 20, group=transport_group, stream_ptr=slab.sim.stream_ptr(),
                              stream_ordered=None if discipline is None else discipline == "stream")
        slab.sim.set_comm(slab.comm.struct)
    slab.n_fluid = slab.sim.n_fluid
    return slab.sim, slab


def pick_transport(dist, torch, rank, world, local_rank, gloo):
    """RCCL (nccl backend) device-to-device halo exchange when it works; otherwise the same protocol over a gloo
    side group with host staging.  Every rank takes the same decision (the probe result is all-reduced over `gloo`)."""
    from cfd_taichi_amd.slab import TorchComm
    if os.environ.get("SPH_TRANSPORT", "nccl") == "gloo":
        return gloo, "gloo (host staged, forced by SPH_TRANSPORT)"
    ok = 1
    try:
        probe = TorchComm(rank, world, device=local_rank, capacity_bytes=1 << 16)
        rl, rr = probe.exchange_counts(3, 5)
        probe.bufs["send_left"][:64] = rank
        probe.bufs["send_right"][:64] = rank
        probe.exchange_buffers(64 if rank > 0 else 0, 64 if rank < world - 1 else 0, 64 if rank > 0 else 0, 64 if rank < world - 1 else 0)
        torch.cuda.synchronize()
        if rank > 0 and (rl != 5 or int(probe.bufs["recv_left"][0]) != rank - 1):
            ok = 0
        if rank < world - 1 and (rr != 3 or int(probe.bufs["recv_right"][0]) != rank + 1):
            ok = 0
        if probe.allreduce([1.0], 0) != [float(world)]:
            ok = 0
    except Exception as e:  # noqa: BLE001
        print("[bench] rank %d: RCCL halo probe failed (%s); falling back to gloo host staging" % (rank, e), file=sys.stderr)
        ok = 0
    t = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=gloo)
    if int(t.item()) == 1:
        return None, "rccl p2p (device buffers over xGMI)"
    return gloo, "gloo (host staged: RCCL probe failed)"


def choose_discipline(nat, scenes, scene_name, world, rank, local_rank, dist, torch, rebalance, gloo, note=None):
    """Which RCCL discipline drives the halo: "native" (the library issues ncclSend / ncclRecv / ncclAllReduce itself on its stream),
    "stream" (torch.distributed calls ordered on the library's stream) or "sync" (host waits around every transfer).  The faster ones
    are only used if two steps of the workload give the very bytes the synchronous discipline gives (SHA-1 of every rank's owned
    particles).  Agreement -- and any exception on any rank -- is all-reduced over the gloo side group: all ranks decide alike."""
    import hashlib
    import numpy as np

    def agreed(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=gloo)
        return bool(int(t.item()))

    def digest(mode):
        ok, sim, out = True, None, None
        try:
            sim, slab = make_sim(nat, scenes, scene_name, world, rank, local_rank, gloo if mode == "native" else None, rebalance, discipline=mode)
            if mode != "native":
                ok = slab.comm.stream_ordered == (mode == "stream")
            if ok:
                for _ in range(2):
                    (sim.step_dfsph if slab.solver == "dfsph" else sim.step_wcsph)(1)
                hsh = hashlib.sha1()
                for f in (nat.F_POS, nat.F_VEL):
                    ids, vals = sim.download_owned(f)
                    hsh.update(np.ascontiguousarray(vals[np.argsort(ids, kind="stable")]).tobytes())
                out = hsh.hexdigest()
        except Exception as e:  # noqa: BLE001
            print("[bench] rank %d: the %s discipline failed in the self-check: %s" % (rank, mode, e), file=sys.stderr)
            ok = False
        if sim is not None:
            sim.close()
        return out if agreed(ok) else None

    ref = digest("sync")
    if ref is None:
        return "sync", "the synchronous discipline itself failed in the self-check on some rank"
    if note:
        note("sync_ok")
    wanted = [m for m in os.environ.get("SPH_SLAB_DISCIPLINES", "native,stream").split(",") if m in ("native", "stream")]
    for mode in wanted:
        d = digest(mode)
        if agreed(d is not None and d == ref):
            return mode, "2 steps reproduce the synchronous discipline byte for byte"
    return "sync", "no faster discipline reproduced it"


# ---- first contact with real RCCL, bounded ---------------------------------------------------------------------------
# `bench.py --gpus N` is a chain of code that has never met librccl with more than one rank: the nccl process group, the halo probe, three
# handle generations of the discipline self-check.  Every step has a sound fallback for an EXCEPTION; none has one for a HANG, and a hang in a
# measuring rank ends as the driver's timeout with no line at all.  So the chain first runs in a short-lived CHILD job of fresh processes
# (`bench.py --first-contact`: same ranks, same GPUs, the small dfsph_1m scene) under a wall-clock limit; the child reports every stage it
# completes on stdout, and what it proved -- not what it hoped -- is handed to the measuring ranks through the environment
# (FIRST_CONTACT_ENV).  They then only execute what a fresh process has already survived:
#   child finished              -> its verdict (transport rccl | gloo, discipline native | stream | sync)
#   stuck after "sync_ok"       -> rccl transport, synchronous discipline (the faster ones hung or never answered)
#   stuck after "transport"     -> whatever transport the probe chose; over rccl nothing beyond the probe is proven: gloo host staging
#   stuck before that           -> the nccl process group itself is suspect: gloo backend, gloo host staging
# The child is a process GROUP of its own and is killed as one on expiry; a rank that has touched the GPU is never re-executed.
FIRST_CONTACT_ENV = "SPH_BENCH_FIRST_CONTACT"
FIRST_CONTACT_TAG = "FIRST_CONTACT "
# seconds a stage may take after the one before it was reported (the first import of torch on a fresh box pages the image in: up to 2 minutes)
FIRST_CONTACT_LIMITS = {"start": 60.0, "import": 240.0, "init": 90.0, "transport": 60.0, "sync_ok": 90.0, "discipline": 90.0}
FIRST_CONTACT_TOTAL = 420.0
FIRST_CONTACT_ORDER = ("start", "import", "init", "transport", "sync_ok", "discipline")


def first_contact_verdict(stages, finished, why):
    """What the measuring ranks may rely on, from the stages the child reported ({stage: payload})."""
    v = {"backend": "gloo", "transport": "gloo", "discipline": None, "reached": [k for k in FIRST_CONTACT_ORDER if k in stages], "how": why}
    if "discipline" in stages:             # everything was reported (a child that then fails to EXIT -- e.g. in destroy_process_group -- proved no less)
        v.update(backend="nccl", transport=stages["transport"].get("transport", "gloo"), discipline=stages["discipline"].get("discipline"))
    elif "sync_ok" in stages and stages.get("transport", {}).get("transport") == "rccl":
        v.update(backend="nccl", transport="rccl", discipline="sync")
    elif stages.get("transport", {}).get("transport") == "gloo":
        v.update(backend="nccl", transport="gloo")
    return v


def first_contact_probe(nproc, rebalance=0, cmd=None, limits=None, total=None):
    """Start the child job, follow its stage lines under the limits, kill its process group on expiry.  Nothing here imports torch or touches
    the GPU.  `cmd` replaces the child's command line (tests/test_first_contact.py hangs a stub at every stage)."""
    import queue
    import signal
    import socket
    import subprocess
    import threading
    limits = dict(FIRST_CONTACT_LIMITS, **(limits or {}))
    total = FIRST_CONTACT_TOTAL if total is None else total
    if cmd is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__), "--first-contact", "--gpus", str(nproc), "--rebalance", str(rebalance)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                                                                 "MASTER_PORT", FIRST_CONTACT_ENV) and not k.startswith("TORCHELASTIC_")}
    t0 = time.monotonic()
    print("[bench] first contact with RCCL in a child job (limit %.0f s): %s" % (total, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, env=env, start_new_session=True, text=True, bufsize=1)
    lines = queue.Queue()

    def pump():
        for line in child.stdout:
            lines.put(line)
        lines.put(None)

    threading.Thread(target=pump, daemon=True).start()
    stages, nxt, finished, why = {}, 0, False, None
    stage_deadline = t0 + limits[FIRST_CONTACT_ORDER[0]]
    while True:
        now = time.monotonic()
        wait = min(stage_deadline, t0 + total) - now
        if wait <= 0:
            why = "no '%s' within %.0f s (child killed after %.0f s)" % (FIRST_CONTACT_ORDER[min(nxt, len(FIRST_CONTACT_ORDER) - 1)],
                                                                         limits[FIRST_CONTACT_ORDER[min(nxt, len(FIRST_CONTACT_ORDER) - 1)]], now - t0)
            break
        try:
            line = lines.get(timeout=min(wait, 1.0))
        except queue.Empty:
            continue
        if line is None:                        # stdout closed: the child is ending
            try:
                rc = child.wait(timeout=max(1.0, min(30.0, t0 + total - time.monotonic())))
            except subprocess.TimeoutExpired:
                rc = None
            finished = rc == 0 and "discipline" in stages
            why = "child finished in %.0f s" % (time.monotonic() - t0) if finished else "child ended with code %s after '%s'" % (rc, FIRST_CONTACT_ORDER[nxt - 1] if nxt else "nothing")
            break
        if not line.startswith(FIRST_CONTACT_TAG):
            continue
        try:
            msg = json.loads(line[len(FIRST_CONTACT_TAG):])
        except ValueError:
            continue
        st = msg.get("stage")
        if st in FIRST_CONTACT_ORDER:
            stages[st] = msg
            nxt = max(nxt, FIRST_CONTACT_ORDER.index(st) + 1)
            if nxt < len(FIRST_CONTACT_ORDER):
                stage_deadline = time.monotonic() + limits[FIRST_CONTACT_ORDER[nxt]]
            else:
                stage_deadline = time.monotonic() + 30.0          # everything reported: the job only has to exit
    if child.poll() is None:                    # expiry, or a child that will not end: the whole process group goes
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            child.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass
    v = first_contact_verdict(stages, finished, why)
    v["seconds"] = round(time.monotonic() - t0, 1)
    print("[bench] first contact: %s" % json.dumps(v), file=sys.stderr, flush=True)
    return v


def first_contact_for_rank(local_rank, world, rebalance):
    """The ranks were started by somebody else's launcher (the driver's torch.distributed.run): local rank 0 runs the probe before anything in
    this process has touched torch or the GPU, the others wait for its verdict (one node: a file keyed by the launcher's pid)."""
    import tempfile
    started = time.time()
    path = os.path.join(tempfile.gettempdir(), "sph_first_contact_%d_%s.json" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
    if local_rank == 0:
        v = first_contact_probe(world, rebalance)
        v["written_at"] = time.time()
        tmp = path + ".%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump(v, f)
        os.replace(tmp, path)
        return v
    deadline = time.monotonic() + FIRST_CONTACT_TOTAL + 90.0
    while time.monotonic() < deadline:
        try:
            with open(path) as f:
                v = json.load(f)
            if v.get("written_at", 0) >= started - 5.0:      # (not a leftover of an earlier job with the same pid and port)
                return v
        except (OSError, ValueError):
            pass
        time.sleep(0.2)
    raise SystemExit("[bench] local rank %d: no first-contact verdict from local rank 0 within %.0f s" % (local_rank, FIRST_CONTACT_TOTAL + 90.0))


def first_contact_child(args, rank, world, local_rank):
    """`bench.py --first-contact` under torch.distributed.run: the chain the measuring ranks would otherwise meet for the first time, on the small
    scene, every completed stage reported by rank 0 as one tagged JSON line on stdout."""
    def note(stage, **kw):
        if rank == 0:
            print(FIRST_CONTACT_TAG + json.dumps(dict(stage=stage, **kw)), flush=True)

    note("start")
    import torch
    import torch.distributed as dist
    note("import")
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    side = dist.new_group(backend="gloo")
    dist.barrier(group=side)
    note("init")
    from cfd_taichi_amd import _native as nat
    from cfd_taichi_amd import scenes
    transport_group, transport = pick_transport(dist, torch, rank, world, local_rank, side)
    note("transport", transport="rccl" if transport_group is None else "gloo", detail=transport)
    discipline, why = None, "host-staged transport: no discipline to choose"
    if transport_group is None:
        if os.environ.get("SPH_SLAB_SYNC", "0") == "1":
            note("sync_ok")
            discipline, why = "sync", "forced by SPH_SLAB_SYNC=1"
        else:
            try:
                discipline, why = choose_discipline(nat, scenes, "dfsph_1m", world, rank, local_rank, dist, torch, args.rebalance, side, note=note)
            except Exception as e:  # noqa: BLE001
                discipline, why = "sync", "the self-check raised %s" % type(e).__name__
    else:
        note("sync_ok")
    dist.barrier(group=side)
    note("discipline", discipline=discipline, why=why)
    dist.destroy_process_group()
    return 0


def self_launch(nproc):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) as a CHILD
    process -- nothing in this process has touched torch or HIP yet, and it never execs -- pass its output through and return its
    exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = None
    if FIRST_CONTACT_ENV not in os.environ and os.environ.get("SPH_BENCH_REHEARSAL") != "1" and "--first-contact" not in sys.argv:
        rebalance = sys.argv[sys.argv.index("--rebalance") + 1] if "--rebalance" in sys.argv[:-1] else os.environ.get("SPH_REBALANCE_EVERY", "50")
        env = dict(os.environ)
        env[FIRST_CONTACT_ENV] = json.dumps(first_contact_probe(nproc, rebalance))
    print("[bench] no launcher in the environment: starting %s" % " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world
    if args.first_contact:
        raise SystemExit(first_contact_child(args, rank, world, local_rank) if world > 1 else "[bench] --first-contact needs N > 1 ranks")
    rehearsal = os.environ.get("SPH_BENCH_REHEARSAL") == "1"   # all ranks on GPU 0 over gloo: lets a 1-GPU box exercise this file
    # N > 1 on real GPUs: what a short-lived child job has proven about RCCL on this node (first_contact_probe), from whoever launched us or --
    # under somebody else's launcher -- from local rank 0, before anything in this process touches torch or the GPU
    contact = None
    if world > 1 and not rehearsal and os.environ.get("SPH_BENCH_NO_FIRST_CONTACT") != "1":
        contact = json.loads(os.environ[FIRST_CONTACT_ENV]) if os.environ.get(FIRST_CONTACT_ENV) else first_contact_for_rank(local_rank, world, args.rebalance)

    import torch
    dist = None
    transport_group, transport = None, None
    host_backend = rehearsal or (contact is not None and contact["backend"] == "gloo")      # control tensors live on the host
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            local_rank = 0
            dist.init_process_group(backend="gloo")
            side = dist.new_group(backend="gloo")
            transport_group, transport = None, "gloo (rehearsal on one GPU)"
        elif contact is not None:
            torch.cuda.set_device(local_rank)
            if contact["backend"] == "gloo":
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            side = dist.new_group(backend="gloo")
            if contact["transport"] == "rccl":
                transport_group, transport = None, "rccl p2p (device buffers over xGMI)"
            else:
                transport_group, transport = side, "gloo (host staged)"
            transport += "; chosen by a first-contact child job under a time limit (%s; stages reached: %s)" % (contact["how"], ", ".join(contact["reached"]) or "none")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            side = dist.new_group(backend="gloo")        # control decisions (probe results) travel over gloo
            transport_group, transport = pick_transport(dist, torch, rank, world, local_rank, side)

    from cfd_taichi_amd import _native as nat
    from cfd_taichi_amd import scenes

    # N = 1: config 3 of BASELINE.json (dfsph_1m).  N > 1: config 4 (dfsph_10m), x-slabs with ghost exchange: the total
    # work is fixed over N = 2, 4, 8 (strong scaling); 1.25 M particles per GPU at N = 8.
    scene_name = args.workload or ("dfsph_1m" if world == 1 else "dfsph_10m")
    cfg = scenes.get(scene_name)
    solver_kind = cfg["solver"]["name"]
    discipline = None
    if world > 1 and transport_group is None and (not rehearsal or os.environ.get("SPH_BENCH_VERIFY") == "1"):      # RCCL device-to-device transport
        if os.environ.get("SPH_SLAB_SYNC", "0") == "1":
            discipline, why = "sync", "forced by SPH_SLAB_SYNC=1"
        elif contact is not None:          # the measuring ranks never execute a discipline a fresh process has not survived
            discipline, why = contact["discipline"] or "sync", "first-contact child job: %s" % contact["how"]
        else:
            try:
                discipline, why = choose_discipline(nat, scenes, scene_name, world, rank, local_rank, dist, torch, args.rebalance, side)
            except Exception as e:  # noqa: BLE001 - never lose the run over the self-check
                discipline, why = "sync", "the self-check raised %s" % type(e).__name__
        transport += "; discipline: %s (%s)" % ({"native": "native RCCL calls on the library's stream", "stream": "torch.distributed ordered on the library's stream",
                                                  "sync": "synchronous"}[discipline], why)
    group_for = (side if discipline == "native" else transport_group) if world > 1 else None
    sim, slab = make_sim(nat, scenes, scene_name, world, rank, local_rank, group_for, args.rebalance, discipline)
    n_total = sim.n_fluid
    # development overrides (SPH_DEV=1 + SPH_* knobs, SPH_LIB) change what is measured: no line without naming them
    overrides = sim.overrides()
    if overrides and not args.allow_overrides:
        raise SystemExit("[bench] development overrides are in force (%s): refusing to print a line; unset them or pass --allow-overrides "
                         "(they are then named in config.overrides)" % ", ".join(overrides))
    if world == 1:
        sim.build_neighbors()      # (which sweeps run is settled by the first list build: k / rho array, 16-bit lists)
    headline_arith = "relaxed" if sim.scalar(nat.S_ARITH_RELAXED) == 1.0 else "exact"

    has_rigid = bool(cfg.get("solid")) and world == 1
    rigid_active = has_rigid and bool(cfg["solid"].get("active", False))     # main.py:169-171: rs.step() only if ps.active_rigid[None] == 1

    def make_run(sim, kind, with_body):
        def run(nsteps, stats=None):
            if kind == "dfsph":
                for _ in range(nsteps):
                    st = sim.step_dfsph(1)
                    if with_body:
                        sim.rigid_step()
                    if stats is not None:
                        stats.append((st.n_div, st.n_dens, st.n_div_evals))
            elif kind in ("pcisph", "iisph"):
                step = sim.step_pcisph if kind == "pcisph" else sim.step_iisph
                for _ in range(nsteps):
                    st = step(1)
                    if with_body:
                        sim.rigid_step()
                    if stats is not None:
                        stats.append((0, st.n_dens, 0))
            elif kind == "pbf":
                sim.step_pbf(nsteps)
            elif with_body:
                for _ in range(nsteps):
                    sim.step_wcsph(1)
                    sim.rigid_step()
            else:
                sim.step_wcsph(nsteps)
        return run

    def fence(sim):
        sim.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def timed_window(sim, run, want_state=False):
        """pre-roll (timed on the side as the early phase), warm-up, then EXACTLY args.steps steps between two fences; max over ranks"""
        fence(sim)
        t0 = time.perf_counter()
        probe = 0
        if world > 1 and solver_kind == "dfsph" and args.preroll >= 24 and not args.no_overlap_probe:
            started = sim.slab_info()["halo_overlapped"]
            try:                                   # can this handle run both protocols?  (created with the streams of the overlapped one)
                sim.set_slab_overlap(True)
                probe = 6 if sim.slab_info()["halo_overlapped"] else 0
            except nat.SphError:
                probe = 0
            sim.set_slab_overlap(started)
        run(args.preroll - 2 * probe)
        if probe:
            # Which protocol is faster HERE?  The dfsph loops with the halo and the reductions on their own streams, or in order on one stream: the
            # same bits either way, and the answer depends on the link (with the link time at zero the in-order form wins by 12 % at 1.2 M particles
            # per rank).  The last 2 x 6 steps of the pre-roll time both; every rank takes the same decision (MAX over ranks).
            times = {}
            for mode in (True, False):
                sim.set_slab_overlap(mode)
                fence(sim)
                tp = time.perf_counter()
                run(probe)
                sim.synchronize()
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device="cpu" if host_backend else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                times[mode] = float(tt[0].item()) / probe * 1e3
            keep = times[True] <= times[False]
            sim.set_slab_overlap(keep)
            overlap_probe.update({"overlapped_ms_per_step": times[True], "in_order_ms_per_step": times[False], "kept": "overlapped" if keep else "in order",
                                  "steps": "%d-%d" % (args.preroll - 2 * probe + 1, args.preroll)})
        fence(sim)
        early = time.perf_counter() - t0
        run(args.warmup)
        state = None
        if want_state:       # the device state the CPU baseline continues from (host copies, outside the timed region)
            warm = sim.download(nat.F_WARM_K) if solver_kind == "dfsph" else None
            state = (sim.download(nat.F_POS), sim.download(nat.F_VEL), warm, sim.scalar(nat.S_DELTA_TIME))
        fence(sim)
        stats = []
        if world > 1:
            sim.comm_stats(reset=True)               # transport requests of the timed steps only (config.rank0_comm)
        t0 = time.perf_counter()
        run(args.steps, stats)
        sim.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed, early], dtype=torch.float64, device="cpu" if host_backend else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, early = float(t[0].item()), float(t[1].item())
            dist.barrier()
        return elapsed, early, stats, state

    overlap_probe = {}
    run = make_run(sim, solver_kind, rigid_active)
    want_relaxed = rank == 0 and world == 1 and solver_kind in ("dfsph", "wcsph") and not has_rigid and not args.no_relaxed and headline_arith == "exact"
    want_state = rank == 0 and world == 1 and not has_rigid and solver_kind in ("dfsph", "wcsph", "pbf") and (not args.no_cpu_baseline or want_relaxed)
    elapsed, early, stats, state = timed_window(sim, run, want_state)
    final_exact = (sim.download(nat.F_POS), sim.download(nat.F_VEL)) if want_relaxed else None

    value = n_total * args.steps / elapsed / 1e6
    slab_info = sim.slab_info() if world > 1 else None
    comm_stats_timed = sim.comm_stats() if world > 1 else None

    out = {
        "metric": "million particle-steps/sec (%s dam-break)" % solver_kind.upper(),
        "value": value, "unit": "Mparticle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        # one series: N = 2, 4, 8 shard a FIXED workload (dfsph_10m); the N = 1 line carries that workload's one-GPU point as strong_scaling_base
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": scene_name, "solver": solver_kind, "particles": n_total, "wall_particles": sim.n_wall,
                   "grid": list(sim.grid), "preroll_steps": args.preroll, "arith": headline_arith, "overrides": overrides,
                   "timed_steps": "%d-%d" % (args.preroll + args.warmup + 1, args.preroll + args.warmup + args.steps),
                   "parallelism": "1 GPU" if world == 1 else "%d x-slabs, %s, halo transport: %s, cuts re-balanced every %d steps" % (
                       world, ("2 ghost cell columns per side (one halo refresh per dfsph solver iteration, %s)" % ("edge tiles of the residual sweeps first, halo and reductions on their own streams"
                                                                                                   if (slab_info or {}).get("halo_overlapped") else "in order on one stream")) if solver_kind == "dfsph"
                       else "1 ghost cell column per side", transport, args.rebalance)},
    }
    if args.preroll > 0:
        out["early_phase"] = {"steps": "1-%d" % args.preroll, "value": n_total * args.preroll / early / 1e6, "ms_per_step": early / args.preroll * 1e3,
                              "note": "the pre-roll from rest (density loop at its minimum of 2 iterations for most of it); not the headline"}
    if overlap_probe:
        out["config"]["slab_protocol_probe"] = overlap_probe
    if slab_info is not None:
        out["config"]["rank0_slab"] = slab_info
        cs = comm_stats_timed if comm_stats_timed is not None else sim.comm_stats()
        steps_c = max(1, cs["steps"])
        out["config"]["rank0_comm"] = {"per_step": {k: cs[k] / steps_c for k in cs if k != "steps"}, "steps": cs["steps"],
                                       "transport": (slab.comm.stats if slab.comm is not None else "native RCCL (the library's own communicator, no callbacks)"),
                                       "note": "requests of rank 0 during the timed steps: p2p_groups = send / recv pairs with the slab neighbours, bytes one way, "
                                               "count_exchanges = host round trips; read against the cost model in DESIGN.md section 6"}
    if stats:
        nd = [s[0] for s in stats]; ns = [s[1] for s in stats]; ne = [s[2] for s in stats]
        out["config"].update({"n_div_mean": sum(nd) / len(nd), "n_dens_mean": sum(ns) / len(ns), "n_div_evals_mean": sum(ne) / len(ne)})
        if solver_kind == "dfsph":
            algo_step = 272 + 88 * (sum(nd) / len(nd)) + 80 * (sum(ns) / len(ns))
            out["config"]["algorithmic_bytes_per_particle_step"] = algo_step
            out["step_hbm_frac_algorithmic"] = algo_step * n_total * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world)
        else:
            out["config"].pop("n_div_mean"); out["config"].pop("n_div_evals_mean")
            out["config"]["pressure_iterations_mean"] = out["config"].pop("n_dens_mean")

    # ---- roofline leg: HIP-event timing of every kernel on a fresh handle that replays the same steps; the events cover the
    # warm-up + timed steps (the pre-roll runs unprofiled), a rocprofv3 trace of this command additionally sees both pre-rolls ----
    if args.profile_steps > 0:
        sim.close()
        sim, slab = make_sim(nat, scenes, scene_name, world, rank, local_rank, group_for, args.rebalance, discipline)
        run = make_run(sim, solver_kind, rigid_active)
        run(args.preroll)
        sim.synchronize()
        sim.profile_enable(True)
        nprof = args.warmup + args.steps
        run(nprof)
        sim.synchronize()
        prof = sim.profile()
        sim.profile_enable(False)
        n_local = sim.slab_info()["owned"] + sim.slab_info()["ghosts"] if world > 1 else n_total
        tot = sum(ms for ms, _ in prof.values())
        # kernels with change propagation return at once on the tiles whose inputs did not change: a launch does a
        # fraction of the algorithmic work, so bytes x N / time is not a roofline fraction for them -- they are not priced and cannot be
        # the dominant kernel of the roofline object (their times stay in kernel_breakdown_us)
        sweeps = {k: v for k, v in prof.items() if ALGO_BYTES.get(k, 0) > 0 and k not in TILE_SKIPPING}
        dom = max(sweeps, key=lambda k: sweeps[k][0])
        ms, n = prof[dom]
        avg_s = ms / n / 1e3
        algo = ALGO_BYTES[dom] * n_local
        achieved = algo / avg_s / 1e9
        committed = world == 1 and scene_name == "dfsph_1m"       # the committed PMC passes were taken on this workload
        traffic, traffic_status = load_traffic(dom) if committed else (None, None)
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           # "current": the passes were taken on the kernel sources this run executes (sha256 of csrc/ + flags stored with them); "stale": not
                           "traffic_status": traffic_status,
                           "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes of this workload; NOT measured in this run)" if traffic else None,
                           "algorithmic_bytes_per_launch": algo, "avg_launch_us": avg_s * 1e6, "launches": n,
                           "share_of_gpu_time": ms / tot if tot else None, "rank": 0, "particles_on_rank": n_local,
                           "binding_limit": None,
                           "not_priced": [k for k in prof if k in TILE_SKIPPING],
                           "note": "priced on the HBM axis as north_star asks (algorithmic bytes / launch time / 8 TB/s); kernels in not_priced skip the tiles "
                                   "whose inputs did not change, so a launch does a fraction of the algorithmic work: they are never the dominant kernel here"}
        mix = load_valu_mix(dom) if committed else None
        if mix:
            # stated only where an instruction mix of THIS kernel on THIS workload is committed, and computed from this run's launch time
            out["roofline"]["binding_limit"] = "valu issue: this run's launches take %.2fx the issue floor of the kernel's own instruction mix (see valu)" % (
                avg_s * 1e6 / mix["issue_floor_us"])
            out["roofline"]["note"] += ("; the kernel's binding limit is f32 instruction issue, not HBM: %.1f M wave-instructions per launch (the reference's "
                                        "correctly rounded sqrt and divides per pair) for %.0f MB of algorithmic bytes -- see roofline.valu and DESIGN.md section 4"
                                        % (mix["wave_insts_per_launch"] / 1e6, algo / 1e6))
            # second opinion on the same kernel: its VALU instruction mix (committed SQ_INSTS_VALU_* pass) priced with the per-class issue
            # costs tools/valu_issue.hip measured on this chip -- the time the SIMDs need just to ISSUE the kernel's instructions
            g_inst = mix["wave_insts_per_launch"] / avg_s / 1e9
            out["roofline"]["valu"] = {"wave_insts_per_launch": mix["wave_insts_per_launch"], "plain": mix["plain"], "transcendental": mix["trans"], "other": mix["other"],
                                       "status": mix.get("status"),
                                       "source": "profiles/valu_mix.json (committed SQ_INSTS_VALU_* passes priced with profiles/valu_issue.json; NOT measured in this run)",
                                       "issue_floor_us": mix["issue_floor_us"], "frac_of_issue_floor": mix["issue_floor_us"] / (avg_s * 1e6),
                                       "achieved": g_inst, "unit": "G wave64-inst/s", "measured_issue_ceiling_plain_fma": load_ceiling(),
                                       "note": "gfx950 issues a plain VGPR-operand f32/int instruction in ~2.3-2.7 cycles per SIMD (two waves alternating), one with an SGPR "
                                               "operand, a packed v_pk_* or a VCC producer/consumer in ~4, a transcendental in 8 (tools/valu_issue.hip, profiles/valu_issue.json); "
                                               "the floor is the kernel's own mix at those costs on 1024 SIMDs at 2.4 GHz"}
        out["kernel_breakdown_us"] = {k: {"avg_us": v[0] / v[1] * 1e3, "launches_per_step": v[1] / nprof,
                                          "share": v[0] / tot} for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    if has_rigid:
        out["config"]["rigid_particles"] = sim.n_rigid
        out["config"]["rigid_active"] = rigid_active
    sim.close()
    # ---- N = 1, default workload: the one-GPU point of the strong-scaling series (the N > 1 workload on this GPU, same flags) ----
    if world == 1 and args.workload is None and not args.no_scaling_base:
        base_name = "dfsph_10m"
        bsim, _ = make_sim(nat, scenes, base_name, 1, 0, local_rank, None)
        b_elapsed, b_early, b_stats, _ = timed_window(bsim, make_run(bsim, "dfsph", False))
        out["strong_scaling_base"] = {"workload": base_name, "n_gpus": 1, "particles": bsim.n_fluid, "value": bsim.n_fluid * args.steps / b_elapsed / 1e6,
                                      "unit": "Mparticle-steps/s", "ms_per_step": b_elapsed / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                                      "preroll_steps": args.preroll, "n_dens_mean": sum(x[1] for x in b_stats) / len(b_stats),
                                      "note": "bench.py --gpus N (N > 1) runs this workload sharded into N x-slabs (scaling: strong); this is its N = 1 point"}
        bsim.close()
    if want_relaxed:
        out["relaxed"] = relaxed_leg(nat, scenes, scene_name, local_rank, args, state, final_exact, stats, fence, solver_kind)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene_name, solver_kind, state, args.preroll + args.warmup + 1)
        out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
