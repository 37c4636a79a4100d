"""Worker of the loopback-transport tests: the library's NATIVE transport (ncclSend / ncclRecv / ncclAllReduce issued by the library itself on
its main, halo and reduction streams -- the discipline a multi-GPU node runs, with no host wait between the sweeps) on ONE GPU.

`world` slab handles live in this one process, one thread each; tests/loopback_rccl.hip stands in for librccl (SPH_RCCL_LIB, a development
override) and moves the bytes device-to-device under the same ordering rules.  The owned particles of all handles are compared bit for bit
with the same steps on a one-GPU handle.  With --time the steps are timed as well (all handles share the GPU: what this measures is the GPU
work a sharded step adds -- packing, ghosts' sweeps, the split launches -- not a scaling figure)."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shim_path(build=True):
    so = os.path.join(ROOT, "tests", "_build", "libloopback_rccl.so")
    src = os.path.join(ROOT, "tests", "loopback_rccl.hip")
    if build and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)):
        import subprocess
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so], check=True, cwd=ROOT)
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rebalance", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--arith", type=int, default=0)
    ap.add_argument("--time", type=int, default=0, help="time this many further steps after the compared ones (no comparison of those)")
    ap.add_argument("--no-compare", action="store_true")
    ap.add_argument("--toggle-overlap", type=int, default=0, help="switch the overlapped protocol off / on every this many steps (sph_slab_set_overlap), on every rank alike")
    ap.add_argument("--replay-rank", type=int, default=-1, help="record what this rank receives, then run it ALONE against that log and time it (needs --time)")
    ap.add_argument("--log-mb", type=int, default=8192)
    ap.add_argument("--save-log", default="", help="with --replay-rank: write the log to this file and stop (the replay runs in a process of its own: --load-log)")
    ap.add_argument("--load-log", default="", help="replay --replay-rank alone against a saved log (one thread, one handle: what a profiler should see)")
    ap.add_argument("--one-gpu", action="store_true", help="with a replay: time the same steps of the whole scene on a one-GPU handle in the same process")
    ap.add_argument("--overflow-rank", type=int, default=-1, help="this rank's handle gets --max-neighbors list rows: its lists overflow, and EVERY rank must fail the step")
    ap.add_argument("--max-neighbors", type=int, default=0)
    ap.add_argument("--param", action="append", default=[], help="name=value: a solver attribute (sph_set_scalar SPH_P_*) set on every slab handle AND on the one-GPU reference")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.environ["SPH_DEV"] = "1"
    os.environ["SPH_RCCL_LIB"] = shim_path()
    os.environ.setdefault("SPH_SLAB_CHECK", "1")
    from cfd_taichi_amd import _native as nat
    from cfd_taichi_amd import scenes
    cfg = json.load(open(args.scene)) if os.path.exists(args.scene) else scenes.get(args.scene)
    world = args.world
    rigid = None
    if cfg.get("solid"):
        from cfd_taichi_amd import mesh
        rigid = mesh.rigid_from_config(cfg)
    active = bool(rigid and rigid.get("active"))
    sims = [] if args.load_log else [nat.Simulation(nat.config_from_dict(cfg, slab_rank=r, slab_count=world, slab_rebalance_every=args.rebalance,
                                                                          slab_ghost_layers=args.layers, slab_overlap=args.overlap, arith=args.arith,
                                                                          max_neighbors=args.max_neighbors if r == args.overflow_rank else 0), rigid=rigid)
                                     for r in range(world)]
    params = [(kv.split("=")[0], float(kv.split("=")[1])) for kv in args.param]
    for sim_ in sims:
        for k_, v_ in params:
            sim_.set_param(k_, v_)
    wcsph = nat.config_from_dict(cfg).solver == nat.SOLVER_IDS["wcsph"]
    uid = nat.rccl_unique_id()
    shim = None
    if args.replay_rank >= 0:
        import ctypes
        shim = ctypes.CDLL(os.environ["SPH_RCCL_LIB"])
        shim.loopback_record_begin.argtypes = [ctypes.c_int, ctypes.c_size_t]
        shim.loopback_record_size.restype = ctypes.c_longlong
        shim.loopback_marker.argtypes = [ctypes.c_void_p]
        shim.loopback_log_save.argtypes = [ctypes.c_char_p]
        shim.loopback_log_load.argtypes = [ctypes.c_char_p]
        if args.load_log:
            rc = shim.loopback_log_load(args.load_log.encode())
            assert rc == 0, "loopback_log_load: %d" % rc
        else:
            assert shim.loopback_record_begin(args.replay_rank, args.log_mb << 20) == 0

    def state_digest(sim):
        import hashlib
        h = hashlib.sha1()
        for field in (nat.F_POS, nat.F_VEL, nat.F_RHO):
            ids, vals = sim.download_local(field)
            order = np.argsort(ids, kind="stable")          # by particle id (ghosts: ~id): the digest does not depend on the storage order
            h.update(ids[order].tobytes()); h.update(vals[order].tobytes())
        return h.hexdigest()
    stats = [[] for _ in range(world)]
    errors = [None] * world
    codes = [0] * world
    owned_max = [0] * world
    start = threading.Barrier(world)
    timing = {}

    def one_step(r):
        sim = sims[r]
        if wcsph:
            sim.step_wcsph(1)
            return None
        st = sim.step(1)
        if active:
            sim.rigid_step()
        return st

    def run(r):
        try:
            sims[r].rccl_attach(uid, 64 << 20)              # collective: returns when every rank has joined
            for k in range(args.steps):
                if args.toggle_overlap and k % args.toggle_overlap == 0:
                    sims[r].set_slab_overlap((k // args.toggle_overlap) % 2 == 1)
                st = one_step(r)
                owned_max[r] = max(owned_max[r], sims[r].slab_info()["owned"])
                if st is not None:
                    stats[r].append([st.n_div, st.n_dens, st.n_div_evals, float(st.div_first_err), float(st.div_err), float(st.dens_err), float(st.dt)])
            if args.time:
                sims[r].synchronize()
                start.wait()
                t0 = time.perf_counter()
                for _ in range(args.time):
                    one_step(r)
                sims[r].synchronize()
                start.wait()
                if r == 0:
                    timing["ms_per_step"] = (time.perf_counter() - t0) * 1e3 / args.time
        except BaseException as e:  # noqa: BLE001 - reported by the main thread; the other ranks run into the stand-in's bounded waits
            errors[r] = repr(e)
            codes[r] = getattr(e, "code", None)
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [] if args.load_log else [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if args.overflow_rank >= 0:        # every rank must have failed its step, with the overflow code, and none may hang (the caller's timeout)
        with open(args.out, "w") as f:
            json.dump({"codes": codes, "errors": errors, "steps_done": [len(s) for s in stats]}, f)
        for s in sims:
            s.close()
        return
    if any(errors):
        print("loopback worker failed:", errors, file=sys.stderr)
        sys.exit(1)
    infos = []
    for r in range(len(sims)):
        info = sims[r].slab_info()
        info["owned_max"] = owned_max[r]
        info["overrides"] = sims[r].overrides()
        infos.append(info)
    n = sims[0].n_fluid if sims else 0

    def gather(field):
        out = None
        seen = np.zeros(n, dtype=np.int32)
        for s in sims:
            ids, vals = s.download_owned(field)
            if out is None:
                out = np.full((n,) + vals.shape[1:], np.nan, dtype=np.float32)
            out[ids] = vals
            np.add.at(seen, ids, 1)
        assert np.all(seen == 1), "slab ownership is not a partition: %d missing, %d duplicated" % (int((seen == 0).sum()), int((seen > 1).sum()))
        return out

    result = {"world": world, "scene": args.scene, "steps": args.steps, "n": int(n), "slabs": infos, "lib_comm": sims[0].comm_stats() if sims else None,
              "stats_same_on_all_ranks": all(s == stats[0] for s in stats), "timing": timing or None,
              "relaxed": [s.scalar(nat.S_ARITH_RELAXED) for s in sims]}
    if args.save_log:
        rc = shim.loopback_log_save(args.save_log.encode())
        assert rc == 0, "loopback_log_save: %d" % rc
        result["recorded"] = {"rank": args.replay_rank, "digest": state_digest(sims[args.replay_rank]), "log": args.save_log}
        shim = None
    if not args.no_compare and not args.time:
        pos, vel, rho = gather(nat.F_POS), gather(nat.F_VEL), gather(nat.F_RHO)
        bodies = [{"scalars": s.rigid_scalars(), "pos": s.download(nat.F_RIGID_POS, nat.SPECIES_RIGID).tolist()} for s in sims] if rigid else None
        ref = nat.Simulation(nat.config_from_dict(cfg, arith=args.arith), rigid=rigid)
        for k_, v_ in params:
            ref.set_param(k_, v_)
        ref_stats = []
        for _ in range(args.steps):
            if wcsph:
                ref.step_wcsph(1)
            else:
                st = ref.step(1)
                ref_stats.append([st.n_div, st.n_dens, st.n_div_evals, float(st.div_first_err), float(st.div_err), float(st.dens_err), float(st.dt)])
            if active:
                ref.rigid_step()
        rp, rv, rr = ref.download(nat.F_POS), ref.download(nat.F_VEL), ref.download(nat.F_RHO)

        def rel(a, b):
            return float(np.abs(a.astype(np.float64) - b).max() / max(float(np.abs(b).max()), 1e-30))
        result.update({
            "pos_equal": bool(np.array_equal(pos, rp)), "vel_equal": bool(np.array_equal(vel, rv)), "rho_equal": bool(np.array_equal(rho, rr)),
            "pos_rel_err": rel(pos, rp), "vel_rel_err": rel(vel, rv),
            "stats_equal": stats[0] == ref_stats, "stats_last": stats[0][-1] if stats[0] else None, "ref_stats_last": ref_stats[-1] if ref_stats else None,
            "body_equal": None if bodies is None else bool(all(b == {"scalars": ref.rigid_scalars(), "pos": ref.download(nat.F_RIGID_POS, nat.SPECIES_RIGID).tolist()} for b in bodies)),
        })
        ref.close()
    if shim is not None:
        # rank k alone against the log of what it received: the same computation (checked below), nothing else on the GPU
        import ctypes
        k = args.replay_rank
        used = ctypes.c_longlong(0)
        entries = shim.loopback_record_size(ctypes.byref(used))
        assert entries > 0, "the log overflowed: raise --log-mb"
        rid = ctypes.create_string_buffer(128)
        shim.loopback_replay_id(rid)
        solo = nat.Simulation(nat.config_from_dict(cfg, slab_rank=k, slab_count=world, slab_rebalance_every=args.rebalance, slab_ghost_layers=args.layers,
                                                   slab_overlap=args.overlap, arith=args.arith), rigid=rigid)
        solo.rccl_attach(rid.raw, 64 << 20)
        sims.append(solo)
        world = len(sims) - 1
        for _ in range(args.steps):
            one_step(world)
        solo.synchronize()
        shim.loopback_marker(ctypes.c_void_p(solo.stream_ptr()))
        t0 = time.perf_counter()
        for _ in range(args.time):
            one_step(world)
        solo.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / max(args.time, 1)
        shim.loopback_marker(ctypes.c_void_p(solo.stream_ptr()))
        solo.synchronize()
        same = None
        if not args.load_log:
            same = state_digest(sims[k]) == state_digest(solo)
        result["replay"] = {"rank": k, "ms_per_step": dt, "log_entries": int(entries), "log_bytes": int(used.value), "same_state_as_in_the_full_run": same, "digest": state_digest(solo),
                            "owned": solo.slab_info()["owned"], "ghosts": solo.slab_info()["ghosts"]}
        if args.one_gpu:          # the yardstick, same process, same clocks: the whole scene on one handle, same steps
            for s_ in sims:
                s_.close()
            sims = []
            one = nat.Simulation(nat.config_from_dict(cfg, arith=args.arith), rigid=rigid)
            sims.append(one)
            for _ in range(args.steps):
                one_step(0)
            one.synchronize()
            shim.loopback_marker(ctypes.c_void_p(one.stream_ptr()))
            t0 = time.perf_counter()
            for _ in range(args.time):
                one_step(0)
            one.synchronize()
            result["one_gpu"] = {"ms_per_step": (time.perf_counter() - t0) * 1e3 / max(args.time, 1), "n": int(one.n_fluid)}
            shim.loopback_marker(ctypes.c_void_p(one.stream_ptr()))
            one.synchronize()
    with open(args.out, "w") as f:
        json.dump(result, f)
    for s in sims:
        s.close()


if __name__ == "__main__":
    main()
