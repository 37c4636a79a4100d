"""Worker of the multi-process slab tests (launched by torch.distributed.run): runs a sharded simulation on
`world` ranks (all on GPU 0 with the gloo transport when only one GPU is present), gathers the owned particles
on rank 0 and compares them bit-for-bit with the same steps on a single-GPU handle."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", required=True, help="scene name from cfd_taichi_amd.scenes, or a path to a config JSON")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--rebalance", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0, help="SphConfig.slab_ghost_layers (0 = the solver's default)")
    ap.add_argument("--overlap", type=int, default=0, help="SphConfig.slab_overlap (0 = default on, 1 = off)")
    ap.add_argument("--arith", type=int, default=0, help="SphConfig.arith on the slabs AND on the one-GPU reference")
    ap.add_argument("--old-comm", action="store_true", help="hand the transport over as a caller built against the round-3 SphComm would: without exchange_counts_n / reduce_capacity (sph_set_comm_sized)")
    ap.add_argument("--host-loops", action="store_true", help="a transport without allreduce_stream: the library runs the dfsph loops on the host")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    device = local % max(ndev, 1)
    if args.backend == "nccl":
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo")
    from cfd_taichi_amd import _native as nat
    from cfd_taichi_amd import scenes
    from cfd_taichi_amd.slab import SlabSimulation
    cfg = json.load(open(args.scene)) if os.path.exists(args.scene) else scenes.get(args.scene)
    sim = SlabSimulation(cfg, rank, world, device=device, rebalance_every=args.rebalance, slab_ghost_layers=args.layers, slab_overlap=args.overlap,
                         arith=args.arith, host_loops=args.host_loops,
                         comm_struct_size=nat.SphComm.exchange_counts_n.offset if args.old_comm else None)
    dfsph = sim.solver != "wcsph"      # every solver but wcsph reports per-step statistics
    stats = []
    owned_max = 0
    for _ in range(args.steps):
        st = sim.step(1)
        owned_max = max(owned_max, sim.sim.slab_info()["owned"])
        if dfsph:
            stats.append([st.n_div, st.n_dens, st.n_div_evals, float(st.div_first_err), float(st.div_err), float(st.dens_err), float(st.dt)])
    info = sim.sim.slab_info()
    info["owned_max"] = owned_max
    body = None
    if cfg.get("solid"):          # every rank holds the whole body: its state must be the same everywhere, and equal to the one-GPU run's
        body = {"scalars": sim.sim.rigid_scalars(), "pos": sim.sim.download(nat.F_RIGID_POS, nat.SPECIES_RIGID).tolist()}
        bodies = [None] * world if rank == 0 else None
        dist.gather_object(body, bodies, dst=0)
    infos = [None] * world if rank == 0 else None
    dist.gather_object(info, infos, dst=0)
    pos = sim.gather(nat.F_POS)
    vel = sim.gather(nat.F_VEL)
    rho = sim.gather(nat.F_RHO)
    result = None
    if rank == 0:
        if os.environ.get("SLAB_REF_NOSKIP") == "1":      # the one-GPU reference computes every tile in every density iteration
            os.environ["SPH_TILE_SKIP"] = "0"
        rigid = None
        if cfg.get("solid"):
            from cfd_taichi_amd import mesh
            rigid = mesh.rigid_from_config(cfg)
        ref = nat.Simulation(nat.config_from_dict(cfg, device=device, arith=args.arith), rigid=rigid)
        ref_stats = []
        for _ in range(args.steps):
            if dfsph:
                st = ref.step(1)
                ref_stats.append([st.n_div, st.n_dens, st.n_div_evals, float(st.div_first_err), float(st.div_err), float(st.dens_err), float(st.dt)])
            else:
                ref.step_wcsph(1)
            if rigid and rigid.get("active"):
                ref.rigid_step()
        rp, rv, rr = ref.download(nat.F_POS), ref.download(nat.F_VEL), ref.download(nat.F_RHO)

        def rel(a, b):
            return float(np.abs(a.astype(np.float64) - b).max() / max(float(np.abs(b).max()), 1e-30))
        result = {
            "world": world, "scene": args.scene, "steps": args.steps, "n": int(sim.n_fluid), "slabs": infos,
            "pos_equal": bool(np.array_equal(pos, rp)), "vel_equal": bool(np.array_equal(vel, rv)), "rho_equal": bool(np.array_equal(rho, rr)),
            "pos_rel_err": rel(pos, rp), "vel_rel_err": rel(vel, rv),
            "stats_equal": stats == ref_stats, "stats_last": stats[-1] if stats else None, "ref_stats_last": ref_stats[-1] if ref_stats else None,
            "body_equal": None if body is None else bool(all(b == {"scalars": ref.rigid_scalars(), "pos": ref.download(nat.F_RIGID_POS, nat.SPECIES_RIGID).tolist()} for b in bodies)),
            "body_centroid": None if body is None else body["scalars"]["centroid"], "body_omega": None if body is None else body["scalars"]["omega"],
            "comm": sim.comm.stats, "lib_comm": sim.sim.comm_stats(), "relaxed": [sim.sim.scalar(nat.S_ARITH_RELAXED), ref.scalar(nat.S_ARITH_RELAXED)],
        }
        with open(args.out, "w") as f:
            json.dump(result, f)
        ref.close()
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
