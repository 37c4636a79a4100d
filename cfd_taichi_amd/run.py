"""Headless runner: the reference's frame loop (main.py:95-206) without the GGUI window.

    python -m cfd_taichi_amd.run --config config/dfsph_small.json [--solver dfsph] [--until 4.0 | --steps N] [--ply-dir output]
    torchrun --nproc-per-node 8 -m cfd_taichi_amd.run --config config/dfsph_10m.json --steps 100      # one x-slab per GPU (no rigid body)

Per frame: `iter_cnt` fluid steps, then `iter_cnt` rigid steps if a rigid body is active (main.py:165-171),
t += iter_cnt * solver.delta_time[None] (:173); stops at t > 4.0 (:205) or after --steps frames.  With --ply-dir (or
scene.is_output_ply) it writes ASCII PLY frames `output_%06d.ply` at scene.output_fps like main.py:189-195 (vertex
xyz + the constant RGBA of ParticleSystem.py:152) and, when a rigid body exists, `obj_%06d.obj` of its mesh (:196-200)."""
import argparse
import importlib
import os
import time

import numpy as np

from . import ParticleSystem, rigid_solver, utils


def write_ply_ascii(path, pos, rgba):
    """ASCII PLY in the layout of ti.tools.PLYWriter.export_frame_ascii: float x y z red green blue alpha."""
    n = len(pos)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment created by cfd_taichi_amd\nelement vertex %d\n" % n)
        for name in ("x", "y", "z", "red", "green", "blue", "alpha"):
            f.write("property float %s\n" % name)
        f.write("end_header\n")
        np.savetxt(f, np.hstack([pos, rgba]), fmt="%.6f")


def main_sharded(args, config):
    """WORLD_SIZE > 1 (launched by torchrun): one x-slab per rank, native RCCL transport when every rank has its own GPU (else
    torch.distributed callbacks); rank 0 gathers the positions for the PLY frames."""
    import torch
    import torch.distributed as dist
    from . import _native as nat
    from .slab import SlabSimulation
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if config.get("solid"):
        raise SystemExit("rigid bodies are not sharded: run this config on one GPU")
    own_gpu = torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", world))
    device = local if own_gpu else 0
    if own_gpu:
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo")
    transport = args.transport or ("native" if own_gpu else "torch")
    sim = SlabSimulation(config, rank, world, device=device, transport=transport, rebalance_every=50, arith=nat.arith_id(args.arith))
    scene_config, solver_config = config["scene"], config["solver"]
    iter_cnt = solver_config.get("iter_cnt")
    ply_dir = args.ply_dir or ("./output" if scene_config.get("is_output_ply", False) else None)
    if ply_dir and rank == 0:
        os.makedirs(ply_dir, exist_ok=True)
    rgba = np.tile(np.float32([0.0, 0.26, 0.68, 1.0]), (sim.n_fluid, 1))          # ParticleSystem.py:152
    frame_time = 1.0 / scene_config.get("output_fps", 60)
    frame_cnt, ply_cnt, t = 0, 0, 0.0
    start_time = time.time()
    while True:
        sim.step(iter_cnt)
        frame_cnt += 1
        t += iter_cnt * sim.sim.scalar(nat.S_DELTA_TIME)
        if ply_dir and (t / frame_time) > ply_cnt:
            pos = sim.gather(nat.F_POS)
            if rank == 0:
                write_ply_ascii(os.path.join(ply_dir, "output_%06d.ply" % ply_cnt), pos, rgba)
            ply_cnt += 1
        if (args.steps and frame_cnt >= args.steps) or t > args.until or frame_cnt > 100000:
            break
    if rank == 0:
        print("slabs: %d, frames: %d, simulated time: %.6f s, PLY frames: %d" % (world, frame_cnt, t, ply_cnt))
        print("Simulation time: {}".format(time.time() - start_time))
    sim.close()
    dist.barrier()
    dist.destroy_process_group()
    return frame_cnt, t, ply_cnt


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", default="./default.json")          # main.py:14
    ap.add_argument("--solver", default=None, help="override solver.name (wcsph | dfsph | pcisph | iisph | pbf)")
    ap.add_argument("--until", type=float, default=4.0, help="stop when simulated time exceeds this (main.py:205)")
    ap.add_argument("--steps", type=int, default=0, help="stop after this many frames (0 = use --until)")
    ap.add_argument("--ply-dir", default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--transport", default=None, choices=["native", "torch"], help="sharded runs: the library's own RCCL communicator (default with one GPU per rank) or torch.distributed callbacks")
    ap.add_argument("--arith", default=None, choices=["exact", "relaxed"],
                    help="exact (default): the reference's f32 operations in its order; relaxed: the tolerance-grade sweeps (SphConfig.arith)")
    args = ap.parse_args(argv)

    config = utils.read_config(args.config)
    if args.solver:
        config["solver"]["name"] = args.solver
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return main_sharded(args, config)
    scene_config, solver_config = config["scene"], config["solver"]
    print("Simulation Start!")
    start_time = time.time()
    ps = ParticleSystem(config, device=args.device, arith=args.arith)
    name = solver_config.get("name")
    if name not in ("wcsph", "dfsph", "pcisph", "iisph", "pbf"):
        raise SystemExit("solver '%s' is not covered; pass --solver wcsph | dfsph | pcisph | iisph | pbf" % name)
    module = importlib.import_module("cfd_taichi_amd." + name + "_solver")      # main.py:65-68
    solver = getattr(module, name + "_solver")(ps, config)
    rs = rigid_solver(ps, config) if config.get("solid", {}) else None           # :69-71
    iter_cnt = solver_config.get("iter_cnt")
    ply_dir = args.ply_dir or ("./output" if scene_config.get("is_output_ply", False) else None)
    if ply_dir:
        os.makedirs(ply_dir, exist_ok=True)
    np_rgba = ps.rgba.to_numpy()
    frame_time = 1.0 / scene_config.get("output_fps", 60)
    frame_cnt, ply_cnt, t = 0, 0, 0.0
    while True:
        if frame_cnt > 100000:                                                   # :98
            break
        for _ in range(iter_cnt):
            solver.step()
        for _ in range(iter_cnt):
            if rs and ps.active_rigid[None] == 1:
                rs.step()
        frame_cnt += 1
        t += iter_cnt * solver.delta_time[None]
        if ply_dir and (t / frame_time) > ply_cnt:                               # :189
            write_ply_ascii(os.path.join(ply_dir, "output_%06d.ply" % ply_cnt), ps.fluid_particles.pos.to_numpy(), np_rgba)
            if ps.exist_rigid[None] == 1:                                        # :196-200
                ps.update_mesh_vextics()
                with open(os.path.join(ply_dir, "obj_%06d.obj" % ply_cnt), "w") as f:
                    f.write(ps.mesh.export(file_type="obj"))
            ply_cnt += 1
        if (args.steps and frame_cnt >= args.steps) or t > args.until:           # :205
            break
    print("frames: %d, simulated time: %.6f s, PLY frames: %d" % (frame_cnt, t, ply_cnt))
    print("Simulation time: {}".format(time.time() - start_time))               # :211
    return frame_cnt, t, ply_cnt


if __name__ == "__main__":
    main()
