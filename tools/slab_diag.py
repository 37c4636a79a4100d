"""Finds the first step at which a sharded run differs from the single-GPU run (rank 0 steps both in lockstep):
    torchrun --nproc-per-node 3 tools/slab_diag.py --scene dfsph_dam_x --steps 1400 --rebalance 7 --every 1 --from-step 300"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--scene"); ap.add_argument("--steps", type=int); ap.add_argument("--rebalance", type=int, default=0); ap.add_argument("--every", type=int, default=1); ap.add_argument("--from-step", type=int, default=0, dest="start")
a = ap.parse_args()
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
from cfd_taichi_amd import _native as nat, scenes
from cfd_taichi_amd.slab import SlabSimulation
cfg = scenes.get(a.scene)
sim = SlabSimulation(cfg, rank, world, device=0, rebalance_every=a.rebalance)
ref = nat.Simulation(nat.config_from_dict(cfg)) if rank == 0 else None
for s in range(a.steps):
    st = sim.step(1)
    if rank == 0:
        rst = ref.step(1)
        pass
    if s >= a.start and (s % a.every == 0):
        pos = sim.gather(nat.F_POS)
        info = sim.sim.slab_info(); infos = [None]*world if rank == 0 else None
        dist.gather_object((info["owned"], info["ghosts"], info["x_lo"], info["x_hi"]), infos, dst=0)
        stop = [0]
        if rank == 0:
            rp = ref.download(nat.F_POS)
            bad = np.argwhere((pos != rp).any(axis=1)).ravel()
            if len(bad):
                h = 0.1
                print("first mismatch at step", s, "ids", bad[:10], "slabs", infos)
                for b in bad[:6]:
                    print(" id", b, "slab pos", pos[b], "ref pos", rp[b], "cell", np.floor(rp[b]/h).astype(int))
                lost = np.argwhere(((rp < 0) | (np.floor(rp/h) >= np.array(ref.grid))).any(axis=1)).ravel()
                print(" lost ids in ref:", lost[:10], [ (rp[l], np.floor(rp[l]/h).astype(int)) for l in lost[:4]])
                stop[0] = 1
        dist.broadcast_object_list(stop, src=0)
        if stop[0]: break
dist.barrier(); dist.destroy_process_group()
