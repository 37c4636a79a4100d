"""CPU suite for the N>1 path: the torch.distributed transport behind the slab decomposition (world_size 2 and 3,
gloo, CPU tensors) and the host-only slab planner.  The device side is covered by tests/test_slab_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import scenes


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cfd_taichi_amd.slab import TorchComm
    comm = TorchComm(rank, world, capacity_bytes=1 << 16)
    assert comm.on_host
    ok = True
    # counts: every rank announces (10*rank+1 to the left, 10*rank+2 to the right)
    rl, rr = comm.exchange_counts(10 * rank + 1, 10 * rank + 2)
    ok &= rl == (10 * (rank - 1) + 2 if rank > 0 else 0)
    ok &= rr == (10 * (rank + 1) + 1 if rank < world - 1 else 0)
    # buffers: payload sizes differ per direction and per rank; content identifies the sender
    sl = 100 + rank if rank > 0 else 0
    sr = 200 + rank if rank < world - 1 else 0
    exp_rl = 200 + (rank - 1) if rank > 0 else 0
    exp_rr = 100 + (rank + 1) if rank < world - 1 else 0
    comm.bufs["send_left"][:max(sl, 1)] = 2 * rank
    comm.bufs["send_right"][:max(sr, 1)] = 2 * rank + 1
    comm.exchange_buffers(sl, sr, exp_rl, exp_rr)
    if rank > 0:
        ok &= bool((comm.bufs["recv_left"][:exp_rl] == 2 * (rank - 1) + 1).all())
    if rank < world - 1:
        ok &= bool((comm.bufs["recv_right"][:exp_rr] == 2 * (rank + 1)).all())
    # all-reduce: sum and max, through the same code path the C callback uses
    s = comm.allreduce([float(rank + 1), 0.5], 0)
    m = comm.allreduce([float(rank), -float(rank)], 1)
    ok &= s == [world * (world + 1) / 2, 0.5 * world] and m == [float(world - 1), 0.0]
    # the ctypes callback wrappers (what libsph_mi355x calls)
    import ctypes
    a, b = ctypes.c_int32(), ctypes.c_int32()
    ok &= comm._exchange_counts(None, 7, 9, ctypes.pointer(a), ctypes.pointer(b)) == 0
    ok &= a.value == (9 if rank > 0 else 0) and b.value == (7 if rank < world - 1 else 0)
    # the merged particle exchange's count message: n ints per neighbour in one round trip
    nl, nr = comm.exchange_counts_n([rank, 10 + rank, 20 + rank], [100 + rank, 110 + rank, 120 + rank])
    ok &= nl == ([100 + rank - 1, 110 + rank - 1, 120 + rank - 1] if rank > 0 else [0, 0, 0])
    ok &= nr == ([rank + 1, 10 + rank + 1, 20 + rank + 1] if rank < world - 1 else [0, 0, 0])
    sl5, sr5 = (ctypes.c_int32 * 5)(*[rank + k for k in range(5)]), (ctypes.c_int32 * 5)(*[50 + rank + k for k in range(5)])
    rl5, rr5 = (ctypes.c_int32 * 5)(), (ctypes.c_int32 * 5)()
    ok &= comm._exchange_counts_n(None, 5, sl5, sr5, rl5, rr5) == 0
    ok &= list(rl5) == ([50 + rank - 1 + k for k in range(5)] if rank > 0 else [0] * 5)
    ok &= list(rr5) == ([rank + 1 + k for k in range(5)] if rank < world - 1 else [0] * 5)
    vals = (ctypes.c_double * 2)(1.0, float(rank))
    ok &= comm._allreduce(None, vals, 2, 0) == 0 and vals[0] == float(world)
    # the in-place reduce buffer of the device-side loop control (host transport: the library stages reduce_buf itself)
    assert not comm.stream_ordered and comm.struct.stream_ordered == 0 and comm.struct.reduce_buf == comm.reduce_t.data_ptr()
    comm.reduce_t[:2] = torch.tensor([1.5 * (rank + 1), float(rank + 2)], dtype=torch.float64)
    ok &= comm._allreduce_stream(None, 2, 0) == 0
    ok &= comm.reduce_t[:2].tolist() == [1.5 * world * (world + 1) / 2, float(sum(r + 2 for r in range(world)))]
    comm.reduce_t[0] = float(rank)
    ok &= comm._allreduce_stream(None, 1, 1) == 0 and comm.reduce_t[0].item() == float(world - 1)
    results[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_transport_over_gloo(world):
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [results.get(r) for r in range(world)] == [True] * world


def test_slab_planner_partitions_the_lattice():
    cfg = nat.config_from_dict(scenes.get("dfsph_1m"))
    for world in (2, 4, 8):
        cuts, counts = nat.plan_slabs(cfg, world)
        assert cuts[0] == 0 and cuts[-1] == 161 and all(b - a >= 3 for a, b in zip(cuts, cuts[1:]))
        assert sum(counts) == 1000000
        # cell-column granularity: 20k particles per column; the cuts balance particles + ghosts, so the two end slabs (one cut each) own more
        assert max(counts) <= 1.35 * 1000000 / world
        if world > 2:
            assert counts[0] >= max(counts[1:-1]) - 20000 and counts[-1] >= max(counts[1:-1]) - 20000
    with pytest.raises(nat.SphError):
        nat.plan_slabs(nat.config_from_dict(scenes.get("dfsph_small")), 6)    # 16 cell columns cannot hold 6 slabs of >= 3 columns
    cuts, counts = nat.plan_slabs(nat.config_from_dict(scenes.get("dfsph_dam_x")), 3)   # fluid narrower than 3 slabs of >= 3 columns: widths clamp
    assert cuts[:2] == [0, 3] and 6 <= cuts[2] <= 18 and cuts[3] == 21 and counts == [480, 480, 0]


def test_replan_rule_properties():
    """sph_replan_slabs (host-only): equal-count cuts of a column histogram, never narrower than 3 columns (ghost layers + 1), and every cut
    stays strictly between its old neighbours so that migration only ever talks to the adjacent slab."""
    rng = np.random.default_rng(5)
    gx = 41
    for nslab in (2, 3, 4, 8):
        old = [round(k * gx / nslab) for k in range(nslab + 1)]
        for trial in range(50):
            hist = np.zeros(gx, dtype=np.int64)
            lo = int(rng.integers(0, gx - 4))
            hi = int(rng.integers(lo + 3, gx))
            hist[lo:hi] = rng.integers(1, 2000, hi - lo)
            new = nat.replan_slabs(hist, old)
            assert new[0] == 0 and new[-1] == gx
            for k in range(nslab):
                assert new[k + 1] >= new[k] + 3
            for k in range(1, nslab):
                assert old[k - 1] < new[k] < old[k + 1]
            old = new
    # a fixed distribution is reached after a few re-plans and then stays put (idempotence)
    hist = np.zeros(gx, dtype=np.int64)
    hist[20:36] = 1000
    cuts = [0, 3, 6, 9, gx]
    for _ in range(40):
        cuts = nat.replan_slabs(hist, cuts)
    assert cuts == nat.replan_slabs(hist, cuts)
    pre = np.concatenate([[0], np.cumsum(hist)])
    counts = [int(pre[cuts[k + 1]] - pre[cuts[k]]) for k in range(4)]
    assert counts == [4000, 4000, 4000, 4000], (cuts, counts)
    with pytest.raises(nat.SphError):
        nat.replan_slabs(hist, [0, 1, gx])
    # by cost (ghost_layers > 0): a slab pays for its particles AND for the ghost columns beyond each of its cuts, so the two end slabs -- one
    # cut each -- take more columns than the ones between them; the largest load is what is minimised; same bounds, same fixed-point property
    cuts = [0, 3, 6, 9, gx]
    for _ in range(40):
        cuts = nat.replan_slabs(hist, cuts, ghost_layers=2)
    assert cuts == nat.replan_slabs(hist, cuts, ghost_layers=2)
    counts = [int(pre[cuts[k + 1]] - pre[cuts[k]]) for k in range(4)]
    loads = [counts[k] + (2000 if k > 0 else 0) + (2000 if k < 3 else 0) for k in range(4)]
    assert counts == [5000, 3000, 3000, 5000] and max(loads) == 7000, (cuts, counts, loads)          # equal counts would cost 8000 on the inner slabs
    for nslab in (2, 3, 4, 8):
        old = [round(k * gx / nslab) for k in range(nslab + 1)]
        for trial in range(20):
            h2 = np.zeros(gx, dtype=np.int64)
            lo = int(rng.integers(0, gx - 4)); hi = int(rng.integers(lo + 3, gx))
            h2[lo:hi] = rng.integers(1, 2000, hi - lo)
            new = nat.replan_slabs(h2, old, ghost_layers=int(rng.integers(1, 3)))
            assert new[0] == 0 and new[-1] == gx and all(new[k + 1] >= new[k] + 3 for k in range(nslab)) and all(old[k - 1] < new[k] < old[k + 1] for k in range(1, nslab))
            old = new


def test_balanced_cuts_are_optimal_on_small_cases():
    """The cut planner (balanced_cuts in csrc/sph_host_scene.h, through sph_replan_slabs): the largest load -- 4 x particles + 5 x the particles of the
    ghost columns beyond each cut, in quarters of a particle -- is the smallest any admissible set of cuts can reach (brute force over all of them)."""
    rng = np.random.default_rng(11)
    gx, nslab = 15, 3
    old = [0, 5, 10, gx]
    for layers in (1, 2):
        for trial in range(40):
            hist = rng.integers(0, 500, gx).astype(np.int64)
            hist[rng.integers(0, gx, 3)] = 0
            pre = np.concatenate([[0], np.cumsum(hist)])

            def load(r, y, x):
                v = 4 * (pre[x] - pre[y])
                if r > 0:
                    v += 5 * (pre[y] - pre[max(y - layers, 0)])
                if r < nslab - 1:
                    v += 5 * (pre[min(x + layers, gx)] - pre[x])
                return int(v)
            best = None
            for c1 in range(max(old[0] + 1, 3), min(old[2] - 1, gx - 6) + 1):
                for c2 in range(max(old[1] + 1, c1 + 3), min(old[3] - 1, gx - 3) + 1):
                    m = max(load(0, 0, c1), load(1, c1, c2), load(2, c2, gx))
                    best = m if best is None else min(best, m)
            new = nat.replan_slabs(hist, old, ghost_layers=layers)
            got = max(load(0, 0, new[1]), load(1, new[1], new[2]), load(2, new[2], gx))
            assert got == best, (hist.tolist(), new, got, best)
