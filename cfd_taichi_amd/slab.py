"""Multi-GPU x-slab decomposition: the host side (the reference is single-device).

One process per GPU.  Each rank creates a slab handle (`slab_rank`, `slab_count` in SphConfig); the native
library does all packing, sorting and ghost bookkeeping on the device and calls back into this module for
the transport:

  * ghost / migrating particles and per-sweep ghost fields: point-to-point with the left and right slab
    neighbour only (`batch_isend_irecv`) -- with the `nccl` backend (= RCCL) these are device buffers sent
    over the direct xGMI link of each neighbour pair; with `gloo` they are pinned host buffers (tests);
  * message sizes: one tiny all-gather of (send_left, send_right) counts per particle exchange;
  * residual sums and the CFL maximum: one small all-reduce each.

Two disciplines (include/sph_mi355x.h, SphComm): with `gloo` the library synchronises its stream around every transfer
and stages through pinned host buffers; with `nccl` the transfers are *stream-ordered* -- `TorchComm` makes the
library's own HIP stream torch's current stream (`torch.cuda.ExternalStream`) while it issues the RCCL send/recv and
all-reduce, so they queue up behind the pack kernels and in front of the unpack kernels without the host ever waiting:
a whole chunk of solver iterations with its halo refreshes and residual all-reduces is in flight at a time.
`SPH_SLAB_SYNC=1` switches the nccl transport back to the synchronous discipline.

    dist.init_process_group("nccl")
    sim = SlabSimulation(config, rank, world, device=local_rank)
    sim.step(10)
"""
import ctypes

import numpy as np

from . import _native as nat


class TorchComm:
    """SphComm callbacks on top of torch.distributed (backend nccl = RCCL, or gloo)."""

    def __init__(self, rank, world, device=0, capacity_bytes=64 << 20, group=None, stream_ptr=0, stream_ordered=None, reduce_capacity=4, host_loops=False):
        """host_loops: offer no allreduce_stream -- the library then runs the dfsph loops on the host, one read-back and one host all-reduce per
        residual (the minimal SphComm of include/sph_mi355x.h; pcisph / iisph need allreduce_stream).
        stream_ptr: the library handle's hipStream_t (Simulation.stream_ptr()); with the nccl backend and a stream the
        transport is stream-ordered unless stream_ordered=False / SPH_SLAB_SYNC=1."""
        import os
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.group = rank, world, group
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.on_host = dist.get_backend(group) != "nccl"
        self.device = torch.device("cpu") if self.on_host else torch.device("cuda", device)
        kw = {"pin_memory": True} if (self.on_host and torch.cuda.is_available()) else {}
        self.capacity = int(capacity_bytes)
        self.bufs = {k: torch.empty(self.capacity, dtype=torch.uint8, device=self.device, **kw)
                     for k in ("send_left", "send_right", "recv_left", "recv_right")}
        self.reduce_t = torch.zeros(max(4, int(reduce_capacity)), dtype=torch.float64, device=self.device, **kw)
        if stream_ordered is None:
            stream_ordered = os.environ.get("SPH_SLAB_SYNC", "0") != "1"
        self.stream_ordered = bool(stream_ordered and not self.on_host and stream_ptr)
        self.stream = torch.cuda.ExternalStream(stream_ptr, device=self.device) if self.stream_ordered else None
        self._ops_cache = {}
        self.error = None
        self.stats = {"exchange_counts": 0, "exchange_buffers": 0, "allreduce": 0, "allreduce_stream": 0, "bytes_sent": 0}
        self._cb_counts = nat.EXCHANGE_COUNTS_FN(self._exchange_counts)
        self._cb_buffers = nat.EXCHANGE_BUFFERS_FN(self._exchange_buffers)
        self._cb_allreduce = nat.ALLREDUCE_FN(self._allreduce)
        self._cb_allreduce_stream = nat.ALLREDUCE_STREAM_FN(self._allreduce_stream)
        self._cb_counts_n = nat.EXCHANGE_COUNTS_N_FN(self._exchange_counts_n)
        self.struct = nat.SphComm()
        self.struct.user = None
        self.struct.exchange_counts = self._cb_counts
        self.struct.exchange_buffers = self._cb_buffers
        self.struct.allreduce = self._cb_allreduce
        if not host_loops:
            self.struct.allreduce_stream = self._cb_allreduce_stream
        self.struct.exchange_counts_n = self._cb_counts_n
        self.struct.reduce_capacity = self.reduce_t.numel()
        self.struct.reduce_buf = self.reduce_t.data_ptr()
        self.struct.stream_ordered = 1 if self.stream_ordered else 0
        for k, t in self.bufs.items():
            setattr(self.struct, k, t.data_ptr())
        self.struct.capacity = self.capacity
        self.struct.on_host = 1 if self.on_host else 0

    # ---- transport primitives (also called directly by the CPU tests) -------------------------
    def exchange_counts(self, send_left, send_right):
        """Returns (recv_left, recv_right): what the left neighbour sends right, and vice versa."""
        torch, dist = self.torch, self.dist
        mine = torch.tensor([send_left, send_right], dtype=torch.int32, device=self.device)
        allc = torch.empty(2 * self.world, dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        allc = allc.cpu().view(self.world, 2)
        recv_left = int(allc[self.left, 1]) if self.left is not None else 0
        recv_right = int(allc[self.right, 0]) if self.right is not None else 0
        return recv_left, recv_right

    def exchange_counts_n(self, send_left, send_right):
        """n ints to each neighbour in ONE round trip; returns (recv_left, recv_right) lists (zeros where there is no neighbour)."""
        torch, dist = self.torch, self.dist
        n = len(send_left)
        mine = torch.tensor(list(send_left) + list(send_right), dtype=torch.int32, device=self.device)
        allc = torch.empty(2 * n * self.world, dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        allc = allc.cpu().view(self.world, 2, n)
        recv_left = [int(v) for v in allc[self.left, 1]] if self.left is not None else [0] * n
        recv_right = [int(v) for v in allc[self.right, 0]] if self.right is not None else [0] * n
        return recv_left, recv_right

    def _p2p_ops(self, sl, sr, rl, rr):
        """The isend / irecv descriptors of one halo message shape (a step uses three or four shapes over and over: cached)."""
        key = (sl, sr, rl, rr)
        ops = self._ops_cache.get(key)
        if ops is None:
            dist = self.dist
            ops = []
            if self.left is not None:
                if sl:
                    ops.append(dist.P2POp(dist.isend, self.bufs["send_left"][:sl], self.left, self.group))
                if rl:
                    ops.append(dist.P2POp(dist.irecv, self.bufs["recv_left"][:rl], self.left, self.group))
            if self.right is not None:
                if sr:
                    ops.append(dist.P2POp(dist.isend, self.bufs["send_right"][:sr], self.right, self.group))
                if rr:
                    ops.append(dist.P2POp(dist.irecv, self.bufs["recv_right"][:rr], self.right, self.group))
            if len(self._ops_cache) > 64:
                self._ops_cache.clear()
            self._ops_cache[key] = ops
        return ops

    def exchange_buffers(self, sl, sr, rl, rr):
        dist = self.dist
        ops = self._p2p_ops(sl, sr, rl, rr)
        if ops:
            if self.stream_ordered:
                # the library's stream is torch's current stream here: RCCL waits for what the library enqueued before this call (the
                # pack kernels), and req.wait() makes the library's stream -- not the host -- wait for the transfer
                with self.torch.cuda.stream(self.stream):
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
            else:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                if not self.on_host:
                    self.torch.cuda.current_stream(self.device).synchronize()   # the library continues on its own stream
        self.stats["bytes_sent"] += sl + sr

    def allreduce_stream(self, n, op):
        """In-place all-reduce of reduce_t[:n] (the library's (sum, count) pair / CFL maximum)."""
        dist = self.dist
        rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
        if self.stream_ordered:
            with self.torch.cuda.stream(self.stream):
                dist.all_reduce(self.reduce_t[:n], op=rop, group=self.group)      # blocks the library's stream only
        else:
            dist.all_reduce(self.reduce_t[:n], op=rop, group=self.group)
            if not self.on_host:
                self.torch.cuda.current_stream(self.device).synchronize()

    def allreduce(self, values, op):
        torch, dist = self.torch, self.dist
        t = torch.tensor(values, dtype=torch.float64).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=self.group)
        return t.cpu().tolist()

    # ---- C callbacks ---------------------------------------------------------------------------
    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:  # noqa: BLE001 - must not propagate through the C frame
            self.error = e
            return 1

    def _exchange_counts(self, user, send_left, send_right, recv_left, recv_right):
        def run():
            self.stats["exchange_counts"] += 1
            rl, rr = self.exchange_counts(send_left, send_right)
            recv_left[0], recv_right[0] = rl, rr
        return self._guard(run)

    def _exchange_counts_n(self, user, n, send_left, send_right, recv_left, recv_right):
        def run():
            self.stats["exchange_counts"] += 1
            rl, rr = self.exchange_counts_n([send_left[k] for k in range(n)], [send_right[k] for k in range(n)])
            for k in range(n):
                recv_left[k], recv_right[k] = rl[k], rr[k]
        return self._guard(run)

    def _exchange_buffers(self, user, sl, sr, rl, rr):
        def run():
            self.stats["exchange_buffers"] += 1
            self.exchange_buffers(sl, sr, rl, rr)
        return self._guard(run)

    def _allreduce_stream(self, user, n, op):
        def run():
            self.stats["allreduce_stream"] += 1
            self.allreduce_stream(n, op)
        return self._guard(run)

    def _allreduce(self, user, values, n, op):
        def run():
            self.stats["allreduce"] += 1
            out = self.allreduce([values[i] for i in range(n)], op)
            for i in range(n):
                values[i] = out[i]
        return self._guard(run)


def attach_native(sim, rank, capacity_bytes=64 << 20, group=None):
    """Native transport: rank 0's ncclGetUniqueId travels over the existing torch.distributed group, then every rank's library
    opens the communicator itself (collective) and drives ncclSend / ncclRecv / ncclAllReduce on its own stream."""
    import torch.distributed as dist
    box = [nat.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    sim.rccl_attach(box[0], capacity_bytes)


class SlabSimulation:
    """One rank's share of a sharded simulation: a slab handle plus its transport."""

    def __init__(self, config, rank, world, device=0, solver_name=None, capacity_bytes=64 << 20, slab_capacity=0, rebalance_every=0,
                 transport="torch", group=None, host_loops=False, comm_struct_size=None, **native_opts):
        """transport: "torch" (TorchComm callbacks: RCCL through torch.distributed, or gloo) or "native" (the library opens its
        own RCCL communicator and issues the transfers itself; one GPU per rank required)."""
        self.rank, self.world = rank, world
        cfg = nat.config_from_dict(config, solver_name=solver_name, device=device, slab_rank=rank, slab_count=world,
                                   slab_capacity=slab_capacity, slab_rebalance_every=rebalance_every, **native_opts)
        self.solver = {v: k for k, v in nat.SOLVER_IDS.items()}[cfg.solver]
        # a rigid body (config["solid"]): replicated on every rank (dfsph, two ghost columns); the sums that keep the copies identical go through the
        # transport's reduce buffer, 4 doubles per rigid sample
        rigid = None
        if config.get("solid"):
            from . import mesh
            rigid = mesh.rigid_from_config(config)
        self.rigid_active = bool(rigid and rigid.get("active"))
        self.sim = nat.Simulation(cfg, rigid=rigid)
        if transport == "native":
            self.comm = None
            attach_native(self.sim, rank, capacity_bytes, group)
        else:
            self.comm = TorchComm(rank, world, device=device, capacity_bytes=capacity_bytes, group=group, stream_ptr=self.sim.stream_ptr(),
                                  reduce_capacity=4 * self.sim.n_rigid + 8 if rigid else 4, host_loops=host_loops)
            self.sim.set_comm(self.comm.struct, comm_struct_size)
        self.n_fluid = self.sim.n_fluid

    def step(self, nsteps=1):
        try:
            if self.solver == "wcsph":
                self.sim.step_wcsph(nsteps)
                return None
            st = None
            for _ in range(nsteps):
                st = self.sim.step(1)
                if self.rigid_active:
                    self.sim.rigid_step()          # main.py:169-171
            return st
        except nat.SphError:
            if self.comm is not None and self.comm.error is not None:
                raise self.comm.error
            raise

    def owned(self, field):
        return self.sim.download_owned(field)

    def gather(self, field, dst=0):
        """All owned particles of all ranks assembled in original particle order on rank `dst` (else None)."""
        import torch.distributed as dist
        ids, vals = self.owned(field)
        parts = [None] * self.world if self.rank == dst else None
        dist.gather_object((ids, vals), parts, dst=dst)
        if self.rank != dst:
            return None
        shape = (self.n_fluid,) + vals.shape[1:]
        out = np.full(shape, np.nan, dtype=np.float32)
        seen = np.zeros(self.n_fluid, dtype=np.int32)
        for pid, pv in parts:
            out[pid] = pv
            np.add.at(seen, pid, 1)
        if not np.all(seen == 1):
            raise RuntimeError("slab ownership is not a partition: %d particles missing, %d duplicated" %
                               (int((seen == 0).sum()), int((seen > 1).sum())))
        return out

    def close(self):
        self.sim.close()
