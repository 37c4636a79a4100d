"""Single-rank RCCL check of the stream-ordered transport plumbing (a 1-GPU box cannot host two RCCL ranks): the library's stream
wrapped as torch's current stream, an in-place all-reduce of the device reduce buffer ordered between two library-side operations,
and an (empty) halo exchange.  Exit code 0 = ok."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from cfd_taichi_amd import _native as nat
    from cfd_taichi_amd import scenes
    from cfd_taichi_amd.slab import TorchComm
    sim = nat.Simulation(nat.config_from_dict(scenes.get("dfsph_tiny_wall")))
    ptr = sim.stream_ptr()
    assert ptr, "the handle's stream is a real (non-null) hipStream_t"
    comm = TorchComm(0, 1, device=0, capacity_bytes=1 << 16, stream_ptr=ptr)
    assert comm.stream_ordered and not comm.on_host and comm.struct.stream_ordered == 1
    assert comm.struct.reduce_buf == comm.reduce_t.data_ptr()
    # work enqueued on the library's stream before the all-reduce must be seen by it, and the result by what follows -- without a host wait
    ext = comm.stream
    with torch.cuda.stream(ext):
        comm.reduce_t[:2] = torch.tensor([3.25, 7.0], dtype=torch.float64, device="cuda")
    assert comm._allreduce_stream(None, 2, 0) == 0, comm.error
    with torch.cuda.stream(ext):
        out = (comm.reduce_t[:2] * 2.0).clone()
    sim.step_dfsph(2)                              # the library keeps using the same stream
    sim.synchronize()
    assert out.tolist() == [6.5, 14.0], out.tolist()
    assert comm._allreduce_stream(None, 1, 1) == 0
    assert comm._exchange_buffers(None, 0, 0, 0, 0) == 0
    sim.synchronize()
    assert comm.reduce_t[0].item() == 3.25
    # the synchronous discipline on the same backend
    sync = TorchComm(0, 1, device=0, capacity_bytes=1 << 16, stream_ptr=ptr, stream_ordered=False)
    assert not sync.stream_ordered and sync.struct.stream_ordered == 0
    sync.reduce_t[0] = 2.0
    assert sync._allreduce_stream(None, 1, 0) == 0 and sync.reduce_t[0].item() == 2.0
    # the native transport: the library's own communicator of one rank (dlopen'ed librccl, ncclCommInitRank, ncclAllReduce and an
    # empty neighbour exchange on the handle's stream), then ordinary stepping on the same handle
    sim.rccl_attach(nat.rccl_unique_id(), 1 << 16)
    assert sim.rccl_selftest([1.5, -2.0, 4.0], 0) == [1.5, -2.0, 4.0]
    assert sim.rccl_selftest([1.5, -2.0], 1) == [1.5, -2.0]
    sim.step_dfsph(2)
    sim.synchronize()
    sim.close()
    dist.destroy_process_group()
    print("nccl single-rank transport ok")


if __name__ == "__main__":
    main()
