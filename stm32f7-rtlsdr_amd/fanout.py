"""Fan-out / fan-in of stream batches across the GPUs of one node (C1 scatter, C2 gather).

Streams are independent, so the DSP path itself has no collective: rank r owns the contiguous block of
n_streams/world streams and runs its own FmDemod handle.  RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the
CPU tests) is used only to distribute IQ batches from a root and to collect audio back.
"""
import torch
import torch.distributed as dist


def shard_range(n_streams, rank, world):
    """Contiguous block [lo, hi) of streams owned by `rank` (first n_streams % world ranks get one extra): sdrfm_shard_range of the
    C-ABI, the definition the plain-C multi-GPU host (examples/multi_gpu_main.c) uses too."""
    import ctypes as C
    from .lib import load_library
    first, count = C.c_uint32(), C.c_uint32()
    st = load_library().sdrfm_shard_range(int(n_streams), int(world), int(rank), C.byref(first), C.byref(count))
    if st != 0:
        raise ValueError("sdrfm_shard_range(%r, %r, %r): status %d" % (n_streams, world, rank, st))
    return first.value, first.value + count.value


def scatter_streams(iq_root, n_streams, nbytes, device, src=0, group=None):
    """C1: root holds [n_streams, nbytes] uint8; every rank receives its shard [n_local, nbytes].

    Implemented as grouped point-to-point sends (root egress is what bounds it on xGMI: 7 links per GPU)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_streams, rank, world)
    local = torch.empty((hi - lo, nbytes), dtype=torch.uint8, device=device)
    if world == 1:
        local.copy_(iq_root[lo:hi])
        return local
    ops = []
    if rank == src:
        for r in range(world):
            rlo, rhi = shard_range(n_streams, r, world)
            if r == src:
                local.copy_(iq_root[rlo:rhi])
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.isend, iq_root[rlo:rhi].contiguous(), r, group))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, local, src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return local


def gather_audio(audio_local, n_streams, dst=0, group=None):
    """C2: every rank contributes [n_local, n_audio] float32; root returns [n_streams, n_audio], others None."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return audio_local.clone()
    n_audio = audio_local.shape[1]
    ops, out = [], None
    if rank == dst:
        out = torch.empty((n_streams, n_audio), dtype=audio_local.dtype, device=audio_local.device)
        for r in range(world):
            rlo, rhi = shard_range(n_streams, r, world)
            if r == dst:
                out[rlo:rhi].copy_(audio_local)
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], r, group))
    elif audio_local.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, audio_local.contiguous(), dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out
