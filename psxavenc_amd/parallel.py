"""Multi-GPU sharding of the hot path.

Frames and ADPCM chains are independent units (encode_frame_bs resets all bitstream / DC state per
attempt, mdec.c:678-686; every XA channel / SPU stream owns its own state, adpcm.c:202-209), so the path
shards with NO data-path collective: rank r of R encodes a contiguous range of units and writes its own
slice of the output.  torch.distributed (RCCL on the GPU box, gloo in the CPU tests) is used only for
start/stop barriers and for gathering a few counters.
"""
from typing import List, Tuple


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: returns (first, count) for `rank`; the first n % world ranks get one extra."""
    assert 0 <= rank < world and n_units >= 0
    base, extra = divmod(n_units, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def shard_table(n_units: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n_units, r, world) for r in range(world)]


def str_frame_budgets(n_frames: int, base_overflow: int, overflow_den: int, first_frame: int = 0):
    """Per-frame byte budgets of the STR muxer (mdec.c:768-775) as a closed-form list, so any rank can
    compute the budgets of its own frame range: num += base; max = num / den * 2016; num %= den."""
    out = []
    num = 0
    for i in range(first_frame + n_frames):
        num += base_overflow
        size = num // overflow_den * 2016
        num %= overflow_den
        if i >= first_frame:
            out.append(size)
    return out


def gather_counters(dist, values, device=None):
    """all-gather a short list of per-rank integers (frames done, sum of quant scales, ...)."""
    import torch
    t = torch.tensor(values, dtype=torch.int64, device=device)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.tolist() for o in outs]


def encode_frames_sharded(encoder, frame_source, n_frames_total, frame_max_sizes, rank, world, dist=None):
    """Encode this rank's contiguous share of `n_frames_total` frames.

    frame_source(first, count) -> uint8 CUDA tensor (count, frame_bytes) for frames first..first+count-1.
    frame_max_sizes: int, or a sequence of n_frames_total ints (e.g. str_frame_budgets).
    Returns (first, d_out, d_results); the caller concatenates rank outputs in rank order.
    """
    import torch
    first, count = shard_range(n_frames_total, rank, world)
    d_frames = frame_source(first, count)
    if isinstance(frame_max_sizes, int):
        sizes = frame_max_sizes
    else:
        sizes = torch.tensor(list(frame_max_sizes[first:first + count]), dtype=torch.int32, device=d_frames.device)
    if dist is not None:
        dist.barrier()
    d_out, d_res = encoder.encode_frames_device(d_frames, sizes)
    if dist is not None:
        torch.cuda.synchronize(d_frames.device)
        dist.barrier()
    return first, d_out, d_res
