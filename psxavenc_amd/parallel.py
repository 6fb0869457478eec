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


# ---------------------------------------------------------------------------------------------------------
# ADPCM chains sharded ALONG TIME: the one place on this path with a real exchange step.
# Rank r owns units [a_r, b_r) of every chain; the state at a_r is rank r-1's final state.  Every rank first
# encodes from a guessed state (speculate-and-verify inside the rank), then ranks exchange their chains' final
# states (an all-gather of 8 bytes per chain) and re-verify until a whole round changes nothing.  At that
# fixpoint every rank started from its predecessor's final state, i.e. the concatenation equals the serial encode.
# ---------------------------------------------------------------------------------------------------------
def time_shard_protocol(session, rank, world, initial_states):
    """Generator implementing one rank's side.  `session.run(start_states, known) -> (final_states, changed)`.
    Yields (final_states, changed_flag); the driver sends back the list of all ranks' yields in rank order.
    Returns (StopIteration.value) the chains' final states as seen by the LAST rank."""
    import numpy as np
    n = int(np.asarray(initial_states).shape[0])
    start = np.asarray(initial_states, dtype=np.int32).reshape(n, 2).copy()
    known = np.ones(n, np.uint8) if rank == 0 else np.zeros(n, np.uint8)
    final, _ = session.run(start, known)
    changed = True
    for _ in range(world + 1):
        gathered = yield (final.copy(), int(changed))
        if not any(flag for (_, flag) in gathered):
            return gathered[world - 1][0]
        changed = False
        if rank > 0:
            truth = np.asarray(gathered[rank - 1][0], dtype=np.int32).reshape(n, 2)
            if not known.all() or not np.array_equal(truth, start):
                start = truth.copy()
                known[:] = 1
                new_final, rewrote = session.run(start, known)
                changed = bool(rewrote) or not np.array_equal(new_final, final)
                final = new_final
    raise RuntimeError("time_shard_protocol: no fixpoint after world+1 rounds (cannot happen: truth advances one rank per round)")


def run_time_sharded(session, rank, world, dist, initial_states, device=None):
    """SPMD driver of time_shard_protocol over torch.distributed (RCCL on GPUs, gloo on CPU)."""
    import numpy as np
    import torch
    gen = time_shard_protocol(session, rank, world, initial_states)
    msg = next(gen)
    while True:
        final, flag = msg
        payload = torch.tensor(np.concatenate([final.reshape(-1), [flag]]).astype(np.int64), device=device)
        outs = [torch.zeros_like(payload) for _ in range(world)]
        if dist is not None:      # (also at world size 1 when a group was forced: the collective then runs, over one rank)
            dist.all_gather(outs, payload)
        else:
            outs = [payload]
        gathered = []
        for o in outs:
            v = o.cpu().numpy()
            gathered.append((v[:-1].astype(np.int32).reshape(-1, 2), int(v[-1])))
        try:
            msg = gen.send(gathered)
        except StopIteration as done:
            return done.value


def simulate_time_sharded(sessions, initial_states):
    """Single-process lockstep driver (tests, single-GPU boxes): sessions[r] plays rank r."""
    world = len(sessions)
    gens = [time_shard_protocol(s, r, world, initial_states) for r, s in enumerate(sessions)]
    msgs = [next(g) for g in gens]
    while True:
        results, done = [], 0
        for g in gens:
            try:
                results.append(g.send(list(msgs)))
            except StopIteration as fin:
                results.append(fin.value)
                done += 1
        if done == world:
            return results[-1]
        assert done == 0, "ranks must finish in the same round"
        msgs = results
