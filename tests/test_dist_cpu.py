"""Multi-process sharding of the hot path on CPU (gloo, world_size 2): every rank takes a contiguous unit range,
no data-path collective; only a barrier and a counter all-gather are exchanged (psxavenc_amd/parallel.py).
The per-rank encode is played by the CPU oracle here (no GPU in this container): what is under test is the
partition / ordering / counter logic that bench.py and the N-GPU path rely on."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from psxavenc_amd.parallel import shard_range, shard_table
    for n in (0, 1, 7, 8, 1000, 10000, 77760000):
        for world in (1, 2, 3, 4, 8):
            t = shard_table(n, world)
            assert t[0][0] == 0 and sum(c for _, c in t) == n
            for (f0, c0), (f1, _) in zip(t, t[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in t) - min(c for _, c in t) <= 1
    assert shard_range(10000, 3, 8) == (3750, 1250)          # config 'sbs v3': 1250 frames per GPU


def test_str_budget_sequence_is_rank_independent():
    from psxavenc_amd.parallel import str_frame_budgets
    full = str_frame_budgets(40, 1050, 120)
    assert full[:5] == [16128, 18144, 18144, 18144, 16128]
    assert str_frame_budgets(13, 1050, 120, first_frame=27) == full[27:40]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as O
    from psxavenc_amd.parallel import gather_counters, shard_range, str_frame_budgets
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    w, h, total = 48, 32, 21
    budgets = str_frame_budgets(total, 9, 4)                  # 2.25 sectors per frame -> 4032, 4032, 4032, 6048, ...
    first, count = shard_range(total, rank, world)
    frames = O.synth_frames(w, h, count, seed=77, amp=8, first=first)
    dist.barrier()
    out, res, rc = O.mdec_encode(1, w, h, frames, budgets[first:first + count], stride=max(budgets))
    assert rc == 0
    dist.barrier()
    counters = gather_counters(dist, [count, int(res[:, 0].sum())])
    np.save(os.path.join(tmpdir, "out_%d.npy" % rank), out)
    if rank == 0:
        np.save(os.path.join(tmpdir, "counters.npy"), np.array(counters))
    dist.destroy_process_group()


def test_two_rank_sharded_encode_equals_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from psxavenc_amd.parallel import str_frame_budgets
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w, h, total = 48, 32, 21
    budgets = str_frame_budgets(total, 9, 4)
    frames = O.synth_frames(w, h, total, seed=77, amp=8)
    want, want_res, rc = O.mdec_encode(1, w, h, frames, budgets, stride=max(budgets))
    assert rc == 0
    got = np.concatenate([np.load(tmp_path / "out_0.npy"), np.load(tmp_path / "out_1.npy")])
    assert np.array_equal(got, want)
    counters = np.load(tmp_path / "counters.npy")
    assert counters[:, 0].tolist() == [11, 10]
    assert int(counters[:, 1].sum()) == int(want_res[:, 0].sum())


# ---------------------------------------------------------------- ADPCM chains sharded along time
class _OracleSession:
    """CPU stand-in for psxavenc_amd.adpcm.AdpcmSession: encodes this rank's unit range serially with the oracle from
    whatever start state it is told (zero when unknown) -- guess quality only affects the number of rounds."""

    def __init__(self, pcm, first_unit, n_units):
        self.pcm, self.first, self.n = pcm, first_unit, n_units
        self.out = [None] * len(pcm)

    def run(self, start_states, known=None):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        finals, changed = [], False
        for c, x in enumerate(self.pcm):
            st = O.Chan(int(start_states[c][0]), int(start_states[c][1]))
            seg = x[self.first * 28:(self.first + self.n) * 28]
            if self.n > 0:
                data, st = O.spu_encode(seg, state=st)
            else:
                data = np.zeros(0, np.uint8)
            if self.out[c] is None or not np.array_equal(self.out[c], data):
                changed = True
            self.out[c] = data
            finals.append([st.prev1, st.prev2])
        return np.array(finals, np.int32), changed


def _time_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as O
    from psxavenc_amd.parallel import run_time_sharded, shard_range
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n_units = 41
    pcm = [O.synth_pcm(3, c, 0, n_units * 28, k) for c, k in enumerate((0, 4, 2))]
    first, count = shard_range(n_units, rank, world)
    sess = _OracleSession(pcm, first, count)
    final = run_time_sharded(sess, rank, world, dist, np.array([[5, -7], [0, 0], [100, 200]], np.int32))
    np.save(os.path.join(tmpdir, "adpcm_%d.npy" % rank), np.concatenate([o for o in sess.out]))
    np.save(os.path.join(tmpdir, "final_%d.npy" % rank), final)
    dist.destroy_process_group()


def test_adpcm_time_sharding_protocol_two_ranks(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from psxavenc_amd.parallel import shard_range
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_time_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n_units = 41
    starts = [[5, -7], [0, 0], [100, 200]]
    want, finals = [], []
    for c, k in enumerate((0, 4, 2)):
        x = O.synth_pcm(3, c, 0, n_units * 28, k)
        data, st = O.spu_encode(x, state=O.Chan(*starts[c]))
        want.append(data)
        finals.append([st.prev1, st.prev2])
    got = [np.load(tmp_path / ("adpcm_%d.npy" % r)) for r in range(2)]
    for c in range(3):
        parts = []
        for r in range(2):
            f, cnt = shard_range(n_units, r, 2)
            offs = sum(shard_range(n_units, r, 2)[1] * 16 for _ in range(c))
            parts.append(got[r][offs:offs + cnt * 16])
        assert np.array_equal(np.concatenate(parts), want[c]), c
    assert np.load(tmp_path / "final_0.npy").tolist() == finals and np.load(tmp_path / "final_1.npy").tolist() == finals


def test_adpcm_time_sharding_protocol_simulated_many_ranks():
    """lockstep simulation with more ranks than some chains have units (empty ranges pass the state through)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from psxavenc_amd.parallel import shard_range, simulate_time_sharded
    n_units, world = 11, 5
    pcm = [O.synth_pcm(8, c, 0, n_units * 28, k) for c, k in enumerate((4, 0))]
    sessions = [_OracleSession(pcm, *shard_range(n_units, r, world)) for r in range(world)]
    final = simulate_time_sharded(sessions, np.zeros((2, 2), np.int32))
    for c in range(2):
        data, st = O.spu_encode(pcm[c])
        got = np.concatenate([s.out[c] for s in sessions])
        assert np.array_equal(got, data)
        assert final[c].tolist() == [st.prev1, st.prev2]
