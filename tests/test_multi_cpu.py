"""Host-side scheduling of the several-devices entry points (psxhip_multi.cpp), no GPU needed: the contiguous
partition equals the one the ranks use (psxavenc_amd/parallel.py), the ticket queue hands out every unit exactly once
under contention, and the entry points fail loudly without a device."""
import threading

import numpy as np
import pytest


def test_shard_range_matches_the_rank_partition():
    from psxavenc_amd import multi
    from psxavenc_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 9, 1000, 10000, 540000, 2 ** 40 + 3):
        for world in (1, 2, 3, 4, 8, 64):
            covered = 0
            for r in range(world):
                assert multi.shard_range(n, r, world) == shard_range(n, r, world)
                f, c = multi.shard_range(n, r, world)
                assert f == covered
                covered += c
            assert covered == n
    assert multi.shard_range(10, 5, 4) == (0, 0) and multi.shard_range(10, -1, 4) == (0, 0)        # out of range: nothing


@pytest.mark.parametrize("n_units,ticket,threads", [(10000, 384, 8), (1, 5, 3), (0, 4, 2), (4097, 1, 16), (1250, 1536, 4)])
def test_ticket_queue_hands_out_every_unit_exactly_once(n_units, ticket, threads):
    from psxavenc_amd import multi
    q = multi.TicketQueue(n_units, ticket)
    got = [[] for _ in range(threads)]
    start = threading.Barrier(threads)

    def work(i):
        start.wait()
        while True:
            t = q.next()
            if t is None:
                return
            got[i].append(t)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert q.next() is None                                  # stays empty
    ranges = sorted(r for g in got for r in g)
    hits = np.zeros(n_units, np.int32)
    for f, c in ranges:
        assert 0 < c <= ticket and f % ticket == 0
        hits[f:f + c] += 1
    assert (hits == 1).all()
    assert len(ranges) == -(-n_units // ticket)
    for g in got:                                            # every worker draws in increasing order
        assert g == sorted(g)
    q.close()
    with pytest.raises(ValueError):
        multi.TicketQueue(10, 0)


def test_multi_entry_points_fail_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from psxavenc_amd import _lib, adpcm, multi
    with pytest.raises(_lib.PsxHipError) as e:
        multi.MdecMulti((0, 0), 0, 320, 240, 8192)
    assert e.value.code == _lib.PSXHIP_EDEVICE
    with pytest.raises(_lib.PsxHipError) as e:
        multi.xa_encode_streams_multi((0, 0), adpcm.XaSettings(1, True, 37800, 4, 1, 0), np.zeros((2, 4032), np.int16))
    assert e.value.code == _lib.PSXHIP_EDEVICE
