"""Split launches from several threads / contexts / streams at once: a frame's workgroups wait for each other, so launches that
were dispatched side by side could fill every CU slot with groups waiting for siblings still queued -- the per-device gate
(psxhip_api.cpp: SplitGate) orders them.  Eight threads, one encoder each, one frame per call (the reference's pattern, re-entrant
per object: SURVEY 8(b)); and six contexts launching 12-frame batches on six streams with no synchronisation in between."""
import threading

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_eight_threads_one_encoder_each_one_frame_per_call():
    from psxavenc_amd.mdec import MdecEncoder
    w, h, budget, n = 320, 240, 8192, 24
    frames = [O.synth_frames(w, h, n, seed=300 + t, amp=3 + t) for t in range(8)]
    wants = [O.mdec_encode(0, w, h, f, budget) for f in frames]
    assert all(rc == 0 for _, _, rc in wants)
    errors = []

    def work(t):
        try:
            enc = MdecEncoder(0, w, h, max_frame_size=budget, device=0)
            for rep in range(3):
                for k in range(n):
                    out, res = enc.encode_frames_host(frames[t][k:k + 1], budget)
                    if not (np.array_equal(out[0], wants[t][0][k]) and np.array_equal(res[0], wants[t][1][k])):
                        errors.append((t, rep, k, res[0].tolist(), wants[t][1][k].tolist()))
            enc.close()
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:4]


def test_six_contexts_on_six_streams_twelve_frames_per_launch():
    import torch
    from psxavenc_amd.mdec import MdecEncoder
    w, h, budget, n, C = 320, 240, 8192, 12, 6
    frames = [O.synth_frames(w, h, n, seed=400 + c, amp=2 + 3 * c) for c in range(C)]
    wants = [O.mdec_encode(c % 3, w, h, frames[c], budget) for c in range(C)]
    assert all(rc == 0 for _, _, rc in wants)
    encs = [MdecEncoder(c % 3, w, h, max_frame_size=budget, device=0) for c in range(C)]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(C)]
    d_fr = [torch.from_numpy(f).to("cuda:0") for f in frames]
    d_out = [torch.zeros((n, budget), dtype=torch.uint8, device="cuda:0") for _ in range(C)]
    d_res = [torch.zeros((n, 4), dtype=torch.int32, device="cuda:0") for _ in range(C)]
    torch.cuda.synchronize()          # (the buffers are filled on the default stream: nothing of that is in flight when the launches start)
    outs = []
    for rep in range(20):
        outs = [encs[c].encode_frames_device(d_fr[c], budget, d_out=d_out[c], d_results=d_res[c], stream=streams[c]) for c in range(C)]
    torch.cuda.synchronize()
    for c, (o, r) in enumerate(outs):
        assert np.array_equal(r.cpu().numpy(), wants[c][1]), c
        assert np.array_equal(o.cpu().numpy()[:, :budget], wants[c][0]), c
    lost = sum(e.watchdog() for e in encs)
    assert lost == 0
    for e in encs:
        e.close()


def test_a_stream_that_is_gone_does_not_trouble_the_next_launch():
    """the gate remembers the stream of the last split launch (another stream's launch has to be ordered behind it); the owner may have
    destroyed that stream meanwhile"""
    import gc
    import torch
    from psxavenc_amd.mdec import MdecEncoder
    w, h, budget = 160, 112, 4096
    frames = O.synth_frames(w, h, 3, seed=9, amp=5)
    want, want_res, rc = O.mdec_encode(0, w, h, frames, budget)
    assert rc == 0
    enc = MdecEncoder(0, w, h, max_frame_size=budget, device=0)
    d = torch.from_numpy(frames).to("cuda:0")
    torch.cuda.synchronize()
    for rep in range(3):
        s = torch.cuda.Stream(device="cuda:0")
        o1, r1 = enc.encode_frames_device(d, budget, stream=s)
        s.synchronize()
        del s
        gc.collect()
        o2, r2 = enc.encode_frames_device(d, budget)          # torch's current stream
        torch.cuda.synchronize()
        for o, r in ((o1, r1), (o2, r2)):
            assert np.array_equal(r.cpu().numpy(), want_res) and np.array_equal(o.cpu().numpy()[:, :budget], want)
    enc.close()
