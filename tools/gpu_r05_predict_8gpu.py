#!/usr/bin/env python3
"""What 8 GPUs will show on config 5 (xacd: 16 serial chains sharded ALONG TIME) -- measured on ONE GPU with 8 real sessions playing
the 8 ranks in lockstep (parallel.simulate_time_sharded runs parallel.time_shard_protocol, the code bench.py --config xacd --gpus 8
runs): per round, which ranks (re-)ran, for how long, and how many verify passes.  A rank's time on its own GPU is its session's
time here (sessions run one after the other, each with the whole GPU -- as on 8 GPUs); the predicted wall time of a step is
sum over rounds of (the slowest rank of the round) + rounds x the exchange (an all-gather of 8 bytes per chain: latency only).
usage: python tools/gpu_r05_predict_8gpu.py [--seconds 3600] [--world 8] [--chunk units,warmup] [--kinds 0,2,5] [--json out.json]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from psxavenc_amd import adpcm, synth
from psxavenc_amd.parallel import shard_range, simulate_time_sharded

KINDS = {0: "two tones + noise floor (tonal)", 2: "white noise", 5: "gated tone + noise floor"}


class Timed:
    """a session whose run() is timed (synchronous call) and logged per call"""
    def __init__(self, sess, rank, log):
        self.s, self.rank, self.log, self.calls = sess, rank, log, 0

    def run(self, start, known=None, max_passes=0):
        torch.cuda.synchronize()
        p0 = self.s.passes
        t = time.perf_counter()
        out = self.s.run(start, known, max_passes)
        dt = time.perf_counter() - t
        self.log.append({"rank": self.rank, "call": self.calls, "ms": dt * 1e3, "verify_passes": self.s.passes - p0, "changed": bool(out[1])})
        self.calls += 1
        return out


def one(kind, seconds, world, n_ch=8, seed=1, chunk=None):
    settings = adpcm.XaSettings(adpcm.PSX_AUDIO_XA_FORMAT_XACD, True, 37800, 4, 1, 0)
    sps = adpcm.xa_get_samples_per_sector(settings)
    n_sectors = int(seconds * 37800 / sps)
    res = {}
    for w in (1, world):
        sessions, logs, keep = [], [], []
        for rank in range(w):
            sec0, sec_cnt = shard_range(n_sectors, rank, w)
            lead_sec = min(sec0, 1)
            n_frames = (sec_cnt + lead_sec) * sps
            pcm = torch.empty((n_ch, n_frames * 2), dtype=torch.int16, device="cuda")
            for c in range(n_ch):
                for side in range(2):
                    synth.pcm_device(seed, 2 * c + side, (sec0 - lead_sec) * sps, n_frames, kind, device=0, out=pcm[c][side:], pitch=2)
            chains = adpcm.make_chains([(c * n_frames * 2 + lead_sec * sps * 2 + side) for c in range(n_ch) for side in range(2)], 2,
                                       sec_cnt * sps, sec_cnt * 72, unit_stride=2)
            base = np.array([c * sec_cnt * 144 + side for c in range(n_ch) for side in range(2)], np.int32)
            lead = np.full(2 * n_ch, lead_sec * 72, np.int32)
            d_units = torch.zeros((n_ch * sec_cnt * 144, 16), dtype=torch.uint8, device="cuda")
            cu, wu = adpcm.pick_chunking(int(chains["n_units"].sum()))
            if chunk and w > 1:
                cu, wu = chunk      # (--chunk units,warmup: another chunking for the ranks' shares)
            s = adpcm.AdpcmSession(pcm.reshape(-1), chains, base, 4, 4, d_units=d_units, lead_units=lead, chunk_units=cu, warmup_units=wu)
            keep.append((pcm, d_units, s))
            sessions.append(s)
        init = np.zeros((2 * n_ch, 2), np.int32)
        best = None
        for rep in range(3):
            log = []
            for s in sessions:
                s.reset()
            simulate_time_sharded([Timed(s, r, log) for r, s in enumerate(sessions)], init)
            rounds = max(e["call"] for e in log) + 1
            # call k of rank r is not round k for every rank (a rank whose start state stood skips a round): rebuild the rounds from
            # the protocol -- round 0 = everyone's first call; afterwards rank r re-runs in the round its predecessor's truth reaches it
            per_round = {}
            for e in log:
                per_round.setdefault(e["call"], []).append(e)
            wall = sum(max(x["ms"] for x in per_round[k]) for k in per_round)
            cur = {"predicted_step_ms_without_exchange": round(wall, 3), "rounds_with_work": rounds,
                   "per_round": [{"ranks_that_ran": len(per_round[k]), "slowest_ms": round(max(x["ms"] for x in per_round[k]), 3),
                                  "mean_ms": round(sum(x["ms"] for x in per_round[k]) / len(per_round[k]), 3),
                                  "verify_passes_max": max(x["verify_passes"] for x in per_round[k])} for k in sorted(per_round)],
                   "busy_ms_per_rank": [round(sum(e["ms"] for e in log if e["rank"] == r), 3) for r in range(w)]}
            if best is None or cur["predicted_step_ms_without_exchange"] < best["predicted_step_ms_without_exchange"]:
                best = cur
        res["world_%d" % w] = best
        for pcm, d_units, s in keep:
            s.close()
        del keep
        torch.cuda.empty_cache()
    return n_sectors * n_ch, res


def main():
    argv = sys.argv[1:]
    seconds = float(argv[argv.index("--seconds") + 1]) if "--seconds" in argv else 3600.0
    world = int(argv[argv.index("--world") + 1]) if "--world" in argv else 8
    exchange_ms = 0.1       # an 8-rank all-gather of 136 bytes over xGMI: latency only; 0.1 ms is a generous allowance (RCCL small-message latency is tens of us)
    out = {"seconds_of_audio_per_channel": seconds, "world": world, "exchange_ms_per_round_assumed": exchange_ms, "materials": {}}
    chunk = tuple(int(x) for x in argv[argv.index("--chunk") + 1].split(",")) if "--chunk" in argv else None
    out["chunk_override_units_warmup"] = chunk
    kinds = [int(x) for x in argv[argv.index("--kinds") + 1].split(",")] if "--kinds" in argv else list(KINDS)
    for kind in kinds:
        name = KINDS[kind]
        sectors, res = one(kind, seconds, world, chunk=chunk)
        w1, wn = res["world_1"], res["world_%d" % world]
        asm1 = 8 * 0.205          # sector assembly, 8 channels x 0.205 ms (profiles/r05a_xacd_*_summary.txt), split over the ranks
        t1 = w1["predicted_step_ms_without_exchange"] + asm1
        tn = wn["predicted_step_ms_without_exchange"] + asm1 / world + exchange_ms * (wn["rounds_with_work"] + 1)
        out["materials"][name] = {"sectors": sectors, "one_gpu": w1, "n_gpus": wn, "one_gpu_step_ms": round(t1, 3), "n_gpu_step_ms_predicted": round(tn, 3),
                                  "one_gpu_sectors_per_sec": round(sectors / t1 * 1e3), "n_gpu_sectors_per_sec_predicted": round(sectors / tn * 1e3),
                                  "predicted_speedup": round(t1 / tn, 2)}
        print(name, json.dumps(out["materials"][name]), flush=True)
    if "--json" in argv:
        with open(argv[argv.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
