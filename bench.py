#!/usr/bin/env python3
"""bench.py -- BS-v2 320x240 frames/sec on MI355X (BASELINE.json metric, config "sbs v2").

One "step" = one pass of the hot path over one batch: N_FRAMES synthetic NV21 frames already resident
in HBM -> psxhip_mdec_encode_frames_device -> N_FRAMES x 8192-byte BS frames + results in HBM.
Multi-GPU: one process per GPU (torchrun), frames sharded by rank with no data-path collective; RCCL
is used only for the start/stop barrier and the max-over-ranks reduction of the elapsed time (weak
scaling: every rank encodes its own N_FRAMES).

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      algorithmic bytes (w*h*3/2 read + budget written per frame) / mean kernel time from HIP
                events on the launch stream, against the 8 TB/s HBM peak
  cpu_baseline  oracle/ (this repo's CPU restatement of the reference path, "port") timed on one host core
                on a bounded sample of the same frames.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (one process per GPU)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not args.share_gpu:
        raise SystemExit("--gpus %d but only %d GPU(s) visible (use --share-gpu --dist-backend gloo to test on one)" % (args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def _init_dist(args):
    """Returns (rank, world, local_rank, dev, dist-or-None, xdev)."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if args.share_gpu:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: GPU %d not visible (%d present)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        # --force-dist: a world-size-1 group, so that the RCCL leg (init with device_id, barrier(device_ids), all_gather of
        # device tensors) runs on the one GPU a dev box has before an 8-GPU node runs it for the first time
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a rank that dies must not leave the others in a collective for ever: they fail after this and report it (see _report_failure)
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world, timeout=tmo)
    xdev = dev if args.dist_backend == "nccl" else torch.device("cpu")     # where the tiny exchange tensors live
    return rank, world, local_rank, dev, dist, xdev


def _dist_info(args, dist):
    """what carried the barriers and the gathers of this run (None: a single process without a process group)"""
    if dist is None:
        return None
    import torch
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "exchange_tensors_on": "device" if args.dist_backend == "nccl" else "host",
            "forced_at_world_size_1": bool(args.force_dist and dist.get_world_size() == 1)}
    try:
        info["nccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if args.dist_backend == "nccl" else None
    except Exception:
        pass
    return info


def _barrier(args, dist, local_rank):
    import torch
    torch.cuda.synchronize()
    if dist is not None:
        if args.dist_backend == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()
    torch.cuda.synchronize()


def _gather_ranks(dist, xdev, values):
    """all-gather a few per-rank numbers (float64); returns a list of lists in rank order."""
    import torch
    t = torch.tensor(values, dtype=torch.float64, device=xdev)
    if dist is None:
        return [t.tolist()]
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.tolist() for o in outs]


LINE_LIMIT = 6144      # bytes: the driver's record keeps the last 8 KB of stdout; the line stays well inside it (tests/test_bench_line.py)


def _short(v, n=160):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def _scalars(d, n=160, skip=()):
    """the scalar entries of a dict (strings cut to n characters): what the one-line record keeps of a block"""
    return {k: _short(v, n) for k, v in (d or {}).items() if k not in skip and (v is None or isinstance(v, (bool, int, float, str)))}


def compact_line(full, detail_path=None):
    """The ONE line stdout carries: the contract's keys for the headline and nothing that grows.  `full` is the complete record
    (every child's line, counter dumps, notes); it goes to a side file (`detail_file`) and to stderr, never to stdout.  Children are
    summarised as {value, unit, frac, bit_exact} in config.secondary_summary."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_region_s", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    c = _scalars(cfg, 420)
    for k in ("quant_scale_hist", "budget_cycle", "secondary_summary", "legs_summary"):
        if cfg.get(k) is not None:
            c[k] = cfg[k]
    line["config"] = c
    roof = full.get("roofline") or {}
    r = _scalars(roof, 140, skip=("note", "traffic_note", "kernel_ms_method"))          # (traffic_key stays: tools/make_*_profile_summary.py key the counter index on it)
    if isinstance(roof.get("issue"), dict):
        r["valu_busy_frac"] = roof["issue"].get("valu_busy_frac")
    if isinstance(roof.get("step"), dict):
        r["step_frac"] = roof["step"].get("frac")
    line["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb:
        b = _scalars(cb, 200)
        if isinstance(cb.get("all_cores"), dict):
            b["all_cores"] = {k: cb["all_cores"].get(k) for k in ("value", "cores", "nproc")}
        line["cpu_baseline"] = b
    else:
        line["cpu_baseline"] = None
    line["parity"] = full.get("parity")
    line["results_sane"] = full.get("results_sane")
    if full.get("per_rank"):
        line["per_rank"] = [{k: pr.get(k) for k in ("rank", "frames_per_sec", "sectors_per_sec", "kernel_ms", "roofline_frac", "results_sane") if k in pr} for pr in full["per_rank"]]
    if full.get("predicted") is not None:
        line["predicted"] = full["predicted"]
    line["dist"] = _scalars(full.get("dist")) if full.get("dist") else None
    line["detail_file"] = detail_path
    # belt and braces: the line never outgrows the limit, whatever a future key brings along
    for victim in (("config", "secondary_summary"), ("per_rank",), ("config", "quant_scale_hist"), ("cpu_baseline", "sample"), ("config", "workload")):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        d = line
        for k in victim[:-1]:
            d = d.get(k) or {}
        if victim[-1] in d:
            d[victim[-1]] = "dropped: line over %d bytes, see detail_file" % LINE_LIMIT
    return line


def _emit(full, args):
    """full record -> side file + stderr; compact line (<= LINE_LIMIT bytes) -> the last line of stdout"""
    path = getattr(args, "detail_file", None)
    if not path:
        d = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail_%s_n%d.json" % (getattr(args, "config", None) or "run", full.get("n_gpus") or 1))
        except OSError:
            path = os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1)
            fh.write("\n")
        shown = os.path.relpath(path, ROOT)
    except OSError:
        shown = None
    sys.stderr.write("bench detail: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    line = compact_line(full, shown)
    s = json.dumps(line)
    assert len(s) <= LINE_LIMIT, len(s)
    print(s, flush=True)


AUDIO_KINDS = {0: "two tones + noise floor (tonal: a wrong start state survives thousands of units -- the hard case)",
               1: "quiet tone + noise", 2: "white noise, full scale (states coincide within a few units)", 3: "silence",
               4: "one tone, half scale", 5: "gated tone + noise floor (bursts and silence)"}


def _stats(xs):
    xs = sorted(xs)
    n = len(xs)
    med = xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])
    return {"median": round(med, 5), "min": round(xs[0], 5), "max": round(xs[-1], 5), "mean": round(sum(xs) / n, 5), "n": n}


def _profile_traffic(key):
    """HBM-side traffic per launch of the dominant kernel, from the committed rocprofv3 PMC summary whose key
    (workload + library version) matches this run -- PMC counters cannot be read from inside the process, they come
    from separate `rocprofv3 --pmc` passes of this same command (tools/gpu_rocprof_mdec.sh).  None when no summary matches."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_index.json")) as fh:
            idx = json.load(fh)
        e = idx.get(key)
        if e:
            return e["traffic_bytes_per_launch"], e["source"], e
        # no PMC pass of exactly this library / kernel revision: the newest one of the same workload, SAID to be of another revision
        # (a lookup that silently missed left "traffic": null in a driver-run child)
        wl = key.split(" | ")[0]
        same = sorted((k for k in idx if k.split(" | ")[0] == wl), key=lambda k: [int(x) if x.isdigit() else x for x in __import__("re").split(r"(\d+)", k.split(" | ")[-1])])
        if same:
            e = dict(idx[same[-1]])
            e["measured_on"] = same[-1].split(" | ")[-1]
            return e["traffic_bytes_per_launch"], e["source"] + " (counters of %s, not of this revision)" % e["measured_on"], e
    except Exception:
        pass
    return None, None, None


def _run_cpu_bench(argv):
    """oracle/cpu_bench (a C pthread harness over the CPU checker, oracle/cpu_bench.c) -> its JSON line, or None"""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "cpu_bench")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "cpu_bench"], check=False, stdout=subprocess.DEVNULL)
    if not os.path.exists(exe):
        return None
    r = subprocess.run([exe] + [str(a) for a in argv], capture_output=True, text=True)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def _usable_cores():
    """(threads worth running, nproc, note): the GPU box shows 256 logical CPUs but its cgroup grants 16 CPUs' worth of time
    (cpu.max 1600000 100000) -- more threads than that only time-slice (measured: 1 -> 16 threads scale 16.0x, 32..256 threads
    stay at the 16-thread rate or below)."""
    nproc = os.cpu_count() or 1
    try:
        nproc = min(nproc, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    use = nproc if quota is None else max(1, min(nproc, int(quota)))
    note = "all %d logical CPUs" % nproc if quota is None else "cgroup cpu quota %.1f CPUs of %d logical (threads beyond the quota only time-slice)" % (quota, nproc)
    return use, nproc, note


def _cpu_baseline_mdec(codec, w, h, budget, amp, seed, seconds):
    """oracle/mdec_oracle.c on the host cores of the GPU box: one core (the reference is single-threaded: the faithful
    number), then every core, one encoder per thread over disjoint frames (legal: no globals).  Timed by a C
    pthread harness (oracle/cpu_bench.c) on a bounded sample of the same workload (same generator, same budget)."""
    nthr, cores, cores_note = _usable_cores()
    one = _run_cpu_bench(["mdec", 1, seconds * 0.6, codec, w, h, budget, amp, seed])
    if not one:
        return None
    out = {"value": one["units_per_sec"], "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same workload in %.1f s (oracle/mdec_oracle.c, gcc -O3, C harness oracle/cpu_bench.c; the "
                     "FFmpeg-linked reference cannot be built here)" % (one["units"], one["seconds"])}
    many = _run_cpu_bench(["mdec", nthr, seconds * 0.4, codec, w, h, budget, amp, seed])
    if many:
        out["all_cores"] = {"value": many["units_per_sec"], "unit": "frames/s", "cores": nthr, "nproc": cores, "cores_note": cores_note,
                            "speedup_vs_1_core": round(many["units_per_sec"] / max(one["units_per_sec"], 1e-9), 1),
                            "per_thread_min": many["per_thread_min"], "per_thread_max": many["per_thread_max"],
                            "sample": "%d frames in %.1f s, one encoder per pthread" % (many["units"], many["seconds"])}
    return out


def _cpu_baseline_xa(seed, seconds):
    """the reference's own psx_audio_xa_encode (libpsxav/adpcm.c compiled unchanged into oracle/_ref) when it is there, else
    this repository's restatement; 37800 Hz 4-bit stereo XACD sectors; one core, then every core (C pthread harness)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "libpsxav_ref.so")
    extra = [ref] if os.path.exists(ref) else []
    nthr, cores, cores_note = _usable_cores()
    one = _run_cpu_bench(["xa", 1, seconds * 0.6, seed] + extra)
    if not one:
        return None
    out = {"value": one["units_per_sec"], "unit": "sectors/s", "cores": 1, "kind": one["kind"],
           "sample": "%d sectors of the same signal class in %.1f s (%s; C harness oracle/cpu_bench.c)"
                     % (one["units"], one["seconds"], "libpsxav/adpcm.c compiled unchanged, gcc -O3 -ffast-math" if one["kind"] == "reference"
                        else "oracle/adpcm_oracle.c, gcc -O3")}
    many = _run_cpu_bench(["xa", nthr, seconds * 0.4, seed] + extra)
    if many:
        out["all_cores"] = {"value": many["units_per_sec"], "unit": "sectors/s", "cores": nthr, "nproc": cores, "cores_note": cores_note,
                            "speedup_vs_1_core": round(many["units_per_sec"] / max(one["units_per_sec"], 1e-9), 1),
                            "sample": "%d sectors in %.1f s, one encoder state per pthread" % (many["units"], many["seconds"])}
    return out


# BASELINE.json's configs by name; "sbs_v2" is the headline metric and the default
PRESETS = {
    "sbs_v2": dict(workload="sbs", codec=0, width=320, height=240, budget=8192, amp=4, frames=1000, total_frames=0,
                   baseline_config="sbs v2: 1000 synthetic 320x240 YCbCr frames, 1 GPU, bit-exact check"),
    "sbs_v3": dict(workload="sbs", codec=1, width=640, height=480, budget=8192, amp=4, frames=0, total_frames=10000,
                   baseline_config="sbs v3: 10 000 synthetic 640x480 frames sharded across 8xMI355X over xGMI"),
    "strcd": dict(workload="strcd", frames=1000,
                  baseline_config="strcd v2: 320x240 15 fps + 37800 Hz 4-bit stereo XA, 2x speed, 1 GPU (combined MDEC+ADPCM path)"),
    "xacd": dict(workload="xacd", audio_seconds=3600.0, xa_channels=8,
                 baseline_config="xacd: 8-channel 37800 Hz 4-bit XA, 60 min synthetic audio, 8 GPU ADPCM-only throughput"),
}


def _predicted(args, world):
    """What DESIGN section 5 commits to for this preset at this world size, so that a SCALE record reads against the prediction
    without opening DESIGN.md.  Predictions from one-GPU measurements (profiles/r05_predict_*), never a measured curve."""
    if args.config == "sbs_v2":
        return {"value": round(8.7e6 * world), "unit": "frames/s", "basis": "N x 8.7 M: no data-path exchange (DESIGN section 5)"}
    if args.config == "sbs_v3":
        v = {1: 1.5e6, 2: 3.6e6, 4: 5.3e6, 8: 14.5e6}.get(world)
        return {"value": v, "unit": "frames/s", "basis": "10 000 frames strong-sharded; per-rank launches measured on one GPU (DESIGN section 5)"} if v else None
    if args.config == "xacd" and getattr(args, "audio_kind", 0) == 0:
        v = {1: 27e6, 2: 35e6, 4: 58e6, 8: 68e6}.get(world)
        return {"value": v, "unit": "sectors/s", "basis": "lockstep simulation of N sessions on one GPU, profiles/r05_predict_{2,4,8}gpu_xacd.json"} if v else None
    return None


def _report_failure(exc):
    """One JSON line instead of (in front of) a traceback: whichever rank fails says so on stdout -- the driver keeps the tail of
    stdout -- and exits non-zero; the other ranks run into the collective's timeout (--dist-timeout) and report that."""
    import traceback
    tb = traceback.format_exc().strip().splitlines()
    line = {"error": "%s: %s" % (type(exc).__name__, exc), "rank": int(os.environ.get("RANK", "0")),
            "world_size": int(os.environ.get("WORLD_SIZE", "1")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
            "where": tb[-3:] if len(tb) >= 3 else tb, "argv": sys.argv[1:]}
    print(json.dumps(line), flush=True)


def main():
    try:
        return _main()
    except SystemExit as e:
        if e.code not in (0, None) and not isinstance(e.code, int):
            _report_failure(e)
        raise
    except BaseException as e:      # noqa: BLE001 -- the line is the point
        _report_failure(e)
        sys.exit(1)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(PRESETS), default=None,
                    help="one of BASELINE.json's configs by name: sbs_v2 (the headline metric; the default workload), sbs_v3 (config 4: "
                         "10 000 640x480 v3 frames STRONG-sharded over the GPUs, 1250 per GPU at 8), strcd (config 3), xacd (config 5: "
                         "8 channels x 60 min).  Explicit flags below override a preset's values")
    ap.add_argument("--launches-per-step", type=int, default=0,
                    help="a step = this many back-to-back launches, cycling over --batches distinct batches (default: enough for "
                         "--steps 20 to time over ten seconds of GPU work: 4800 launches of 1000 320x240 frames)")
    ap.add_argument("--audio-kind", type=int, default=0, help="xacd: synthetic material (psxhip_synth_pcm_device kind): 0 two tones + noise "
                    "floor -- tonal, start-state guesses do not converge, the hard case and the default; 2 white noise; 5 gated tone")
    ap.add_argument("--str-streams", type=int, default=1, help="strcd: independent streams per psxhip_str_encode_device call (their XA tracks share the verify passes); "
                    "1 = BASELINE's config 3 (one stream); the eight-streams-per-call rate is measured beside it as a leg")
    ap.add_argument("--batches", type=int, default=4, help="sbs: distinct input batches the launches cycle over")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per launch (config 'sbs v2': 1000)")
    ap.add_argument("--total-frames", type=int, default=None, help="sbs: frames per launch over ALL GPUs (strong scaling; overrides --frames)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--budget", type=int, default=None, help="frame_max_size = sbs alignment (args.c:184)")
    ap.add_argument("--codec", type=int, default=None, help="0 = BS v2, 1 = v3, 2 = v3dc")
    ap.add_argument("--budget-cycle", default=None,
                    help="sbs: per-frame budgets cycling over this comma-separated list instead of one uniform --budget, e.g. "
                         "16128,18144,18144,18144 = what encode_sector_str asks for at 320x240 15 fps 2x speed (mdec.c:768-775; config 3's video leg alone, device-resident)")
    ap.add_argument("--amp", type=int, default=None, help="synthetic noise amplitude (4: final scale 3; 8: scale 5-6)")
    ap.add_argument("--content", choices=["uniform", "mixed"], default="uniform",
                    help="sbs: uniform = every frame the same noise amplitude (--amp; the headline), mixed = the scene-structured sequence of "
                         "psxavenc_amd/mixed.py (scenes of 5..30 frames at noise +-2..40, ~5 %% hand-made frames), batch b = frames [b n, (b + 1) n) of it")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--streams", type=int, default=1, help="sbs: contexts / streams the launches are dealt over (default 1: in-order launches on one stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steps-only", action="store_true", help="strcd: nothing but whole steps in the process -- no legs, no parity run (for the counter passes of tools/gpu_r05_strcd_pmc.sh)")
    ap.add_argument("--no-secondary", action="store_true", help="sbs: skip the untimed secondary measurements (noise +-8, cold context)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--check-frames", type=int, default=64, help="frames diffed against the oracle after timing")
    ap.add_argument("--workload", choices=["sbs", "xacd", "strcd"], default=None,
                    help="sbs = the headline MDEC metric (default); xacd = config 5, ADPCM-only XA sectors/s; "
                         "strcd = config 3, MDEC + XA ADPCM muxed into 2352-byte sectors")
    ap.add_argument("--audio-seconds", type=float, default=None, help="xacd: seconds of 37800 Hz stereo audio per XA channel")
    ap.add_argument("--xa-channels", type=int, default=None)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (testing: several ranks may share a GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even with one rank (world size 1): runs the RCCL leg -- init_process_group('nccl', device_id), "
                         "barrier(device_ids), device all_gather -- on a one-GPU box")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds a collective waits for a rank that has died before the others report failure")
    ap.add_argument("--lanes", type=int, default=None,
                    help="sbs: launch lanes of the encoder context (psxhip_mdec_set_lanes): 1 = every launch ordered on the caller's stream behind "
                         "the one before; 2 (the headline's default) = consecutive launches of the ONE context on the ONE caller stream may overlap "
                         "(inputs ordered on the stream, results ordered by the next call or psxhip_mdec_fence)")
    ap.add_argument("--device-list", default=None,
                    help="sbs: devices for the secondary C-ABI device-list leg (psxhip_mdec_multi_*), e.g. '0,0' to run it on a one-GPU box; "
                         "default: all visible devices when there is more than one")
    ap.add_argument("--detail-file", default=None, help="where the full record goes (default gpurun_out/bench_detail_<config>_n<N>.json); stdout carries the compact line only")
    ap.add_argument("--no-config-secondaries", action="store_true",
                    help="default run only: skip the other BASELINE configs (sbs_v3 share, xacd, strcd) and the RCCL world-size-1 self-test that "
                         "are run as subprocesses after the timed region")
    args = ap.parse_args()
    preset = dict(PRESETS[args.config or "sbs_v2"])
    if args.workload and args.workload != preset["workload"]:          # --workload alone keeps its round-2 meaning
        preset = {"workload": args.workload, "baseline_config": None}
        if args.workload == "sbs":
            preset = dict(PRESETS["sbs_v2"])
    defaults = dict(codec=0, width=320, height=240, budget=8192, amp=4, frames=1000, total_frames=0, audio_seconds=600.0, xa_channels=8)
    for k, v in defaults.items():
        if getattr(args, k, None) is None:
            setattr(args, k, preset.get(k, v))
    args.workload = preset["workload"]
    args.baseline_config = preset.get("baseline_config")
    args.config = args.config or ("sbs_v2" if args.workload == "sbs" else args.workload)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _spawn_ranks(args)
    if args.workload == "xacd":
        return bench_xacd(args)
    if args.workload == "strcd":
        return bench_strcd(args)

    import numpy as np
    import torch

    rank, world, local_rank, dev, dist, xdev = _init_dist(args)

    from psxavenc_amd import _lib, synth
    from psxavenc_amd.mdec import MdecEncoder, query_geometry
    from psxavenc_amd.parallel import shard_range

    w, h, budget = args.width, args.height, args.budget
    cycle = [int(x) for x in args.budget_cycle.split(",")] if args.budget_cycle else None
    if cycle:
        budget = max(cycle)
    strong = args.total_frames > 0
    if strong:          # the job's frames are fixed; every rank takes its contiguous share
        first, n = shard_range(args.total_frames, rank, world)
    else:               # every rank encodes its own `--frames` frames per launch
        first, n = shard_range(args.frames * world, rank, world)
        assert n == args.frames
    fsz = w * h * 3 // 2
    nb = max(1, args.batches)
    lps = args.launches_per_step
    if lps <= 0:        # >= ~0.5 s of GPU work per step at the measured rates: 20 steps time > 10 s (a 5-second utilisation sampler cannot miss it)
        lps = max(nb, min(4800, int(round(4800 * (1000 * 115200) / float(max(n, 1) * fsz)))))
    ns = max(1, args.streams)
    encs = [MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank) for _ in range(ns)]
    enc = encs[0]
    # launch lanes of the ONE context on the ONE caller stream (psxhip_mdec_set_lanes): 2 = consecutive launches may overlap (a
    # launch's tail is filled by the next one's head); the caller double-buffers its outputs (one buffer per distinct batch here)
    # and orders itself behind the last launch with psxhip_mdec_fence at the end of every step
    lanes = args.lanes if args.lanes is not None else (2 if ns == 1 and nb >= 2 else 1)
    if ns > 1:
        lanes = 1
    if lanes > 1:
        enc.set_lanes(lanes)
    geo = query_geometry(args.codec, w, h, budget, device=local_rank)
    # `nb` distinct batches of this rank's frames: batch b = the same frame indices drawn with seed + b
    if args.content == "mixed":
        from psxavenc_amd import mixed
        whole = mixed.frames_device(w, h, args.seed, first * nb, n * nb, device=local_rank)
        d_batches = [whole[b * n:(b + 1) * n] for b in range(nb)]
    else:
        d_batches = [synth.frames_device(w, h, args.seed + b, first, n, args.amp, device=local_rank) for b in range(nb)]
    d_frames = d_batches[0]
    ostride = (budget + 3) & ~3
    # --budget-cycle: frame i of the job (not of the rank's share) gets cycle[i % len]: any rank can budget its own range
    d_sizes = torch.tensor([cycle[(first + i) % len(cycle)] for i in range(n)], dtype=torch.int32, device=dev) if cycle else None
    d_outs = [torch.zeros((n, ostride), dtype=torch.uint8, device=dev) for _ in range(max(ns, nb))]
    d_ress = [torch.zeros((n, 4), dtype=torch.int32, device=dev) for _ in range(max(ns, nb))]
    # --streams 1 (default): every launch on torch's current stream, one context.  --streams S: S contexts on S streams,
    # launches dealt round-robin -- consecutive launches then overlap (one launch's tail with the next one's head), the way
    # two independent encoders sharing a GPU would
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(ns - 1)]
    torch.cuda.synchronize()

    def launch(k=0):
        i = k % ns
        b = k % nb
        o = b if ns == 1 else i          # one output buffer per batch (in-order launches) or per stream (overlapping launches)
        encs[i].encode_frames_device(d_batches[b], budget if d_sizes is None else d_sizes, d_out=d_outs[o], d_results=d_ress[o], stream=streams[i])

    for k in range(args.warmup * lps):
        launch(k)
    if lanes > 1:
        enc.fence(streams[0])
    torch.cuda.synchronize()

    # HIP events on the stream the kernel is launched on (torch's current stream).  One pair per STEP, bracketing its `lps`
    # back-to-back launches: the kernel's average launch duration is that time / lps.  (One pair per launch was the first
    # version: the two extra packets between consecutive kernels cost 5 % of the throughput being measured -- 161 us per
    # launch against 153 us without them, rocprofv3 putting the kernel itself at 155 us.)  With --streams S > 1 launches
    # of different streams overlap and a step has no single stream to bracket it: one pair per launch there.
    per_step = ns == 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps * (1 if per_step else lps))]
    _barrier(args, dist, local_rank)
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        if per_step:
            ev[s_][0].record(streams[0])
        for j in range(lps):
            k = s_ * lps + j
            if not per_step:
                ev[k][0].record(streams[k % ns])
            launch(k)
            if not per_step:
                ev[k][1].record(streams[k % ns])
        if per_step:
            if lanes > 1:
                enc.fence(streams[0])          # the step's last launch is ordered into the stream: the event below sees all of them
            ev[s_][1].record(streams[0])
    _barrier(args, dist, local_rank)
    elapsed_local = time.perf_counter() - t0

    kernel_ms = [a.elapsed_time(b) / (lps if per_step else 1) for a, b in ev]      # per launch
    kstat = _stats(kernel_ms)
    alg_bytes = (fsz + budget) * n                       # per launch: NV21 read + frame_max_size written, per frame
    if cycle:
        alg_bytes = fsz * n + sum(cycle[(first + i) % len(cycle)] for i in range(n))
    achieved_local = alg_bytes / (kstat["mean"] * 1e-3) / 1e9
    if ns > 1:      # launches overlap: a launch's own duration says little, the aggregate rate is what the GPU sustains
        achieved_local = alg_bytes * args.steps * lps / elapsed_local / 1e9
    overlapped = None
    if lanes > 1:
        # With two launch lanes consecutive launches overlap, and a step's event time / its launches is a launch's SHARE of the
        # GPU's time, not its duration.  The roofline block describes the KERNEL: its duration is measured right here, live, with
        # HIP events around the same launches in strict stream order (one lane: every launch waits for the one before) -- the
        # figure rocprofv3's per-kernel average of `bench.py --lanes 1` agrees with (profiles/).  `value` stays the timed region's.
        overlapped = {"launch_lanes": lanes, "ms_per_launch_share": kstat["mean"], "share_stats": kstat,
                      "achieved": round(alg_bytes / (kstat["mean"] * 1e-3) / 1e9, 3), "frac": round(alg_bytes / (kstat["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                      "note": "algorithmic bytes of a launch / (a step's HIP-event time / its launches): what the GPU sustains with the head of launch k+1 "
                              "filling the tail of launch k; in a kernel trace of this run a kernel's own span is about twice its share"}
        enc.set_lanes(1)
        m = min(lps, 400)
        for k in range(m // 4):
            launch(k)
        io = []
        for _blk in range(5):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(streams[0])
            for k in range(m):
                launch(k)
            b_.record(streams[0])
            torch.cuda.synchronize()
            io.append(a_.elapsed_time(b_) / m)
        kstat = _stats(io)
        kstat["launches"] = 5 * m
        achieved_local = alg_bytes / (kstat["mean"] * 1e-3) / 1e9
        enc.set_lanes(lanes)
    per_rank = _gather_ranks(dist, xdev, [elapsed_local, float(n * args.steps * lps), kstat["mean"], achieved_local])
    elapsed = max(r[0] for r in per_rank)                   # max over ranks

    # ---- post-timing: every rank checks its results are sane; rank 0 diffs a sample of every batch against the oracle
    ress = [d_ress[b if ns == 1 else 0].cpu().numpy() for b in range(nb if ns == 1 else 1)]
    res = ress[0]
    ok_local = all(bool(((r[:, 0] >= 1) & (r[:, 0] <= 63)).all()) for r in ress)
    scale_sum = _gather_ranks(dist, xdev, [float(sum(int(r[:, 0].sum()) for r in ress)), float(ok_local)])
    parity = None
    cpu_baseline = None
    secondary = None
    if rank == 0:
        import oracle_lib as O
        per = max(1, min(args.check_frames, n * len(ress)) // len(ress))
        ok, checked = True, 0
        # (--streams S > 1: an output buffer per STREAM; stream 0's buffer holds the batch of its last launch)
        last_b0 = (((args.steps * lps - 1) // ns) * ns) % nb
        for b, r in enumerate(ress):
            idx = np.linspace(0, n - 1, min(per, n)).astype(np.int64)
            tidx = torch.from_numpy(idx).to(dev)
            fr = d_batches[b if ns == 1 else last_b0][tidx].cpu().numpy()
            want, want_res, rc = O.mdec_encode(args.codec, w, h, fr, budget if not cycle else np.array([cycle[(first + int(i)) % len(cycle)] for i in idx], np.int32), stride=budget)
            got = d_outs[b if ns == 1 else 0][tidx].cpu().numpy()[:, :budget]
            ok = ok and bool(rc == 0 and np.array_equal(got, want) and np.array_equal(r[idx], want_res))
            checked += int(idx.size)
        parity = {"frames_checked": checked, "batches_checked": len(ress), "bit_exact": ok}
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = _cpu_baseline_mdec(args.codec, w, h, budget, args.amp, args.seed, args.cpu_seconds)
        if world == 1 and not args.no_secondary and ns == 1:
            secondary = _secondary_sbs(args, torch, dev, local_rank, w, h, budget, n, first)
            if torch.cuda.device_count() > 1 or args.device_list:
                secondary["c_abi_device_list"] = _secondary_device_list(args, torch, w, h, budget, d_batches)
            if args.config == "sbs_v2" and not args.no_config_secondaries:
                for e in encs:          # (the children get the GPU to themselves)
                    e.close()
                encs = []
                secondary["per_call_drop_in"] = _secondary_per_call()
                secondary.update(_secondary_configs(args))

    version = _lib.lib().psxhip_version().decode()
    wl_key = "sbs codec=%d %dx%d budget=%s frames=%d %s | %s" % (args.codec, w, h, ("%d" % budget) if not cycle else "cycle(" + args.budget_cycle + ")", n,
                                                                  ("amp=%d" % args.amp) if args.content == "uniform" else "content=mixed", version)
    traffic, traffic_src, pmc = _profile_traffic(wl_key)
    # what actually bounds the kernel: VALU issue.  A wave64 VALU instruction holds its SIMD's issue slot for 4 cycles;
    # instructions per launch come from the same committed PMC pass as the traffic (SQ_INSTS_VALU).
    issue = None
    if pmc and pmc.get("valu_insts_per_launch") and not getattr(args, "share_gpu", False):      # (ranks sharing one GPU: a rank's kernel time is not the GPU's)
        simds, clock_ghz = 256 * 4, 2.4
        eff_ms = kstat["mean"] if ns == 1 else elapsed_local * 1e3 / (args.steps * lps)     # (several contexts: their launches' share of the wall clock; launch lanes: kstat is the in-order kernel duration)
        slots = simds * clock_ghz * 1e9 / 4.0 * (eff_ms * 1e-3)
        issue = {"valu_insts_per_launch": pmc["valu_insts_per_launch"], "salu_insts_per_launch": pmc.get("salu_insts_per_launch"),
                 "lds_insts_per_launch": pmc.get("lds_insts_per_launch"),
                 "valu_issue_slots_per_launch": int(slots), "valu_busy_frac": round(pmc["valu_insts_per_launch"] / slots, 4),
                 "note": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction x this run's mean kernel time; counters from " + str(traffic_src)}

    total_frames = sum(r[1] for r in per_rank)
    value = total_frames / elapsed
    achieved = achieved_local
    scales, counts = np.unique(np.concatenate([r[:, 0] for r in ress]), return_counts=True)

    if rank == 0:
        headline = args.codec == 0 and w == 320 and h == 240
        sec_summary = _secondary_summary(secondary)
        line = {
            "metric": "bs_v2_320x240_frames_per_sec" if headline else "bs_%s_%dx%d_frames_per_sec" % (["v2", "v3", "v3dc"][args.codec], w, h),
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "timed_region_s": round(elapsed, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "sbs %s: %s synthetic %dx%d NV21 frames per launch (%d per GPU), %d launches per step cycling over %d distinct "
                                   "batches, frame_max_size %d, %s; inputs resident in HBM, outputs stay in HBM (no D2H inside the "
                                   "timed region)"
                                   % (["v2", "v3", "v3dc"][args.codec], ("%d" % args.total_frames) if strong else ("%d x %d" % (world, n)), w, h, n,
                                      lps, nb, budget, ("noise +-%d" % args.amp) if args.content == "uniform" else "scene-structured content (psxavenc_amd/mixed.py)"),
                       "preset": args.config, "baseline_config": args.baseline_config,
                       "frames_per_gpu_per_launch": n, "launches_per_step": lps, "distinct_batches": nb, "width": w, "height": h,
                       "frame_max_size": budget, "budget_cycle": cycle,
                       "secondary_summary": sec_summary,
                       "in_order_frames_per_sec": round(n / (kstat["mean"] * 1e-3), 1) if ns == 1 else None,
                       "in_order_note": "every launch waiting for the one before (one launch lane): n / roofline.kernel_ms; `value` is the two-lane rate" if lanes > 1 else None,
                       "parallelism": "frames sharded x%d (contiguous ranges per rank), no data-path collective" % world,
                       "kernel_shape": {"groups_per_cu": geo.groups_per_cu, "wavefronts_per_group": geo.wavefronts_per_group,
                                        "image_tile_bytes": geo.image_tile_bytes, "frames_in_flight": geo.frames_in_flight},
                       "quant_scale_hist": {str(int(s)): int(c) for s, c in zip(scales, counts)},
                       "library": version, "streams": ns, "contexts": ns, "launch_lanes": lanes,
                       "launch_order": ("one context, one caller stream, two launch lanes (psxhip_mdec_set_lanes): inputs stream-ordered, a launch's results ordered into the "
                                        "stream by the next call / psxhip_mdec_fence at the end of each step; consecutive launches overlap") if lanes > 1 else
                                       ("strict stream order: every launch waits for the one before" if ns == 1 else "%d contexts on %d streams" % (ns, ns))},
            "per_rank": [{"rank": i, "frames_per_sec": round(r[1] / r[0], 1), "elapsed_s": round(r[0], 4), "kernel_ms": round(r[2], 5),
                          "roofline_achieved_gbs": round(r[3], 2), "roofline_frac": round(r[3] / HBM_PEAK_GBS, 6),
                          "quant_scale_sum": int(q[0]), "results_sane": bool(q[1])} for i, (r, q) in enumerate(zip(per_rank, scale_sum))],
            "roofline": {"bound": "hbm", "kernel": "mdec_encode_frames_kernel", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "frac_overlapped": overlapped["frac"] if overlapped else None, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_key": wl_key, "kernel_ms": kstat["mean"], "kernel_ms_stats": kstat,
                         "launches_timed": (kstat.get("launches") if lanes > 1 else args.steps * lps),
                         "kernel_ms_method": ("strict stream order (one launch lane), measured after the timed region: one HIP event pair per block of %d back-to-back "
                                              "launches, / %d; stats over 5 blocks" % (min(lps, 400), min(lps, 400))) if lanes > 1 else
                                             (("one HIP event pair per step of %d back-to-back launches, / %d; stats over steps" % (lps, lps)) if per_step else "one HIP event pair per launch"),
                         "algorithmic_bytes_per_launch": alg_bytes, "issue": issue, "overlapped": overlapped,
                         **({"note": "launches of %d contexts overlap: achieved = algorithmic bytes of all launches / elapsed; kernel_ms are "
                                     "per-launch durations while sharing the GPU" % ns} if ns > 1 else {})},
            "cpu_baseline": cpu_baseline,
            "secondary": secondary,
            "parity": parity,
            "results_sane": all(bool(q[1]) for q in scale_sum),
            "dist": _dist_info(args, dist),
            "predicted": _predicted(args, world),
        }
        _emit(line, args)
    for e in encs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


def _secondary_summary(sec):
    """the secondaries in a few dozen bytes each, inside `config`: in-process figures as bare frames/s, child processes (the other
    BASELINE configs) as {value, unit, frac, bit_exact}; their full lines are in the detail file"""
    if not sec:
        return None
    out = {}

    def g(path, key):
        d = sec
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
            if d is None:
                return
        out[key] = d
    g(("noise_amp_8", "two_lanes", "frames_per_sec"), "noise_amp_8_two_lanes")
    g(("noise_amp_8", "one_lane_in_order", "frames_per_sec"), "noise_amp_8_in_order")
    g(("cold_context", "first_four_launches_two_lanes", "frames_per_sec"), "cold_context_two_lanes")
    g(("cold_context", "first_launch_alone", "frames_per_sec"), "cold_context_first_launch")
    g(("mixed_content", "two_lanes", "frames_per_sec"), "mixed_content_two_lanes")
    g(("mixed_content", "one_lane_in_order", "frames_per_sec"), "mixed_content_in_order")
    g(("mixed_content", "passes", "passes_per_frame"), "mixed_content_passes_per_frame")
    g(("per_call_drop_in", "encode_frame_bs_320x240_v2", "us_per_call_median"), "per_call_encode_frame_bs_us")
    g(("per_call_drop_in", "encode_frame_bs_320x240_v2", "frames_per_sec"), "per_call_encode_frame_bs_frames_per_sec")
    for name, child in sec.items():
        if isinstance(child, dict) and ("value" in child or "rc" in child):
            e = {"value": child.get("value"), "unit": child.get("unit"), "frac": (child.get("roofline") or {}).get("frac"),
                 "bit_exact": (child.get("parity") or {}).get("bit_exact")}
            if child.get("error"):
                e["error"] = _short(str(child["error"]), 60)
            cfg = child.get("config") or {}
            if "eight_streams_sectors_per_sec" in cfg:      # config 3: value = ONE stream (BASELINE's config); eight per call beside it
                e["s8"] = cfg["eight_streams_sectors_per_sec"]
            out[name] = e
    return out


def _secondary_sbs(args, torch, dev, local_rank, w, h, budget, n, first):
    """Untimed extras of the default line (after the timed region, rank 0, one GPU): the same batch size on content whose
    answer flips between neighbouring scales (noise +-8), and on a COLD context (no hint from a previous launch: every
    group's first frame runs the pilot).  The headline is the warm steady state; these are what it does not show."""
    import numpy as np
    from psxavenc_amd import synth
    from psxavenc_amd.mdec import MdecEncoder
    out = {}
    d_out = torch.zeros((n, (budget + 3) & ~3), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((n, 4), dtype=torch.int32, device=dev)

    def timed(enc, batches, launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(launches):
            enc.encode_frames_device(batches[k % len(batches)], budget, d_out=d_out, d_results=d_res)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / launches

    # The headline launches are in order on one stream: a launch's tail (a CU here and there still finishing its second frame) is
    # not filled by the next launch's head.  Two contexts on two streams -- what a caller with independent batches can do -- are.
    try:
        encs = [MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank) for _ in range(2)]
        strs = [torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)]
        outs = [(d_out, d_res), (torch.zeros_like(d_out), torch.zeros_like(d_res))]
        b4 = [synth.frames_device(w, h, args.seed + 300 + b, first, n, args.amp, device=local_rank) for b in range(4)]

        def both(launches):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(strs[0])
            strs[1].wait_stream(strs[0])
            for k in range(launches):
                i = k & 1
                encs[i].encode_frames_device(b4[k % 4], budget, d_out=outs[i][0], d_results=outs[i][1], stream=strs[i])
            strs[0].wait_stream(strs[1])
            b.record(strs[0])
            torch.cuda.synchronize()
            return a.elapsed_time(b) / launches

        both(16)
        ms2 = min(both(400), both(400))
        out["two_contexts_two_streams"] = {"frames_per_sec": round(n / ms2 * 1e3, 1), "ms_per_launch": round(ms2, 5), "launches": 400}
        for e in encs:
            e.close()
    except Exception as e:
        out["two_contexts_two_streams"] = {"error": repr(e)}
    # psxhip_mdec_encode_batches_device: a caller that holds four batches hands them over together -- one launch, frames of all four
    # drawn from one ticket counter, no launch boundary between them, no concatenation of its buffers
    try:
        enc = MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank)
        o4 = [(torch.zeros_like(d_out), torch.zeros_like(d_res)) for _ in range(4)]
        res4 = {}
        for amp_ in sorted({args.amp, 8}):
            bb = [synth.frames_device(w, h, args.seed + 400 + b, first, n, amp_, device=local_rank) for b in range(4)]
            lst = [(bb[i], o4[i][0], o4[i][1]) for i in range(4)]

            def run(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    enc.encode_batches_device(lst, budget)
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / reps
            run(4)
            ms4 = run(50)
            res4["noise_amp_%d" % amp_] = {"frames_per_sec": round(4 * n / ms4 * 1e3, 1), "ms_per_launch": round(ms4, 5), "launches": 50}
        res4["batches_per_launch"] = 4
        res4["frames_per_batch"] = n
        out["four_batches_one_launch"] = res4
        enc.close()
    except Exception as e:
        out["four_batches_one_launch"] = {"error": repr(e)}
    o4 = [(torch.zeros_like(d_out), torch.zeros_like(d_res)) for _ in range(4)]

    def rates(batches, launches=64):
        """frames/s for one launch of `n` frames at a time over four distinct batches: every launch waiting for the one before (one
        lane), the way the headline is measured (two lanes: results lag one call), and the four batches as ONE batch list"""
        r = {}
        for lanes in (1, 2):
            enc = MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank)
            if lanes > 1:
                enc.set_lanes(2)

            def run(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for k in range(reps):
                    enc.encode_frames_device(batches[k % 4], budget, d_out=o4[k % 4][0], d_results=o4[k % 4][1])
                if lanes > 1:
                    enc.fence()
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / reps
            run(8)
            ms = min(run(launches), run(launches))
            r["one_lane_in_order" if lanes == 1 else "two_lanes"] = {"frames_per_sec": round(n / ms * 1e3, 1), "ms_per_launch": round(ms, 5), "launches": launches}
            enc.close()
        enc = MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank)
        lst = [(batches[i], o4[i][0], o4[i][1]) for i in range(4)]

        def run4(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                enc.encode_batches_device(lst, budget)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps
        run4(3)
        ms4 = min(run4(16), run4(16))
        r["four_batches_one_launch"] = {"frames_per_sec": round(4 * n / ms4 * 1e3, 1), "ms_per_launch": round(ms4, 5)}
        enc.close()
        allr = torch.cat([o[1][:, 0] for o in o4]).cpu()
        sc, cn = allr.unique(return_counts=True)
        r["quant_scale_hist"] = {str(int(a)): int(b) for a, b in zip(sc.tolist(), cn.tolist())}
        return r

    def cold(batches, lanes, launches):
        """a FRESH context (no hint from a previous launch, no verdict on foreign hints): its first `launches` launches"""
        ts = []
        for t in range(5):
            enc = MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank)
            if lanes > 1:
                enc.set_lanes(2)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(launches):
                enc.encode_frames_device(batches[(t + k) % len(batches)], budget, d_out=o4[k % 4][0], d_results=o4[k % 4][1])
            if lanes > 1:
                enc.fence()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / launches)
            enc.close()
        ts.sort()
        return {"frames_per_sec": round(n / ts[len(ts) // 2] * 1e3, 1), "ms_per_launch_median": round(ts[len(ts) // 2], 5), "ms_min": round(ts[0], 5),
                "ms_max": round(ts[-1], 5), "contexts": len(ts), "launches_per_context": launches, "launch_lanes": lanes}

    def passes_hist(batches):
        """from the diagnostics instantiation (PSXHIP_MDEC_STATS=1, its own context): passes per frame over one cycle of the four batches"""
        import ctypes as C
        from psxavenc_amd import _lib
        os.environ["PSXHIP_MDEC_STATS"] = "1"
        try:
            enc = MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank)
        finally:
            del os.environ["PSXHIP_MDEC_STATS"]
        t = (C.c_ulonglong * 8)()
        for k in range(4):
            enc.encode_frames_device(batches[k], budget, d_out=d_out, d_results=d_res)
        torch.cuda.synchronize()
        _lib.lib().psxhip_mdec_read_stats(enc._h, t, 8, 1)
        for k in range(4):
            enc.encode_frames_device(batches[k], budget, d_out=d_out, d_results=d_res)
        torch.cuda.synchronize()
        _lib.lib().psxhip_mdec_read_stats(enc._h, t, 8, 1)
        enc.close()
        st = list(t)
        return {"frames": 4 * n, "passes_per_frame": round(st[1] / float(4 * n), 4), "frames_started_incl_handed_on": st[0],
                "passes_hist_per_frame_start_0_1_2_3_4_5plus": st[2:8]}

    try:
        b8 = [synth.frames_device(w, h, args.seed + 100 + b, first, n, 8, device=local_rank) for b in range(4)]
        r8 = rates(b8)
        out["noise_amp_8"] = dict(r8, frames_per_sec=r8["two_lanes"]["frames_per_sec"], measured_like="the headline: one context, one stream, two launch lanes, four distinct batches",
                                  passes=passes_hist(b8))
        b4 = [synth.frames_device(w, h, args.seed + 200 + b, first, n, args.amp, device=local_rank) for b in range(4)]
        c2 = cold(b4, 2, 4)
        out["cold_context"] = {"first_four_launches_two_lanes": c2, "first_launch_alone": cold(b4, 1, 1), "frames_per_sec": c2["frames_per_sec"],
                               "note": "a fresh context: no hint from a previous launch, every group's first frame runs the pilot"}
        # ---- content that is not the friendliest point of the space: the scene-structured sequence of
        #      psxavenc_amd/mixed.py -- runs of 5..30 similar frames, cuts between noise amplitudes 2..40, one frame in twenty flat /
        #      hard edges / escape-heavy -- 4 x n frames, one n-frame launch at a time
        if (w, h, budget, args.codec) == (320, 240, 8192, 0):
            from psxavenc_amd import mixed
            whole = mixed.frames_device(w, h, args.seed, first, 4 * n, device=local_rank)
            bm = [whole[i * n:(i + 1) * n] for i in range(4)]
            rm = rates(bm)
            out["mixed_content"] = dict(rm, frames_per_sec=rm["two_lanes"]["frames_per_sec"], passes=passes_hist(bm), cold_context=cold(bm, 2, 4),
                                        content="psxavenc_amd/mixed.py: %d frames, scenes of 5..30 frames at noise +-2..40, ~5 %% hand-made frames (flat, hard edges, escapes); "
                                                "integer-only, frame i a function of (seed, i)" % (4 * n),
                                        distinct_scales=len(rm["quant_scale_hist"]))
            import oracle_lib as O
            idx = np.linspace(0, 4 * n - 1, 48).astype(np.int64)
            fr = whole[torch.from_numpy(idx).to(dev)].cpu().numpy()
            want, want_res, rc = O.mdec_encode(args.codec, w, h, fr, budget)
            got = torch.cat([o[0] for o in o4])[torch.from_numpy(idx).to(dev)].cpu().numpy()[:, :budget]
            out["mixed_content"]["parity"] = {"frames_checked": int(idx.size), "bit_exact": bool(rc == 0 and np.array_equal(got, want))}
    except Exception as e:          # secondary figures never fail the bench line
        out["error"] = repr(e)
    # The frames of the headline are already NV21 in HBM.  Where they come from decoded pictures, the colour-conversion / scaling
    # front-end (psxhip_scaler_*; own arithmetic, parity with libswscale unpinned) makes them on the device: its
    # rate alone and in a chain with the encoder on one stream (320x240 targets only).
    try:
        if (w, h) == (320, 240) and args.codec == 0:
            from psxavenc_amd.frontend import Scaler
            sc = Scaler(0, 640, 480, w, h, device=local_rank)
            enc = MdecEncoder(args.codec, w, h, max_frame_size=budget, device=local_rank)
            d_src = torch.randint(0, 256, (n, sc.source_bytes), dtype=torch.uint8, device=dev)
            d_frames = torch.empty((n, sc.frame_bytes), dtype=torch.uint8, device=dev)

            def chain(reps, with_encoder):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    sc.convert_device(d_src, d_frames)
                    if with_encoder:
                        enc.encode_frames_device(d_frames, budget, d_out=d_out, d_results=d_res)
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / reps

            chain(3, True)
            ms_sc, ms_chain = chain(20, False), chain(20, True)
            out["pictures_to_bs_chain"] = {"source": "rgb24 640x480 (random pictures, resident in HBM)", "scaler_pictures_per_sec": round(n / ms_sc * 1e3, 1),
                                           "scaler_algorithmic_gbs": round((sc.source_bytes + sc.frame_bytes) * n / ms_sc / 1e6, 1),
                                           "chain_frames_per_sec": round(n / ms_chain * 1e3, 1), "parity": "front-end unpinned (libswscale absent)"}
            enc.close()
            sc.close()
    except Exception as e:
        out["pictures_to_bs_chain"] = {"error": repr(e)}
    return out


def _secondary_device_list(args, torch, w, h, budget, d_batches):
    """More than one GPU visible to a single-process run: the C-ABI's own sharding (psxhip_multi.cpp: one host thread, encoder
    context and pinned staging pair per listed device; the reference's host side is one C loop, filefmt.c:633-662, it has no
    ranks) over ALL visible devices, host buffers in and out, every byte compared with device 0 alone."""
    import numpy as np
    try:
        from psxavenc_amd.mdec import MdecEncoder
        from psxavenc_amd.multi import MdecMulti, SCHED_STATIC, SCHED_TICKETS
        devs = [int(x) for x in args.device_list.split(",")] if args.device_list else list(range(torch.cuda.device_count()))
        frames = np.concatenate([b.cpu().numpy() for b in d_batches], axis=0)
        n = frames.shape[0]
        one = MdecEncoder(args.codec, w, h, max_frame_size=budget, device=0)
        want, want_res = one.encode_frames_host(frames, budget)
        t0 = time.perf_counter()
        one.encode_frames_host(frames, budget, out=want, res=want_res)
        t_one = time.perf_counter() - t0
        one.close()
        out = {"devices": devs, "frames": int(n), "one_device_frames_per_sec": round(n / t_one, 1), "host_buffers": True}
        m = MdecMulti(devs, args.codec, w, h, max_frame_size=budget)
        for name, sched in (("static", SCHED_STATIC), ("tickets", SCHED_TICKETS)):
            got, got_res = m.encode_frames_host(frames, budget, schedule=sched)
            t0 = time.perf_counter()
            m.encode_frames_host(frames, budget, schedule=sched, out=got, res=got_res)
            dt = time.perf_counter() - t0
            out[name] = {"frames_per_sec": round(n / dt, 1), "bit_exact_vs_device_0": bool(np.array_equal(got, want) and np.array_equal(got_res, want_res)),
                         "per_device": m.last_report}
        m.close()
        return out
    except Exception as e:      # secondary figures never fail the bench line
        return {"error": repr(e)}


def _secondary_per_call():
    """The reference's own call pattern on the drop-in surface -- psx_audio_spu_encode per 28 samples (filefmt.c:243),
    psx_audio_xa_encode per sector (:184), encode_frame_bs per frame (:643) -- timed by a plain-C harness (examples/percall_bench.c)."""
    import subprocess
    try:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "examples"), "percall_bench"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        r = subprocess.run([os.path.join(ROOT, "examples", "percall_bench"), "2000", "300", "300"], capture_output=True, text=True, timeout=120)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}


def _secondary_configs(args):
    """The default run's other BASELINE configs, each as a child `python bench.py --config ...` after the headline's timed region
    (its own process: own contexts, own HBM, a failure stays its own), timed > 0.5 s, with its parity sample and CPU baseline:
    one GPU's share of config 4 (`sbs v3`, 1250 of the 10 000 frames), config 5 (`xacd`, all 540 000 sectors), config 3 (`strcd`).
    And the RCCL leg at world size 1 under torch.distributed.run, exactly as the driver launches N ranks."""
    import socket
    import subprocess
    me = os.path.abspath(__file__)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    seed = ["--seed", str(args.seed)]
    jobs = [
        ("sbs_v3_1250", [sys.executable, me, "--config", "sbs_v3", "--total-frames", "1250", "--steps", "8", "--warmup", "2", "--no-secondary",
                         "--cpu-seconds", "4"] + seed),
        ("xacd_config5", [sys.executable, me, "--config", "xacd", "--steps", "20", "--warmup", "2", "--cpu-seconds", "5"] + seed),
        # the same job on other material (config 5's number is set by its tonal test signal: one chain that does not converge)
        ("xacd_config5_white_noise", [sys.executable, me, "--config", "xacd", "--audio-kind", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"] + seed),
        ("xacd_config5_gated_tone", [sys.executable, me, "--config", "xacd", "--audio-kind", "5", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"] + seed),
        ("strcd_config3", [sys.executable, me, "--config", "strcd", "--steps", "150", "--warmup", "5", "--cpu-seconds", "4"] + seed),
        ("rccl_world_size_1", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                               "--master-port", str(port), me, "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "2", "--warmup", "1",
                               "--launches-per-step", "200", "--no-secondary", "--no-cpu-baseline"] + seed),
    ]
    keep = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "timed_region_s", "scaling", "config", "roofline", "cpu_baseline", "parity",
            "dist", "error", "detail_file")
    out = {}
    ddir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(ddir, exist_ok=True)
    for name, cmd in jobs:
        t0 = time.perf_counter()
        try:
            cmd = cmd + ["--detail-file", os.path.join(ddir, "bench_detail_child_%s.json" % name)]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            line = None
            for ln in reversed(r.stdout.strip().splitlines()):
                if ln.startswith("{"):
                    try:
                        line = json.loads(ln)
                        break
                    except Exception:
                        pass
            if line is None:
                out[name] = {"error": "no JSON line", "rc": r.returncode, "stderr_tail": r.stderr.strip().splitlines()[-3:]}
            else:
                out[name] = {k: line[k] for k in keep if k in line}
                out[name]["rc"] = r.returncode
        except Exception as e:
            out[name] = {"error": repr(e)}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        out[name]["cmd"] = " ".join(cmd[1:] if cmd[1] != "-m" else cmd[1:])
    return out


def bench_xacd(args):
    """Config 'xacd': 8 XA channels x stereo, 37800 Hz, 4-bit -> 2352-byte sectors.  16 serial chains; parallelism comes
    from speculate-and-verify along time (psxhip_adpcm_session_*), and across GPUs from time-sharding with a final-state
    all-gather (psxavenc_amd/parallel.py).  A step = encode the whole audio from scratch + assemble this rank's sectors."""
    import numpy as np
    import torch
    rank, world, local_rank, dev, dist, xdev = _init_dist(args)
    from psxavenc_amd import adpcm, synth
    from psxavenc_amd.parallel import run_time_sharded, shard_range

    settings = adpcm.XaSettings(adpcm.PSX_AUDIO_XA_FORMAT_XACD, True, 37800, 4, 1, 0)
    sps = adpcm.xa_get_samples_per_sector(settings)                      # 2016 sample frames per sector
    n_sectors = max(world, int(args.audio_seconds * 37800 / sps))        # per channel
    n_ch = args.xa_channels
    sec0, sec_cnt = shard_range(n_sectors, rank, world)                  # this rank's sectors of every channel
    units_per_sector_chain = 18 * 8 // 2                                 # per L/R chain
    lead_sec = min(sec0, 1)                                              # one sector of history for the start-state guess
    n_frames = (sec_cnt + lead_sec) * sps
    pcm = torch.empty((n_ch, n_frames * 2), dtype=torch.int16, device=dev)
    for c in range(n_ch):
        for side in range(2):
            synth.pcm_device(args.seed, 2 * c + side, (sec0 - lead_sec) * sps, n_frames, args.audio_kind, device=local_rank,
                             out=pcm[c][side:], pitch=2)
    chains = adpcm.make_chains([(c * n_frames * 2 + lead_sec * sps * 2 + side) for c in range(n_ch) for side in range(2)], 2,
                               sec_cnt * sps, sec_cnt * units_per_sector_chain, unit_stride=2)
    base = np.array([c * sec_cnt * 144 + side for c in range(n_ch) for side in range(2)], np.int32)
    lead = np.full(2 * n_ch, lead_sec * units_per_sector_chain, np.int32)
    d_units = torch.zeros((n_ch * sec_cnt * 144, adpcm.record_bytes(4)), dtype=torch.uint8, device=dev)
    init = np.zeros((2 * n_ch, 2), np.int32)
    torch.cuda.synchronize()

    chunk_units, warmup_units = adpcm.pick_chunking(int(chains["n_units"].sum()))       # this GPU's share: chunks are cut to fill ITS wavefront slots

    # the session (chunk tables, per-unit state storage) is set up once; a step starts the encode over on it
    sess = adpcm.AdpcmSession(pcm.reshape(-1), chains, base, 4, 4, d_units=d_units, lead_units=lead, chunk_units=chunk_units,
                              warmup_units=warmup_units)

    def step():
        sess.reset()
        run_time_sharded(sess, rank, world, dist, init, device=xdev)
        outs = [adpcm.xa_assemble_device(d_units[c * sec_cnt * 144:], sec_cnt, settings, first_lba=sec0) for c in range(n_ch)]
        return outs, sess.passes

    for _ in range(args.warmup):
        step()

    def barrier():
        _barrier(args, dist, local_rank)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs, passes = step()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = _gather_ranks(dist, xdev, [elapsed, float(sec_cnt)])
    elapsed = max(r[0] for r in per_rank)                   # max over ranks
    # ---- the dominant kernel, live: HIP events around the speculate launch (adpcm_chunks_kernel<false, ..>) and around the verify
    #      passes of a few further steps (psxhip_adpcm_session_set_timing: the events are on the session's own stream, inside the call)
    spec_ms, verify_ms = [], []
    try:
        sess.set_timing(True)
        for _ in range(3):
            step()
            a_, b_ = sess.last_timing()
            spec_ms.append(a_)
            verify_ms.append(b_)
        sess.set_timing(False)
    except Exception:
        pass

    parity = None
    cpu_baseline = None
    if rank == 0:
        import oracle_lib as O
        k = min(sec_cnt, 40)
        x = pcm[0][:(k * sps + 4032) * 2].cpu().numpy() if sec_cnt * sps >= k * sps + 4032 else np.concatenate(
            [pcm[0].cpu().numpy(), np.zeros(8064, np.int16)])
        os_ = O.XaSettings(1, 1, 37800, 4, 1, 0)
        want, _ = O.xa_encode(os_, x, k * sps, lba=0)
        got = outs[0][:k].cpu().numpy().reshape(-1)
        parity = {"sectors_checked": int(k), "bit_exact": bool(np.array_equal(got, want))}
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = _cpu_baseline_xa(args.seed, args.cpu_seconds)
        total_sectors = n_sectors * n_ch * args.steps
        value = total_sectors / elapsed
        alg = (sps * 4 + 2352) * sec_cnt * n_ch           # int16 stereo in + sector out, per step per rank
        spec_bytes = (sps * 4 + 144 * (adpcm.record_bytes(4) + 8)) * sec_cnt * n_ch      # the speculate kernel's own: PCM in, unit records + states out
        from psxavenc_amd import _lib as _plib
        try:
            _plib.lib().psxhip_adpcm_kernel_rev.restype = __import__("ctypes").c_char_p
            arev = _plib.lib().psxhip_adpcm_kernel_rev().decode()
        except Exception:
            arev = "?"
        xa_key = "xacd kind=%d ch=%d sectors=%d | %s" % (args.audio_kind, n_ch, n_sectors, arev)
        traffic, traffic_src, pmc = _profile_traffic(xa_key)
        kroof = None
        if spec_ms:
            sm = sorted(spec_ms)[len(spec_ms) // 2]
            kroof = {"kernel_ms": round(sm, 4), "kernel_ms_all": [round(x, 4) for x in spec_ms], "verify_ms": [round(x, 4) for x in verify_ms],
                     "achieved": round(alg / (sm * 1e-3) / 1e9, 3), "frac": round(alg / (sm * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
            if pmc and pmc.get("valu_insts_per_launch"):
                slots = 1024 * 2.4e9 / 4.0 * sm * 1e-3
                kroof["valu_busy_frac"] = round(pmc["valu_insts_per_launch"] / slots, 4)
                kroof["valu_insts_per_launch"] = pmc["valu_insts_per_launch"]
        _emit({
            "metric": "xa_37800_4bit_stereo_sectors_per_sec", "value": round(value, 2), "unit": "sectors/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "timed_region_s": round(elapsed, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic", "dist": _dist_info(args, dist),
            "config": {"workload": "xacd: %d XA channels x stereo x %.0f s @ 37800 Hz, 4-bit, %d sectors per channel, time-sharded x%d"
                                   % (n_ch, n_sectors * sps / 37800.0, n_sectors, world),
                       "preset": args.config, "baseline_config": args.baseline_config,
                       "material": AUDIO_KINDS.get(args.audio_kind, str(args.audio_kind)),
                       "verify_passes_last_step": passes, "chunk_units": chunk_units, "warmup_units": warmup_units, "realtime_factor": round(value * sps / 37800.0 / n_ch, 1)},
            "roofline": {"bound": "hbm", "kernel": "whole step: adpcm_chunks_kernel (speculate from guessed start states, verify passes) + 8 x xa_assemble_kernel",
                         "achieved": round(alg * args.steps / elapsed / 1e9, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_key": xa_key,
                         "algorithmic_bytes_per_launch": alg,
                         # the dominant kernel alone: ITS bytes (all PCM read once + unit records and per-unit states written) over ITS duration
                         "speculate_kernel": ({"kernel": "adpcm_chunks_kernel<false, 12>", "kernel_ms": kroof["kernel_ms"], "kernel_ms_all": kroof["kernel_ms_all"],
                                               "own_bytes": spec_bytes, "achieved": round(spec_bytes / (kroof["kernel_ms"] * 1e-3) / 1e9, 3),
                                               "frac": round(spec_bytes / (kroof["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                               "valu_busy_frac": kroof.get("valu_busy_frac"), "valu_insts_per_launch": kroof.get("valu_insts_per_launch"),
                                               "method": "HIP events on the session's stream around the speculate launch, 3 steps after the timed region (median)"} if kroof else None),
                         "kernel_ms": kroof["kernel_ms"] if kroof else None,
                         "verify_ms": kroof["verify_ms"] if kroof else None,
                         "per_kernel_counters": (pmc or {}).get("kernels"),
                         "note": "achieved / frac: algorithmic bytes of the step (PCM in + sectors out, SURVEY 8(d)) over the whole step's time -- speculate + verify "
                                 "passes + sector assemblies; the speculate kernel's own figure is `speculate_kernel` (a dependent chain per sound unit: VALU issue bounds it)"},
            "cpu_baseline": cpu_baseline, "parity": parity, "predicted": _predicted(args, world),
            "per_rank": [{"rank": i, "sectors_per_sec": round(n_ch * r[1] * args.steps / r[0], 1), "elapsed_s": round(r[0], 4)} for i, r in enumerate(per_rank)]}, args)
    sess.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_strcd(args):
    """Config 'strcd v2' (SURVEY 3.2 / 8(d)): 320x240 @15 fps BS v2 + 37800 Hz 4-bit stereo XA muxed into 2352-byte sectors,
    budgets cycling 16128 / 18144 x3.  A step = one psxhip_str_encode_device call over `--str-streams` independent streams of
    `--frames` frames per GPU: frames and PCM resident in HBM, sectors land in HBM (the headline's protocol, BASELINE section 4) --
    one batched MDEC launch over all streams' frames, the streams' XA tracks as chains of one speculate-and-verify session, video
    sectors built by a kernel, audio sectors assembled into their slots.  The host-buffer entry point (PCIe + host interleave
    inside the timed region) is measured beside it."""
    import numpy as np
    import torch
    rank, world, local_rank, dev, dist, xdev = _init_dist(args)
    from psxavenc_amd import _lib, strmux, synth
    from psxavenc_amd.parallel import shard_range
    w, h, n = 320, 240, args.frames
    S = max(1, args.str_streams)
    s = strmux.settings(fmt=strmux.FORMAT_STRCD, codec=0, width=w, height=h, fps_num=15, fps_den=1, cd_speed=2)
    first, count = shard_range(n * world, rank, world)
    # a little more audio than video, so that the video ends the stream (the reference's loop stops with whichever ends first)
    na = (strmux.plan(s, n).n_audio_sectors + 2) * 2016 + 100
    d_frames = torch.stack([synth.frames_device(w, h, args.seed + 17 * i, first, n, args.amp, device=local_rank) for i in range(S)])
    d_pcm = torch.zeros((S, na * 2), dtype=torch.int16, device=dev)
    for i in range(S):
        for c in range(2):
            synth.pcm_device(args.seed, 2 * i + c, 0, na, args.audio_kind, device=local_rank, out=d_pcm[i][c:], pitch=2)
    p = strmux.plan(s, n, na)
    mux = strmux.StrMuxer((local_rank,))
    d_out = torch.zeros((S, p.n_sectors, p.sector_size), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        mux.encode_device(s, d_frames, d_pcm, d_out=d_out)
    _barrier(args, dist, local_rank)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, p2 = mux.encode_device(s, d_frames, d_pcm, d_out=d_out)
    _barrier(args, dist, local_rank)
    elapsed_local = time.perf_counter() - t0
    per_rank = _gather_ranks(dist, xdev, [elapsed_local, float(p.n_sectors * S * args.steps), float(p2.quant_scale_sum)])
    elapsed = max(r[0] for r in per_rank)
    if rank == 0:
        import hashlib
        import oracle_lib as O
        # ---- legs (untimed): the video leg alone (the same frames, no audio track), one stream alone, the host-buffer entry point
        legs = {}

        def timed_dev(settings, fr, pc, reps):
            o = torch.zeros((fr.shape[0],) + (lambda q: (q.n_sectors, q.sector_size))(strmux.plan(settings, n, na if pc is not None else 0)), dtype=torch.uint8, device=dev)
            m = strmux.StrMuxer((local_rank,))
            for _ in range(3):
                m.encode_device(settings, fr, pc, d_out=o)
            t = time.perf_counter()
            for _ in range(reps):
                m.encode_device(settings, fr, pc, d_out=o)
            dt = (time.perf_counter() - t) / reps
            m.close()
            return dt, o
        try:
            if args.steps_only:
                raise RuntimeError("skipped (--steps-only: every launch of the process belongs to a whole step, for the counter passes)")
            s_v = strmux.settings(fmt=strmux.FORMAT_STRCD, codec=0, width=w, height=h, fps_num=15, fps_den=1, cd_speed=2, channels=0)
            dt_v, _ = timed_dev(s_v, d_frames, None, 20)
            pv = strmux.plan(s_v, n, 0)
            legs["video_only_all_sectors_video"] = {"ms_per_step": round(dt_v * 1e3, 4), "sectors_per_sec": round(pv.n_sectors * S / dt_v, 1),
                                                    "frames_per_sec": round(pv.n_frames_encoded * S / dt_v, 1),
                                                    "algorithmic_gbs": round(((w * h * 3 // 2) * pv.n_frames_encoded + pv.n_sectors * pv.sector_size) * S / dt_v / 1e9, 2)}
            if S > 1:
                dt_1, _ = timed_dev(s, d_frames[:1].contiguous(), d_pcm[:1].contiguous(), 20)
                legs["one_stream_per_call"] = {"ms_per_step": round(dt_1 * 1e3, 4), "sectors_per_sec": round(p.n_sectors / dt_1, 1),
                                               "note": "one stream's XA track alone: 2 chains, bound by how far a wrong start state travels"}
            else:       # value is ONE stream (BASELINE's config 3); what a caller with eight independent streams gets from one call
                S8 = 8
                f8 = torch.stack([d_frames[0]] + [synth.frames_device(w, h, args.seed + 17 * i, first, n, args.amp, device=local_rank) for i in range(1, S8)])
                p8 = torch.zeros((S8, na * 2), dtype=torch.int16, device=dev)
                p8[0] = d_pcm[0]
                for i in range(1, S8):
                    for c in range(2):
                        synth.pcm_device(args.seed, 2 * i + c, 0, na, args.audio_kind, device=local_rank, out=p8[i][c:], pitch=2)
                dt_8, _ = timed_dev(s, f8, p8, 20)
                legs["eight_streams_per_call"] = {"ms_per_step": round(dt_8 * 1e3, 4), "sectors_per_sec": round(p.n_sectors * S8 / dt_8, 1),
                                                  "note": "eight independent streams in one psxhip_str_encode_device call: their XA tracks share the verify passes"}
                del f8, p8
        except Exception as e:
            legs["error"] = repr(e)
        frames0 = d_frames[0].cpu().numpy()
        pcm0 = d_pcm[0].cpu().numpy()
        try:
            if args.steps_only:
                raise RuntimeError("skipped (--steps-only)")
            sectors = np.zeros((p.n_sectors, p.sector_size), np.uint8)
            for _ in range(3):
                strmux.encode(s, frames0, pcm0, device=local_rank, out=sectors)
            t = time.perf_counter()
            reps = 30
            for _ in range(reps):
                strmux.encode(s, frames0, pcm0, device=local_rank, out=sectors)
            dt_h = (time.perf_counter() - t) / reps
            legs["host_buffers_one_stream"] = {"ms_per_step": round(dt_h * 1e3, 4), "sectors_per_sec": round(p.n_sectors / dt_h, 1),
                                               "note": "psxhip_str_encode_host: PCIe in and out + host interleave inside the call (round 4's strcd figure)",
                                               "equals_device_path": bool(np.array_equal(sectors, d_out[0].cpu().numpy()))}
        except Exception as e:
            legs["host_buffers_one_stream"] = {"error": repr(e)}
        # parity on a prefix: the reference's sector loop (tests/str_reference_loop.py over the oracle) on the first frames and
        # their share of the audio -- a complete little stream with its own tail -- through the DEVICE path
        import str_reference_loop as R
        k = min(24, n)
        pk = pcm0[:2 * 2016 * 30]
        if args.steps_only:
            parity = None
        else:
            sub, _ = mux.encode_device(s, d_frames[0, :k].contiguous(), torch.from_numpy(pk).to(dev))
            sub = sub.cpu().numpy()[0]
            osub, _, _ = R.encode_file_str(7, 0, w, h, 15, 1, 2, frames0[:k], pk)
            parity = {"sectors_checked": int(osub.shape[0]), "bit_exact": bool(sub.shape == osub.shape and np.array_equal(sub, osub)),
                      "streams_equal_host_path": legs.get("host_buffers_one_stream", {}).get("equals_device_path")}
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            ref = os.path.join(ROOT, "oracle", "_ref", "libpsxav_ref.so")
            one = _run_cpu_bench(["str", 1, args.cpu_seconds * 0.6, args.seed] + ([ref] if os.path.exists(ref) else []))
            if one:
                nthr, cores, cores_note = _usable_cores()
                cpu_baseline = {"value": one["units_per_sec"], "unit": "sectors/s", "cores": 1, "kind": "port",
                                "sample": "%d sectors in %.1f s: the sector loop of encode_file_str in C (oracle/cpu_bench.c str: orc_mdec_encode_sector_str + %s, "
                                          "mode-2 form-1 headers and EDC; gcc -O3)" % (one["units"], one["seconds"], "the reference's own psx_audio_xa_encode" if os.path.exists(ref) else "orc_xa_encode")}
                many = _run_cpu_bench(["str", nthr, args.cpu_seconds * 0.4, args.seed] + ([ref] if os.path.exists(ref) else []))
                if many:
                    cpu_baseline["all_cores"] = {"value": many["units_per_sec"], "unit": "sectors/s", "cores": nthr, "nproc": cores, "cores_note": cores_note,
                                                 "speedup_vs_1_core": round(many["units_per_sec"] / max(one["units_per_sec"], 1e-9), 1)}
        total = p.n_sectors * S * world * args.steps
        alg = ((w * h * 3 // 2) * p2.n_frames_encoded + na * 4 + p.n_sectors * p.sector_size) * S
        version = _lib.lib().psxhip_version().decode()
        try:
            _lib.lib().psxhip_adpcm_kernel_rev.restype = __import__("ctypes").c_char_p
            arev = _lib.lib().psxhip_adpcm_kernel_rev().decode()
        except Exception:
            arev = "?"
        str_key = "strcd S=%d frames=%d kind=%d | %s, %s" % (S, n, args.audio_kind, version, arev)
        traffic, traffic_src, pmc = _profile_traffic(str_key)
        _emit({
            "metric": "strcd_v2_320x240_sectors_per_sec", "value": round(total / elapsed, 2), "unit": "sectors/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "timed_region_s": round(elapsed, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic", "dist": _dist_info(args, dist),
            "config": {"workload": "strcd v2: %d independent streams x %d frames 320x240 @15 fps + 37800 Hz 4-bit stereo XA per GPU per step -> %d sectors of 2352 "
                                   "bytes per stream (%d video, %d audio); frames and PCM resident in HBM, sectors land in HBM (psxhip_str_encode_device)"
                                   % (S, n, p.n_sectors, p.n_video_sectors, p.n_audio_sectors),
                       "preset": args.config, "baseline_config": args.baseline_config, "tail": "reference (filefmt.c:443-450,492-493)",
                       "streams_per_call": S, "frames_encoded_per_stream": p2.n_frames_encoded,
                       **({"eight_streams_sectors_per_sec": legs["eight_streams_per_call"]["sectors_per_sec"]} if "eight_streams_per_call" in legs else {}),
                       **({"one_stream_sectors_per_sec": legs["one_stream_per_call"]["sectors_per_sec"]} if "one_stream_per_call" in legs else {}),
                       "frames_per_sec": round(p2.n_frames_encoded * S * world * args.steps / elapsed, 1),
                       "realtime_factor": round(p2.n_frames_encoded * S * world * args.steps / elapsed / 15.0, 1),
                       "material": AUDIO_KINDS.get(args.audio_kind, str(args.audio_kind)),
                       "avg_quant_scale": round(p2.quant_scale_sum / max(1, p2.n_frames_encoded * S), 3),
                       "stream0_sha256": hashlib.sha256(d_out[0].cpu().numpy().tobytes()).hexdigest(), "library": version, "legs": legs},
            "roofline": {"bound": "hbm", "kernel": "whole step: mdec_encode_frames_kernel + str_video_sector_kernel beside adpcm_chunks_kernel (speculate, verify) + xa_assemble_kernel",
                         "achieved": round(alg * args.steps / elapsed / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_key": str_key, "algorithmic_bytes_per_launch": alg,
                         "traffic_note": "per STEP: the counters of every kernel of a step added up (tools/gpu_r05_strcd_pmc.sh, tools/make_strcd_profile_summary.py); the frame kernel "
                                         "on this budget cycle alone: profiles/pmc_index.json key 'sbs codec=0 320x240 budget=cycle(16128,18144,18144,18144) ...'",
                         "per_kernel_counters": (pmc or {}).get("kernels"),
                         "video_leg_achieved": legs.get("video_only_all_sectors_video", {}).get("algorithmic_gbs"),
                         "note": "device-resident; the step's time is the XA tracks' verify passes (the tonal test signal), not bytes"},
            "cpu_baseline": cpu_baseline, "parity": parity,
            "per_rank": [{"rank": i, "sectors_per_sec": round(r[1] / r[0], 1), "elapsed_s": round(r[0], 4)} for i, r in enumerate(per_rank)]}, args)
    mux.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
