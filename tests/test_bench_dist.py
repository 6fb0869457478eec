"""bench.py's distributed leg and failure reporting.

`-m gpu`: the RCCL leg at world size 1 -- `torch.distributed.run --nproc-per-node 1` exactly as the driver starts N ranks,
`init_process_group("nccl", device_id=...)`, `barrier(device_ids=...)`, the float64 all_gather of device tensors -- for all
four BASELINE presets, so that the first 8-GPU run is not the first time that code executes on hardware (the sharded loop is
the reference's per-file loop, psxavenc/filefmt.c:633-662).
`-m "not gpu"`: a failing rank prints ONE JSON line on stdout and exits non-zero."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _last_json(stdout):
    for ln in reversed(stdout.strip().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


PRESET_ARGS = {
    "sbs_v2": ["--launches-per-step", "20"],
    "sbs_v3": ["--total-frames", "1250", "--launches-per-step", "4"],
    "xacd": ["--audio-seconds", "120"],
    "strcd": ["--frames", "120"],
}


@pytest.mark.gpu
@pytest.mark.parametrize("preset", sorted(PRESET_ARGS))
def test_rccl_leg_at_world_size_1_under_torchrun(preset):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--config", preset,
           "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"] + PRESET_ARGS[preset]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = _last_json(r.stdout)
    assert r.returncode == 0 and line is not None, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "error" not in line, line
    assert line["n_gpus"] == 1 and line["steps"] == 2
    assert line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 1 and line["dist"]["forced_at_world_size_1"]
    assert line["dist"]["exchange_tensors_on"] == "device"
    assert line["parity"]["bit_exact"] is True
    assert line["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("preset", sorted(PRESET_ARGS))
def test_two_ranks_sharing_the_one_gpu_all_four_presets(preset):
    """World size 2 on hardware, with real sessions and real contexts (these were kept logs, not tests): two ranks
    started by bench.py itself through torch.distributed.run, gloo for the exchange (RCCL wants a device per rank), both on GPU 0 --
    the rank arithmetic (contiguous shares, strong / weak scaling), the gathered per-rank counters, the max-over-ranks time and, for
    xacd, the time-shard protocol (final states all-gathered, second rank re-verified from its predecessor's truth)."""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--config", preset, "--steps", "2", "--warmup", "1",
           "--no-secondary", "--no-cpu-baseline"] + PRESET_ARGS[preset]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    line = _last_json(r.stdout)
    assert r.returncode == 0 and line is not None, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "error" not in line, line
    assert line["n_gpus"] == 2 and line["dist"]["backend"] == "gloo" and line["dist"]["world_size"] == 2
    assert line["parity"]["bit_exact"] is True and line["value"] > 0
    if preset in ("sbs_v2", "sbs_v3"):
        ranks = line["per_rank"]
        assert [x["rank"] for x in ranks] == [0, 1] and all(x["results_sane"] for x in ranks)
        if preset == "sbs_v3":          # strong scaling: the job's frames are split
            assert line["scaling"] == "strong" and line["config"]["frames_per_gpu_per_launch"] == 625
        else:
            assert line["scaling"] == "weak" and line["config"]["frames_per_gpu_per_launch"] == 1000
    if preset == "xacd":
        assert "time-sharded x2" in line["config"]["workload"]


@pytest.mark.gpu
def test_forced_group_without_torchrun_sets_up_its_own_rendezvous():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "1", "--launches-per-step", "8",
                        "--no-secondary", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = _last_json(r.stdout)
    assert r.returncode == 0 and line and line["dist"]["backend"] == "nccl", (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_c_abi_device_list_leg_of_the_bench_line():
    """the leg a single process runs over ALL visible devices of a multi-GPU node (psxhip_mdec_multi_* with one host thread, context
    and pinned staging pair per device), forced here onto the list 0,0: both schedules byte-equal to device 0 alone"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "1", "--launches-per-step", "8", "--no-cpu-baseline",
                        "--no-config-secondaries", "--device-list", "0,0"], env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    line = _last_json(r.stdout)
    assert r.returncode == 0 and line, (r.stdout[-2000:], r.stderr[-2000:])
    assert "secondary" not in line          # stdout carries the compact line; the legs are in the detail file it names
    with open(os.path.join(ROOT, line["detail_file"])) as fh:
        leg = json.load(fh)["secondary"]["c_abi_device_list"]
    assert "error" not in leg, leg
    assert leg["devices"] == [0, 0] and leg["static"]["bit_exact_vs_device_0"] and leg["tickets"]["bit_exact_vs_device_0"]
    assert sum(d["units"] for d in leg["tickets"]["per_device"]) == leg["frames"]


@pytest.mark.gpu
def test_bad_geometry_is_one_json_error_line_on_the_gpu_box():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--width", "328", "--steps", "1", "--warmup", "0", "--no-secondary",
                        "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = _last_json(r.stdout)
    assert r.returncode != 0 and line and "error" in line and line["rank"] == 0, (r.stdout[-2000:], r.stderr[-2000:])


def test_a_failing_rank_prints_one_json_error_line():
    """(CPU container: no GPU -> the product path refuses; on a GPU box the world-size mismatch does the failing)"""
    env = _env()
    env.update(RANK="1", WORLD_SIZE="3", LOCAL_RANK="1")       # --gpus 2 against WORLD_SIZE 3
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    line = _last_json(r.stdout)
    assert r.returncode != 0
    assert line is not None and "error" in line, (r.stdout[-1000:], r.stderr[-1000:])
    assert line["rank"] == 1 and line["world_size"] == 3
    assert len(r.stdout.strip().splitlines()) == 1


REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline", "parity")


@pytest.mark.gpu
def test_the_driver_command_prints_one_parsable_line_under_8_kb():
    """`python bench.py --gpus 1 --steps K --warmup W` exactly as the driver runs it (all children included): the LAST line of stdout is
    the record, it parses, it is under 8 KB (the driver keeps an 8 KB tail; round 5's 36 KB line was lost), and it carries `roofline`
    and `cpu_baseline` for the headline and every BASELINE config as a {value, unit, frac, bit_exact} child."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-seconds", "3"], env=_env(), capture_output=True,
                       text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 8192 and len(r.stdout) < 8192, (len(last), len(r.stdout))
    line = json.loads(last)
    for k in REQUIRED_KEYS:
        assert k in line, k
    assert line["metric"] == "bs_v2_320x240_frames_per_sec" and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert roof["traffic"] is None or roof["traffic"] > 0
    assert cpu["value"] > 0 and cpu["cores"] == 1 and cpu["kind"] in ("port", "reference")
    assert line["parity"]["bit_exact"] is True
    summ = line["config"]["secondary_summary"]
    for child in ("sbs_v3_1250", "xacd_config5", "strcd_config3", "rccl_world_size_1"):
        assert summ[child]["value"] > 0 and summ[child]["bit_exact"] is True, (child, summ[child])
        assert len(json.dumps(summ[child])) <= 150
    assert summ["strcd_config3"]["s8"] > summ["strcd_config3"]["value"]          # value = ONE stream (BASELINE's config 3), eight per call beside it
    with open(os.path.join(ROOT, line["detail_file"])) as fh:
        full = json.load(fh)
    assert full["secondary"]["strcd_config3"]["config"]["streams_per_call"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["sbs_v2", "xacd"])
def test_two_ranks_on_two_distinct_gpus_over_rccl(preset):
    """RCCL at world size 2 over DISTINCT devices -- only where the lease has two GPUs (skipped on the one-GPU boxes); the line carries
    per-rank rates, the world size RCCL reports and the prediction DESIGN section 5 commits to."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    cmd = [sys.executable, BENCH, "--gpus", "2", "--config", preset, "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"] + PRESET_ARGS[preset]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    line = _last_json(r.stdout)
    assert r.returncode == 0 and line is not None and "error" not in line, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert line["n_gpus"] == 2 and line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 2
    assert [x["rank"] for x in line["per_rank"]] == [0, 1] and line["parity"]["bit_exact"] is True
    assert line["predicted"]["value"] > 0
