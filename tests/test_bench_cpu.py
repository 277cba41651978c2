"""bench.py host logic that needs no GPU: how `--gpus N` resolves into ranks (the
argument used to be parsed and ignored, so `--gpus 8` printed a 1-GPU line), and the workload table
against SURVEY.md section 8d."""
import math
import os
import subprocess
import sys

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_runs_in_process():
    assert bench.resolve_world(1, {}, 1) == ("run", 1, 0, 0)
    assert bench.resolve_world(1, {}, 8) == ("run", 1, 0, 0)


def test_multi_gpu_without_launcher_respawns():
    assert bench.resolve_world(8, {}, 8) == ("spawn", 8)
    assert bench.resolve_world(2, {"HOME": "/root"}, 4) == ("spawn", 2)


def test_more_gpus_than_visible_is_refused():
    for gpus, visible in ((2, 1), (8, 4), (1, 0)):
        with pytest.raises(SystemExit) as e:
            bench.resolve_world(gpus, {}, visible)
        assert e.value.code not in (0, None) and "visible" in str(e.value.code)


def test_under_the_launcher_the_world_must_match():
    env = {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"}
    assert bench.resolve_world(8, env, 8) == ("run", 8, 3, 3)
    with pytest.raises(SystemExit):
        bench.resolve_world(4, env, 8)          # --gpus disagrees with the launcher
    with pytest.raises(SystemExit):
        bench.resolve_world(8, env, 2)          # local rank 3 has no GPU
    assert bench.resolve_world(1, {"RANK": "0", "WORLD_SIZE": "1"}, 1) == ("run", 1, 0, 0)


def test_gpus_flag_fails_loudly_without_gpus():
    """On this GPU-less box `python bench.py --gpus 2` must exit non-zero with a message, not print a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "visible" in (r.stderr + r.stdout) and '"metric"' not in r.stdout


def test_workload_table_matches_survey_8d():
    per_gpu = {"cfg2": (231_211_008, 553_396_224, 355.1e9), "cfg3": (1_849_688_064 // 8, 2_575_545_344 // 8, 88.8e9 / 8),
               "cfg4": (452_984_832, 591_675_904, 174.0e9), "cfg5": (7_247_757_312 // 4, 4_045_963_776 // 4, 5566e9 / 4)}
    for name, (ns, nbytes, flop) in per_gpu.items():
        c = bench.WORKLOADS[name]
        K = 3 ** c["nd"]
        assert c["B"] * c["C"] * K * math.prod(c["sp"]) == ns
        assert c["bytes"] == nbytes
        gemm = 2.0 * c["O"] * (c["C"] // c["G"]) * K * c["B"] * math.prod(c["sp"])
        assert abs(3 * gemm - flop) / flop < 2e-3


def _stub_bench(*flags):
    """bench.py end to end with the host-only stub workload (MDCONV_BENCH_STUB=1: sleeps instead of kernels, gloo
    instead of RCCL): the launcher, rank / shard arithmetic, barrier + MAX-over-ranks timing and the JSON line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MDCONV_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                        "--sustain-s", "0.05", "--no-cpu-baseline"] + list(flags),
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_launch_shards_and_reports_whole_job_throughput(scaling):
    """`--gpus 2` without a launcher re-executes under torch.distributed.run with two ranks (here on CPU, gloo);
    weak: B = 32 per rank, global batch 64; strong: global batch 32 in shards of 16; `value`
    is the whole job's samples over the MAX-over-ranks time."""
    d = _stub_bench("--gpus", "2", "--scaling", scaling)
    per_image = 256 * 9 * 56 * 56
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 4 and d["warmup"] == 1
    assert d["config"]["global_batch"] == (64 if scaling == "weak" else 32)
    assert ("B=%d per GPU" % (32 if scaling == "weak" else 16)) in d["config"]["workload"]
    assert d["config"]["parallelism"] == "dp2 batch-sharded"
    want = d["config"]["global_batch"] * per_image / (d["ms_per_step"] * 1e-3) / 1e9
    assert abs(d["value"] - want) / want < 2e-3
    assert d["sustained_ms_per_step"] > 0 and d["sustained_steps"] >= 4
    # the stub sleeps 0.3 ms per image: strong-scaling shards must be about twice as fast as the weak ones
    assert (4.0 < d["ms_per_step"] < 40.0) if scaling == "strong" else (9.0 < d["ms_per_step"] < 60.0)


def test_single_rank_stub_line_has_the_contract_fields():
    d = _stub_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "sustained_ms_per_step"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 32 and d["vs_baseline"] is None
    assert d["data"].startswith("STUB") and d["config"]["kernel_path"] == "stub"   # can never pass for a measurement
