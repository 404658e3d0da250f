"""bench.py's output contract on a small slice of the workload: stdout is exactly ONE JSON line with the fields the
driver reads, at N=1 and through the multi-rank code path (RCCL process group, device-resident records, gather) --
run here with the one rank a one-GPU box has (`--dist-selftest`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def run_bench(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chroms", "20,21,22", "--steps", "3", "--warmup", "1",
                        "--no-dense-leg"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, "stdout must be one JSON line, got %d lines: %r" % (len(lines), r.stdout[:500])
    out = json.loads(lines[0])
    for k in REQUIRED:
        assert k in out, k
    assert out["steps"] == 3 and out["warmup"] == 1 and out["n_gpus"] == 1 and out["vs_baseline"] is None
    assert out["unit"] == "intervals/s" and out["higher_is_better"] is True and out["data"] == "synthetic"
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "workload" in out["config"] and "model" not in out["config"]
    return out


@pytest.mark.gpu
def test_one_json_line_with_cpu_baseline():
    out = run_bench(["--cpu-seconds", "1"])
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "intervals/s" and cb["value"] > 0 and cb["sample"]
    assert out["cn_concordance"] == 1.0
    # the legs for the other single-GPU configurations ride in the same line, and the BAM route's inflate kernel with them
    legs = out["configs"]
    assert set(legs) == {"configs[2]", "configs[4]", "bgzf_inflate"}
    assert legs["configs[2]"]["cn_concordance"] == 1.0 and legs["configs[4]"]["records_per_s"] > 0
    # ... and the end-to-end leg (BAM files written on the spot -> `conga --cohort`), both decoders, its own CPU baseline
    e2e = out["end_to_end"]
    assert e2e["end_to_end"]["decode"] == "gpu" and e2e["end_to_end"]["per_further_sample_ms"] > 0 and e2e["end_to_end_host_decoders"]["first_sample_s"] > 0
    assert e2e["cpu_baseline"]["host_cores"] >= 1 and "checked" in e2e
    assert legs["configs[4]"]["end_to_end"]["gpu_decode_calls_per_sample"] >= 1 and "checked" in legs["configs[4]"]
    assert out["three_contexts"]["ms_per_step"] > 0 and out["hand_over_int32"]["ms_per_step"] > 0
    bz = legs["bgzf_inflate"]
    assert bz["unit"] == "GB/s inflated" and bz["value"] > 5 and bz["blocks"] > 16000 and bz["cpu_baseline"]["kind"] == "zlib"


@pytest.mark.gpu
def test_multi_rank_code_path_under_rccl_with_one_rank():
    out = run_bench(["--dist-selftest", "--cpu-seconds", "0"])
    assert "RCCL gather" in out["config"]["parallelism"]
