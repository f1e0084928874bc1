"""bench.py's ONE stdout line must stay small: the driver keeps an 8 KB tail of stdout, and round 2 lost its whole record
to a 38 KB line (a 122-entry per-site table).  The per-site table goes to a side file; the line holds the headline
fields, the dominant site and short secondaries."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _synthetic(n_sites):
    sites = [{"site": f"gemm[N{128 + i},K{384 + i},<3,1,0,2,1,6> with a long template tail]", "kernel": "gemm_kernel<3,1,0,2,1,6>",
              "bound": "mfma", "achieved": 61.234, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.38927, "traffic": 74000000,
              "launches": 144, "avg_us": 41.7, "work_per_launch": 2.77e9, "total_ms": 6.0 - 0.01 * i, "us_per_step": 250.0,
              "launches_per_step": 6.0} for i in range(n_sites)]
    out = {"metric": "crystals/sec training throughput (Phonon DOS, hidden=128)", "value": 50400.12, "unit": "crystals/s",
           "n_gpus": 1, "steps": 200, "warmup": 20, "ms_per_step": 1.2707, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "phonon_h128_b64: phonon DOSTransformer layers=3 transformer=2 hidden=128, 64 crystals/GPU, "
                                  "full train step (fwd+loss+bwd+AdamW), 8 distinct pre-collated HBM-resident batches",
                      "global_batch": 64, "parallelism": "dp1", "launch": "replay", "bucket": [8, 128]},
           "step_frac": 0.256, "step_gflop": 51.15, "host_ms_per_step": 0.9,
           "roofline": {k: sites[0][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "site", "kernel")},
           "traffic_source": "r03_pmc_traffic.json (git 0123456789ab)",
           "slots": {"hits": 200, "misses": 0, "hit_rate": 1.0, "live": 8, "max": 32},
           "secondary": {"edos_h256_b64": {"value": 7840.0, "unit": "crystals/s", "ms_per_step": 8.17, "steps": 40, "step_frac": 0.5,
                                           "host_ms_per_step": 1.2},
                         "shuffle": {"value": 46200.0, "unit": "crystals/s", "ms_per_step": 1.384, "steps": 200, "step_frac": 0.23,
                                     "host_ms_per_step": 1.1, "hit_rate": 0.99, "live_buckets": 4}},
           "cpu_baseline": {"value": 279.0, "unit": "crystals/s", "cores": 16, "kind": "port", "value_f32": 522.0,
                            "value_ref_threads": 92.0, "ref_threads": 2, "host": "AMD EPYC 9575F 64-Core Processor, 256 logical CPUs",
                            "sample": "62 full train steps of one 64-crystal batch, fp64, 15.1s, 243.5 ms/step, fastest of 2/8/16/32 threads"},
           "kernels_file": "bench_kernels_last.json"}
    return out, sites


def test_record_line_stays_small_with_200_sites():
    import bench
    out, sites = _synthetic(200)
    rec = bench.compact_record(out, sites)
    line = json.dumps(rec)
    assert len(line) < 4000, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "step_frac"):
        assert k in rec, k
    assert "kernels" not in rec and rec["roofline"]["frac"] == sites[0]["frac"]
    assert rec["config"]["workload"].startswith("phonon_h128_b64")
    assert len(rec.get("top_sites", [])) <= 5


def test_record_drops_optional_fields_before_growing():
    import bench
    out, sites = _synthetic(200)
    out["traffic_source"] = "x" * 5000           # something silly on an optional field must not grow the line
    rec = bench.compact_record(out, sites)
    assert len(json.dumps(rec)) < 4000 and "value" in rec and "roofline" in rec and "cpu_baseline" in rec


@pytest.mark.gpu
def test_bench_default_line_is_one_small_json_record(tmp_path):
    """The driver's own invocation shape (no flags except fewer steps): exactly one stdout line, < 4000 bytes, parseable,
    with the roofline object, the cpu baseline and the secondaries; the per-site table is in the side file."""
    kout = tmp_path / "kernels.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "8", "--cpu-budget", "2",
                        "--kernels-out", str(kout)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = p.stdout.decode().strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4000, (len(lines), len(lines[0]))
    rec = json.loads(lines[0])
    assert rec["value"] > 0 and rec["n_gpus"] == 1 and rec["dtype"] == "f32"
    assert abs(rec["value"] - 64 / (rec["ms_per_step"] * 1e-3)) < 1e-2 * rec["value"]
    r = rec["roofline"]
    assert r["bound"] in ("mfma", "hbm") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 0
    assert "error" not in rec.get("secondary", {}), rec["secondary"]
    assert rec["secondary"]["edos_h256_b64"]["value"] > 0 and rec["secondary"]["shuffle"]["value"] > 0
    assert rec["host_ms_per_step"] > 0
    table = json.loads(kout.read_text())
    assert len(table["sites"]) >= 10 and not any(s["site"].startswith("gemm[M") for s in table["sites"])
