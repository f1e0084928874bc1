"""bench.py's ONE stdout line must stay small: the driver keeps an 8 KB tail of stdout (the line is held below 6000 bytes), and
round 2 lost its whole record to a 38 KB line (a 122-entry per-site table).  The per-site table goes to a side file; the line holds the headline
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
    # round 4: the counter-based north_star figures, the data-parallel secondaries, the configs[4] shard
    out["prepare_steps"] = 16
    out["north_star"] = {"scatter_hbm_frac": {"cfg2": 0.29, "4Mi": 0.679},
                         "attn_mfma_util": {"cfg2_cross": 0.0838, "cfg2_self": 0.1589, "edos_cross": 0.22, "edos_self": 0.3352},
                         "source": "r04_north_star.json (rocprofv3 --pmc, git 182ada2ca2c9)"}
    out["secondary"]["edos_h256_t4_b32"] = {"value": 4194.0, "unit": "crystals/s", "ms_per_step": 7.6291, "steps": 30, "step_frac": 0.455,
                                            "host_ms_per_step": 0.68}
    out["secondary"]["dp1_nccl"] = {"value": 49064.0, "unit": "crystals/s", "ms_per_step": 1.3044, "steps": 200, "step_frac": 0.249,
                                    "host_ms_per_step": 0.37, "grad_bucket_bytes": {"early": 3501056, "mid": 0, "late": 3037696},
                                    "exposed_bytes": 3037696, "collectives_per_step": 3, "backend": "nccl"}
    return out, sites


def test_record_line_stays_small_with_200_sites():
    import bench
    out, sites = _synthetic(200)
    rec = bench.compact_record(out, sites)
    line = json.dumps(rec)
    assert len(line) < 6000, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "step_frac"):
        assert k in rec, k
    assert "kernels" not in rec and rec["roofline"]["frac"] == sites[0]["frac"]
    assert "north_star" in rec and "dp1_nccl" in rec["secondary"] and "top_sites" in rec       # nothing had to be dropped
    assert rec["config"]["workload"].startswith("phonon_h128_b64")
    assert len(rec.get("top_sites", [])) <= 5


def test_record_drops_optional_fields_before_growing():
    import bench
    out, sites = _synthetic(200)
    out["traffic_source"] = "x" * 5000           # something silly on an optional field must not grow the line
    rec = bench.compact_record(out, sites)
    assert len(json.dumps(rec)) < 6000 and "value" in rec and "roofline" in rec and "cpu_baseline" in rec


_RANK_SCRIPT = """
import json, os, sys, time
r = int(os.environ["RANK"])
json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                          "HSA_ENABLE_IPC_MODE_LEGACY")} | {"argv": sys.argv[1:]},
          open(os.path.join(sys.argv[1], f"rank{r}.json"), "w"))
mode = sys.argv[2]
if r == 0:
    print("library banner on stdout")
    print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "value": 1.0}))
if mode == "fail" and r == 1:
    sys.exit(3)
if mode == "fail" and r != 1:
    time.sleep(120)            # waits 'in a collective' for the dead rank
"""


def test_self_launch_starts_one_process_per_rank_and_relays_rank0(tmp_path, capfd):
    """`python bench.py --gpus N` without a launcher: N child processes with the environment torch.distributed.run would
    set, the command line passed through, rank 0's LAST stdout line relayed, exit code 0."""
    import bench
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rc = bench.self_launch(3, [str(tmp_path), "ok", "--gpus", "3"], script=str(script))
    assert rc == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == {"n_gpus": 3, "value": 1.0}
    envs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] is not None for e in envs)
    assert all(e["argv"] == [str(tmp_path), "ok", "--gpus", "3"] for e in envs)


def test_self_launch_returns_the_failing_ranks_code_and_ends_the_others(tmp_path):
    import time
    import bench
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    t0 = time.monotonic()
    rc = bench.self_launch(2, [str(tmp_path), "fail"], script=str(script), grace_s=1.0)
    assert rc == 3 and time.monotonic() - t0 < 60


def test_bare_multi_gpu_invocation_becomes_the_launcher_before_touching_the_gpu(monkeypatch):
    """No WORLD_SIZE + --gpus 2: main() hands over to self_launch with the untouched command line and exits with its code;
    nothing GPU-side may run in that process (`python3 bench.py --gpus 8` used to exit 1)."""
    import bench
    import torch
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--share-gpu"])
    monkeypatch.setattr(bench, "self_launch", lambda n, argv, **kw: seen.update(n=n, argv=list(argv)) or 7)

    def boom(*a, **k):
        raise AssertionError("the launcher process touched the GPU")
    for name in ("set_device", "synchronize", "current_stream", "is_available", "init"):
        monkeypatch.setattr(torch.cuda, name, boom)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7 and seen == {"n": 2, "argv": ["--gpus", "2", "--steps", "6", "--warmup", "2", "--share-gpu"]}


def test_bench_refuses_tuning_switches_in_the_environment(monkeypatch):
    """a DOSX_* variable in the environment aborts bench.py before it touches the GPU, unless --allow-env;
    the effective switches are what the record's `env` field holds."""
    import bench
    import torch

    def boom(*a, **k):
        raise AssertionError("bench.py touched the GPU before refusing the environment")
    for name in ("set_device", "synchronize", "current_stream", "is_available", "init"):
        monkeypatch.setattr(torch.cuda, name, boom)
    for k in [k for k in os.environ if k.startswith("DOSX_")]:
        monkeypatch.delenv(k)
    assert bench.tuning_env() == {} and bench.refuse_tuning_env([]) == {}
    monkeypatch.setenv("DOSX_WGRAD_MAXSPLIT", "4")
    monkeypatch.setenv("DOSX_LIB", "/tmp/other.so")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "4"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert "DOSX_WGRAD_MAXSPLIT=4" in str(ei.value.code) and "DOSX_LIB" in str(ei.value.code) and "--allow-env" in str(ei.value.code)
    assert bench.refuse_tuning_env(["--steps", "4", "--allow-env"]) == {"DOSX_LIB": "/tmp/other.so", "DOSX_WGRAD_MAXSPLIT": "4"}


def test_bench_refusal_is_the_process_exit_code():
    """... as a process: non-zero exit, the reason on stderr, nothing on stdout."""
    env = dict(os.environ, DOSX_FUSED_ATT_BWD="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and p.stdout == b"" and b"DOSX_FUSED_ATT_BWD=0" in p.stderr


def test_no_work_skipping_switch_ships_in_the_package():
    """timing-experiment switches that leave launches out live in the stamps build / tools only."""
    import glob
    import re
    pat = re.compile(r"DOSX_DEBUG_SKIP|environ[^\n]*SKIP")
    for f in glob.glob(os.path.join(ROOT, "dostransformer_amd", "**", "*.py"), recursive=True) + [os.path.join(ROOT, "bench.py")]:
        assert not pat.search(open(f).read()), f
    for f in glob.glob(os.path.join(ROOT, "dostransformer_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "dostransformer_amd", "csrc", "*.cpp")):
        src = open(f).read()
        for m in re.finditer(r"DOSX_DEBUG_SKIP", src):
            guard = src.rfind("#ifdef DOSX_STAMPS", 0, m.start())
            assert guard >= 0 and src.find("#endif", guard) > m.start(), f"{f}: DOSX_DEBUG_SKIP outside the stamps build"


def test_rccl_tuning_lines_are_parsed(tmp_path):
    import bench
    f = tmp_path / "t.log"
    f.write_text("h:1:1 [0] NCCL INFO 3407872 Bytes -> Algo 1 proto 2 time 45.0\n"
                 "h:1:1 [0] NCCL INFO AllReduce: 3116032 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..31}\n"
                 "h:1:1 [0] NCCL INFO 3116032 Bytes -> Algo RING proto SIMPLE\nnoise\n")
    got = bench.rccl_choices(str(f), 8)
    assert [(e["bytes"], e["algo"], e["proto"], e["n"]) for e in got] == [(3407872, "1", "2", 1), (3116032, "RING", "SIMPLE", 2)]
    assert isinstance(bench.rccl_choices(None, 8), str) and isinstance(bench.rccl_choices(str(f), 1), str)


@pytest.mark.gpu
def test_bare_two_rank_bench_launches_itself(tmp_path):
    """VERDICT r3 item 1's acceptance line, from a clean environment (no RANK / WORLD_SIZE): one JSON line, rc 0, n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu",
                        "--steps", "6", "--warmup", "2", "--config", "phonon_h64_b8", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = p.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["warmup"] == 2 and rec["steps"] == 6
    assert rec["dp"]["collectives_per_step"] == 3 and rec["dp"]["staged_through_host"] is True      # sse, early, late
    assert rec["dp"]["plan"] == ["prog", "sse", "prog", "early", "prog", "late", "adamw"]
    assert 0 < rec["dp"]["exposed_bytes"] == rec["dp"]["grad_bucket_bytes"]["late"] and rec["dp"]["grad_bucket_bytes"]["mid"] == 0


@pytest.mark.gpu
def test_bench_default_line_is_one_small_json_record(tmp_path):
    """The driver's own invocation shape (no flags except fewer steps): exactly one stdout line, < 4000 bytes, parseable,
    with the roofline object, the cpu baseline and the secondaries; the per-site table is in the side file (< 6000 bytes: the driver
    keeps an 8 KB tail)."""
    kout = tmp_path / "kernels.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "8", "--cpu-budget", "2",
                        "--kernels-out", str(kout)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = p.stdout.decode().strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 6000, (len(lines), len(lines[0]))
    rec = json.loads(lines[0])
    assert rec["value"] > 0 and rec["n_gpus"] == 1 and rec["dtype"] == "f32"
    assert abs(rec["value"] - 64 / (rec["ms_per_step"] * 1e-3)) < 1e-2 * rec["value"]
    r = rec["roofline"]
    assert r["bound"] in ("mfma", "hbm") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 0
    assert "error" not in rec.get("secondary", {}), rec["secondary"]
    assert rec["secondary"]["edos_h256_b64"]["value"] > 0 and rec["secondary"]["shuffle"]["value"] > 0
    assert rec["secondary"]["edos_h256_t4_b32"]["value"] > 0
    ns = rec["north_star"]                                # figures only when the committed PMC file matches the sources
    assert ns["source"] and (ns["scatter_hbm_frac"] is None or 0 < ns["scatter_hbm_frac"]["cfg2"] < 1)
    assert rec["host_ms_per_step"] > 0
    assert rec["warmup"] == 8 and rec["prepare_steps"] == 16          # the driver's consistency check: warm-up as requested
    ck = rec["check"]
    assert rec["env"] == {} and ck["finite"] is True and ck["replay_eq_eager"] is True, (rec["env"], ck)
    assert 0 < ck["loss_last"] < ck["loss_first"] * 1.5 and ck["params_changed_frac"] > 0.5, ck
    d1 = rec["secondary"]["dp1_nccl"]                                 # the data-parallel step on a 1-rank RCCL group
    assert "error" not in d1 and d1["value"] > 0 and d1["collectives_per_step"] == 3 and d1["backend"] == "nccl"
    assert d1["grad_bucket_bytes"]["early"] > 0 and d1["grad_bucket_bytes"]["late"] > 0 and d1["grad_bucket_bytes"]["mid"] == 0
    assert d1["exposed_bytes"] == d1["grad_bucket_bytes"]["late"]
    ev = rec["secondary"]["eval_per_crystal_b64"]                     # >= 50 k crystals/s at B = 64
    assert "error" not in ev and ev["value"] >= 50000 and ev["value"] > 10 * ev["batch1_loop"], ev
    sb = rec["secondary"]["split_bf16"]                               # secondary only, dtype spelled out,
    assert "error" not in sb and "bf16 x 3" in sb["dtype"]            # held to the exact-fp32 kernel's own error level
    for k in ("fc1_fwd", "fc2_dgrad"):
        assert sb[k]["us"] > 0 and sb[k]["speedup"] > 0.9 and sb[k]["err"] <= max(2e-5, 4 * sb[k]["err_fp32"]), sb[k]
    table = json.loads(kout.read_text())
    assert len(table["sites"]) >= 10 and not any(s["site"].startswith("gemm[M") for s in table["sites"])


def test_north_star_tool_classifies_the_attention_launches(tmp_path):
    """tools/pmc_north_star.py: the shape classes are told apart by (template arguments, grid size); utilisation = MFMA-busy
    cycles of the 1024 SIMDs / (1024 x kernel cycles at 2.4 GHz); the scatter rows by grid / traffic."""
    import csv
    rows = []

    def add(did, name, grid, ns, busy):
        rows.append({"Dispatch_Id": did, "Kernel_Name": name, "Grid_Size": grid, "Start_Timestamp": 1000, "End_Timestamp": 1000 + ns,
                     "Counter_Name": "SQ_VALU_MFMA_BUSY_CYCLES", "Counter_Value": busy})
    ns = 10000                                             # 10 us = 24 000 cycles at 2.4 GHz
    add(1, "void (anonymous namespace)::attn_fwd_stream_kernel<1, true>(DosxAttn)", 2 * 64 * 512, ns, 0.10 * 1024 * 24000)
    add(2, "void (anonymous namespace)::attn_bwd_dq_stream_kernel<4, true, true>(DosxAttn)", 2 * 128 * 512, ns, 0.20 * 1024 * 24000)
    add(3, "void (anonymous namespace)::attn_fwd_stream_kernel<3, true>(DosxAttn)", 7 * 128 * 512, ns, 0.25 * 1024 * 24000)
    add(4, "void (anonymous namespace)::attn_fwd_stream_kernel<13, false>(DosxAttn)", 7 * 128 * 512, ns, 0.40 * 1024 * 24000)
    add(5, "void (anonymous namespace)::attn_fwd_stream_kernel<13, false>(DosxAttn)", 7 * 2048 * 512, ns, 0.50 * 1024 * 24000)   # roofline scale: left out
    add(6, "void (anonymous namespace)::attn_bwd_dkv_kernel<2>(DosxAttn)", 4 * 128 * 512, ns, 0.50 * 1024 * 24000)
    f = tmp_path / "cc.csv"
    with open(f, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    sc = tmp_path / "sc.csv"
    sc.write_text("kernel,grid_size,launches,avg_us,HBM_MB_per_launch,GB_per_s,pct_of_8TBs\n"
                  "segment_reduce,28800,110,4.14,9.53,2300.1,28.8\nsegment_reduce,6710912,55,1204.04,6553.23,5442.7,68.0\n"
                  "segment_reduce,13421824,110,744.62,4409.12,5921.3,74.0\n")
    out = tmp_path / "ns.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_north_star.py"), str(f), str(sc), str(out), "abc"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    rec = json.loads(out.read_text())
    assert rec["scatter_hbm_frac"] == {"cfg2": 0.288, "4Mi": 0.68}
    u = rec["attn_mfma_util"]
    assert abs(u["cfg2_cross"] - 0.10) < 1e-3 and abs(u["cfg2_self"] - 0.20) < 1e-3 and abs(u["edos_cross"] - 0.25) < 1e-3
    assert abs(u["edos_self"] - 0.45) < 1e-3                 # forward + dK/dV kernels of the class together, roofline scale left out
    from dostransformer_amd._lib import source_hash
    assert rec["source_hash"] == source_hash() and rec["git_head"] == "abc"


def test_in_step_counter_tool_summarises_the_replayed_steps(tmp_path):
    """tools/pmc_step.py: MFMA-busy per kernel symbol of a counter pass over bench.py itself, the launches that
    contain the attention picked out by symbol, bytes / time of the launch that contains the scatter-add from the traffic file."""
    import csv
    rows = []

    def add(did, name, ns, busy, gui):
        for cn, cv in (("SQ_VALU_MFMA_BUSY_CYCLES", busy), ("GRBM_GUI_ACTIVE", gui)):
            rows.append({"Dispatch_Id": did, "Kernel_Name": name, "Grid_Size": 131072, "Start_Timestamp": 1000, "End_Timestamp": 1000 + ns,
                         "Counter_Name": cn, "Counter_Value": cv})
    ns = 30000                                             # 30 us = 72 000 cycles at 2.4 GHz
    cyc = ns * 2.4
    add(1, "void (anonymous namespace)::ffn_fwd_kernel<false, 64, 2>(DosxFfn)", ns, 0.30 * 1024 * cyc, 8 * cyc)
    add(2, "void (anonymous namespace)::ffn_fwd_kernel<true, 64, 1>(DosxFfn)", ns, 0.10 * 1024 * cyc, 8 * cyc)
    add(3, "void (anonymous namespace)::ffn_fwd_kernel<false, 64, 0>(DosxFfn)", ns, 0.50 * 1024 * cyc, 8 * cyc)      # no attention inside
    add(4, "void (anonymous namespace)::ffn_bwd_kernel<false, 64, 2>(DosxFfnBwd)", 2 * ns, 0.20 * 1024 * 2 * cyc, 16 * cyc)
    add(5, "void (anonymous namespace)::edge_fwd_kernel<256>(DosxEdgeMlp)", ns, 0.25 * 1024 * cyc, 8 * cyc)
    f = tmp_path / "cc.csv"
    with open(f, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    tr = tmp_path / "traffic.json"
    tr.write_text(json.dumps({"kernels": {"edge_fwd_kernel<256>": {"launches": 1, "hbm_bytes_per_launch": 24_000_000}}}))
    out = tmp_path / "step.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_step.py"), str(out), "abc", str(f), str(tr)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    rec = json.loads(out.read_text())
    st = rec["in_step"]["cfg2"]
    assert abs(st["ffn_fwd_att"]["mfma_util"] - 0.20) < 1e-3 and st["ffn_fwd_att"]["launches"] == 2       # <.., 2> and <.., 1>: not <.., 0>
    assert abs(st["ffn_bwd_att"]["mfma_util"] - 0.20) < 1e-3 and abs(st["edge_fwd"]["mfma_util"] - 0.25) < 1e-3
    assert abs(st["scatter_in_edge_fwd"]["hbm_frac"] - 24e6 / 30e-6 / 8e12) < 1e-3
    assert abs(rec["per_symbol"]["cfg2"]["ffn_fwd_kernel<false, 64, 0>"]["mfma_util_active"] - 0.50) < 1e-3
    from dostransformer_amd._lib import source_hash
    assert rec["source_hash"] == source_hash() and rec["git_head"] == "abc"


def test_kernel_only_tool_divides_site_work_by_traced_kernel_time(tmp_path):
    """tools/kernel_only.py: per site, algorithmic work per step / summed kernel durations per step of its symbol in the
    rocprofv3 --stats table (steps = adamw_kernel calls) - VERDICT r5's recomputation (7 kernels x 48.84 us for 6 brackets)."""
    stats = tmp_path / "ks.csv"
    stats.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                     '"(anonymous namespace)::wgrad_grouped_kernel((anonymous namespace)::WgradGroup)",1183,57780552,48842.4,23.9,1,2,3\n'
                     '"void (anonymous namespace)::ffn_bwd_kernel<false, 64, 2>(DosxFfnBwd)",676,35689981,52795.8,14.7,1,2,3\n'
                     '"void (anonymous namespace)::ffn_bwd_kernel<true, 64, 2>(DosxFfnBwd)",338,19578569,57924.7,8.1,1,2,3\n'
                     '"(anonymous namespace)::adamw_kernel(float*, float const*)",169,1537570,9098.0,0.6,1,2,3\n')
    sites = tmp_path / "sites.json"
    sites.write_text(json.dumps({"config": "phonon_h128_b64", "sites": [
        {"site": "wgrad_grouped", "kernel": "wgrad_grouped_kernel", "bound": "mfma", "work_per_launch": 2.2145e9, "launches_per_step": 7.0,
         "brackets_per_step": 6.0},
        {"site": "ffn_bwd[H128,att]", "kernel": "ffn_bwd_kernel", "bound": "mfma", "work_per_launch": 1.618e9, "launches_per_step": 6.0,
         "brackets_per_step": 6.0}]}))
    out = tmp_path / "ko.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_only.py"), str(stats), str(sites), str(out), "abc"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    rec = json.loads(out.read_text())
    w = rec["sites"]["wgrad_grouped"]
    assert rec["steps_traced"] == 169 and w["kernels_per_step"] == 7.0 and abs(w["us_per_step"] - 341.9) < 0.2
    assert abs(w["frac"] - 0.247) < 2e-3                                   # the judge's own figure for round 5
    fb = rec["sites"]["ffn_bwd[H128,att]"]
    assert abs(fb["us_per_step"] - (35689981 + 19578569) / 169 / 1e3) < 0.1 and fb["kernels_per_step"] == 6.0
