#!/usr/bin/env python3
"""Training-throughput benchmark of the DOSTransformer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* - python -m torch.distributed.run
     --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... - or bare: without WORLD_SIZE in the environment the process
     starts its N ranks itself, as fresh child processes, BEFORE it has touched the GPU, and relays rank 0's line)

Metric (BASELINE.json): crystals/s of full training steps (forward + loss + backward + AdamW) of the
Phonon-DOS model, --layers 3 --transformer 2 --hidden 128, batch 64 crystals per GPU (weak
scaling: N GPUs train on a global batch of 64*N), synthetic crystal graphs that are
pre-collated and resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (kind, layers, t_layers, hidden, per-GPU batch)
    "phonon_h128_b64": ("phonon", 3, 2, 128, 64),      # BASELINE.json configs[1] / [3] (64 per GPU)
    "phonon_h64_b8": ("phonon", 3, 1, 64, 8),          # configs[0] (the reference's CPU-runnable case)
    "edos_h256_b64": ("edos", 3, 2, 256, 64),          # configs[2]
    "edos_h256_t4_b32": ("edos", 3, 4, 256, 32),       # configs[4] per-GPU shape
    # the headline model at larger per-GPU batches (`secondary.batch_sweep`: where the step leaves the launch-latency regime)
    "phonon_h128_b128": ("phonon", 3, 2, 128, 128),
    "phonon_h128_b256": ("phonon", 3, 2, 128, 256),
    "phonon_h128_b512": ("phonon", 3, 2, 128, 512),
}
N_DISTINCT_BATCHES = 8
# epoch mode (--shuffle): granularity of the (nodes, edges) shape buckets.  Finer buckets = fewer ghost rows per step but
# more buckets to record (one eager step each, once): 64/1280 1.401 ms (4 live buckets), 32/640 1.358 (8), 16/320 1.349
# (16), 8/160 1.340 (30, at the slot cap) on the synthetic phonon set, misses inside the timed region included
SHUFFLE_BUCKET = (16, 320)
PROMOTE = float(os.environ.get("DOSX_BENCH_PROMOTE", "0.08"))   # Trainer(promote=...) of the --shuffle runs
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak


def build_model(kind, L, T, H, device):
    torch.manual_seed(0)
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        return DOSTransformer_phonon(L, T, 118, 4, H, device, 0.0)
    from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
    return DOSTransformer(L, T, 200, 41, 2, H, device, 0.0)


def make_crystals(kind, n, seed, dtype):
    from dostransformer_amd import synth
    return synth.phonon_crystals(n, seed, dtype) if kind == "phonon" else synth.edos_crystals(n, seed, dtype)


def cpu_baseline(kind, L, T, H, B, budget_s=15.0):
    """The oracle (CPU restatement of the reference math; fp64 for phonon like main_phDOS.py:15-16,
    fp32 for eDOS) timed on this box's host cores: full train steps on one batch."""
    from oracle import dos_oracle as O
    from dostransformer_amd.batch import collate
    dt = torch.float64 if kind == "phonon" else torch.float32
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        model = build_model(kind, L, T, H, "cpu")
        params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    finally:
        torch.set_default_dtype(prev)
    g = collate(make_crystals(kind, B, 1000, dt))
    state = {}
    # pick the thread count the oracle runs fastest with on this box (the reference pins 2,
    # main_phDOS.py:12; all cores of a big host oversubscribe these small ops badly)
    best_t, best, t2 = 2, None, None
    for nt in (2, 8, 16, 32):
        if nt > (os.cpu_count() or 1):
            break
        torch.set_num_threads(nt)
        O.train_step(kind, params, state, g, L, T)        # warm-up at this thread count
        t0 = time.perf_counter()
        O.train_step(kind, params, state, g, L, T)
        el = time.perf_counter() - t0
        if nt == 2:
            t2 = el                                        # the reference's own setting (torch.set_num_threads(2))
        if best is None or el < best:
            best_t, best = nt, el
    torch.set_num_threads(best_t)
    t0 = time.perf_counter()
    n = 0
    while True:
        O.train_step(kind, params, state, g, L, T)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    # the same step in fp32 on the same threads (phonon only: the reference runs it in fp64, main_phDOS.py:15-16, the
    # GPU path computes in fp32 — so the like-for-like arithmetic comparison is this figure, not the fp64 one)
    f32_value = None
    if dt == torch.float64:
        p32 = {k: (v.float() if v.is_floating_point() else v) for k, v in params.items()}
        g32 = collate(make_crystals(kind, B, 1000, torch.float32))
        s32 = {}
        O.train_step(kind, p32, s32, g32, L, T)
        t0 = time.perf_counter()
        n32 = 0
        while True:
            O.train_step(kind, p32, s32, g32, L, T)
            n32 += 1
            e32 = time.perf_counter() - t0
            if e32 > max(2.0, budget_s / 4) or n32 >= 100:
                break
        f32_value = round(B * n32 / e32, 1)
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # value: the reference's arithmetic (fp64 for phonon); value_f32: the same port in the GPU path's arithmetic - the
    # like-for-like GPU/CPU ratio is against THAT one; value_ref_threads: with the reference's own torch.set_num_threads(2)
    return {"value": round(B * n / el, 2), "unit": "crystals/s", "cores": torch.get_num_threads(), "kind": "port",
            "value_f32": f32_value, "value_ref_threads": round(B / t2, 1), "ref_threads": 2,
            "host": f"{cpu_model}, {os.cpu_count()} logical CPUs",
            "sample": f"{n} full train steps of one {B}-crystal batch, {'fp64' if dt == torch.float64 else 'fp32'}, "
                      f"{el:.1f}s, {1e3 * el / n:.1f} ms/step, fastest of 2/8/16/32 threads"}


def algorithmic_flops(kind, L, T, H, N, E, B, n_max):
    """Forward flops of one batch by SURVEY.md §8d's formula (REAL nodes / edges, no ghost padding); a training step
    is 3x this (forward + two backward products per forward product)."""
    S = 51 if kind == "phonon" else 201
    Fa, Fb = (118, 4) if kind == "phonon" else (200, 41)
    f = 2.0 * N * (Fa * H + H * H) + 2.0 * E * (Fb * H + H * H)
    if kind == "edos":
        f += 2.0 * B * (2 * H + H * H)
    f += L * (16.0 * E * H * H + 12.0 * N * H * H)
    f += (2.0 if kind == "phonon" else 4.0) * B * H * H
    f += 5.0 * T * 16 * S * B * H * H + 3.0 * T * 4 * B * S * n_max * H + 2.0 * T * 4 * B * S * S * H
    f += 9.0 * S * B * H * H + 4.0 * S * B * H
    return f


def load_traffic(config="phonon_h128_b64"):
    """HBM bytes per kernel launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py) over THIS configuration's
    replayed step: profiles/r*_pmc_traffic.json for the headline, r*_pmc_traffic_edos.json for edos_h256_b64.  The file records
    the hash of the sources it was measured on; a file measured on different sources is REFUSED (traffic stays null)."""
    import glob
    from dostransformer_amd._lib import source_hash
    suffix = {"phonon_h128_b64": "", "edos_h256_b64": "_edos"}.get(config)
    if suffix is None:
        return {}, "traffic is profiled for phonon_h128_b64 and edos_h256_b64 only"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")))
    if not files:
        return {}, f"no profiles/r*_pmc_traffic{suffix}.json"
    try:
        rec = json.load(open(files[-1]))
    except Exception as e:  # pragma: no cover
        return {}, f"{os.path.basename(files[-1])}: {e}"
    if rec.get("source_hash") != source_hash():
        return {}, f"{os.path.basename(files[-1])} was measured on other sources (hash {rec.get('source_hash')}, now {source_hash()}): refused"
    return rec.get("kernels", {}), f"{os.path.basename(files[-1])} (git {rec.get('git_head', '?')[:12]})"


def load_kernel_only(config="phonon_h128_b64"):
    """Per-site KERNEL-ONLY figures from the committed rocprofv3 --kernel-trace --stats run of the same command
    (tools/kernel_only.py -> profiles/r*_kernel_only*.json): the site's algorithmic work per step / the summed durations of its
    kernel symbol per step - no event brackets, no dispatch gaps, no waiting for CUs.  Hash-checked like `traffic`."""
    import glob
    from dostransformer_amd._lib import source_hash
    suffix = {"phonon_h128_b64": "", "edos_h256_b64": "_edos"}.get(config)
    if suffix is None:
        return {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_kernel_only{suffix}.json")))
    if not files:
        return {}
    try:
        rec = json.load(open(files[-1]))
    except Exception:  # pragma: no cover
        return {}
    if rec.get("source_hash") != source_hash():
        return {}
    return rec.get("sites", {})


def load_north_star():
    """The two counter-based figures of BASELINE.json's north_star (scatter-add HBM fraction, attention MFMA utilisation)
    from the committed rocprofv3 PMC passes (tools/pmc_north_star.py), under the same rule as `traffic`: a file measured on
    other sources is refused."""
    import glob
    from dostransformer_amd._lib import source_hash
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_north_star.json")))
    if not files:
        return {"scatter_hbm_frac": None, "attn_mfma_util": None, "in_step": None, "source": "no profiles/r*_north_star.json"}
    name = os.path.basename(files[-1])
    try:
        rec = json.load(open(files[-1]))
    except Exception as e:  # pragma: no cover
        return {"scatter_hbm_frac": None, "attn_mfma_util": None, "source": f"{name}: {e}"[:100]}
    if rec.get("source_hash") != source_hash():
        return {"scatter_hbm_frac": None, "attn_mfma_util": None, "in_step": None,
                "source": f"{name} was measured on other sources ({rec.get('source_hash')} != {source_hash()}): refused"}
    # in_step: the same two questions asked of the kernels that RUN in the replayed steps (tools/pmc_step.py over rocprofv3 --pmc
    # passes of bench.py itself): MFMA-busy of the launches that contain the attention, bytes / time of the launch that contains
    # the scatter-add.  The two objects above are microbenchmarks of stand-alone kernels (`kind`).
    in_step = None
    sfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma_step.json")))
    if sfiles:
        try:
            srec = json.load(open(sfiles[-1]))
            if srec.get("source_hash") == source_hash():
                # (the line carries one number per launch family: MFMA-busy fraction, or HBM fraction for the scatter-add rows;
                #  durations / launch counts / per-symbol tables stay in the file)
                in_step = {cfg: {k: (v.get("mfma_util") if "mfma_util" in v else v.get("hbm_frac")) for k, v in tab.items()}
                           for cfg, tab in (srec.get("in_step") or {}).items()}
                in_step["file"] = os.path.basename(sfiles[-1])
        except Exception:  # pragma: no cover
            pass
    return {"kind": "scatter_hbm_frac / attn_mfma_util: microbench of stand-alone kernels; in_step: counters over the replayed steps",
            "scatter_hbm_frac": rec.get("scatter_hbm_frac"), "attn_mfma_util": rec.get("attn_mfma_util"), "in_step": in_step,
            "source": f"{name} (rocprofv3 --pmc, git {str(rec.get('git_head', '?'))[:12]})"}


def traffic_of(kernels, key):
    """launch-weighted mean bytes per launch over the profiled symbols that contain `key`"""
    tot = n = 0
    for sym, r in kernels.items():
        if key in sym:
            tot += r["hbm_bytes_per_launch"] * r["launches"]
            n += r["launches"]
    return int(tot / n) if n else None


def rccl_choices(path, world):
    """What RCCL's tuner chose for the collectives of the run, per message size: parsed from the TUNING lines of the debug
    file rank 0 wrote (`<bytes> Bytes -> Algo <a> proto <p> ...`; the spelling differs between RCCL releases, so the
    parser keeps whatever follows `Algo` / `proto`)."""
    import re
    if world <= 1:
        return "1 rank: RCCL reduces in place, no algorithm is selected"
    if not path:
        return "not logged (NCCL_DEBUG was set by the caller, or the backend is not nccl)"
    seen = {}
    try:
        for line in open(path, errors="replace"):
            m = re.search(r"(\d+)\s+Bytes\s*->\s*Algo\s+(\S+)\s+proto\s+(\S+)(.*)", line)
            if m:
                k = int(m.group(1))
                ent = seen.setdefault(k, {"bytes": k, "algo": m.group(2), "proto": m.group(3), "n": 0,
                                          "detail": (m.group(4) or "").strip()[:48]})
                ent["n"] += 1
    except OSError as ex:
        return f"unreadable: {ex}"
    if not seen:
        return "no TUNING lines in the RCCL debug file"
    return sorted(seen.values(), key=lambda e: -e["bytes"])[:4]


LINE_BUDGET = 5600          # bytes of the ONE stdout line (the driver keeps an 8 KB tail of stdout: 8192 bytes)


def tuning_env(environ=None) -> dict:
    """Every DOSX_* variable of the environment: the tuning / path-selection switches of the package and of libdosx (read at
    import / first use).  A benchmark number is only comparable when none is set - bench.py refuses to run with any of them
    unless --allow-env is given, and the record always carries what was in effect (`env`)."""
    environ = os.environ if environ is None else environ
    return {k: environ[k] for k in sorted(environ) if k.startswith("DOSX_")}


def refuse_tuning_env(argv, environ=None) -> dict:
    """The effective DOSX_* switches; SystemExit(2) when there are any and --allow-env is not on the command line."""
    env = tuning_env(environ)
    if env and "--allow-env" not in argv:
        raise SystemExit("bench.py: refusing to run with tuning switches in the environment (" +
                         ", ".join(f"{k}={v}" for k, v in env.items()) + "): unset them, or pass --allow-env to run anyway - "
                         "the record then lists them under `env`")
    return env


def _round_sig(v, n=4):
    if isinstance(v, float):
        return float(f"{v:.{n}g}")
    return v


def compact_record(out: dict, sites: list, budget: int = LINE_BUDGET) -> dict:
    """The record that goes on the ONE stdout line: headline fields + the dominant site + the five largest sites in short
    form.  The full per-site table NEVER goes on the line (round 2 lost its record to a 38 KB line); it is written to a side
    file by the caller.  Optional fields are dropped, least important first, until the line fits ``budget``."""
    rec = dict(out)
    rec["top_sites"] = [{"site": r["site"][:48], "frac": r["frac"], "us_per_step": r.get("us_per_step"), "bound": r["bound"]}
                        for r in sites[:5]]
    for drop in (None, "top_sites", "traffic_source", "slots", "dp", "secondary"):
        if drop is not None:
            rec.pop(drop, None)
        if len(json.dumps(rec)) <= budget:
            return rec
    # last resort: shorten the free-text fields
    if "cpu_baseline" in rec:
        rec["cpu_baseline"] = dict(rec["cpu_baseline"], sample=rec["cpu_baseline"]["sample"][:120])
    rec["config"] = {"workload": rec["config"]["workload"][:160]}
    return rec


def run_workload(name, *, device, world, rank, dp, mode, shuffle, steps, warmup, bucket, pool, instrument=True):
    """Train `steps` timed steps of configuration `name` (after `warmup` untimed ones) and return the measurements of this
    rank: elapsed seconds, host enqueue seconds, per-site kernel timing of the replayed step, slot statistics."""
    import numpy as np
    import torch.distributed as td
    from dostransformer_amd import ops
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    from dostransformer_amd.dist import shard_batch
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import Trainer

    kind, L, T, H, B = CONFIGS[name]
    model = build_model(kind, L, T, H, device).to(device)
    use_graph = mode in ("graph", "replay")          # both run on ghost-padded (N,E) shape buckets
    trainer = Trainer(model, lr=1e-4, beta=1.0, dist=dp, graph=(mode == "graph"), replay=(mode == "replay"), bucket=bucket,
                      promote=PROMOTE if shuffle else 0.0)      # (fresh random batches: rare shapes borrow a live bucket)
    n_global = B * world

    real_dims = []          # (N, E, n_max) of the un-padded batches: the algorithmic-flop count uses real rows only
    if shuffle:
        # Every rank holds its own pool (data-parallel shards of a shuffled epoch are disjoint anyway); the global
        # n_max is fixed to the pool-wide maximum so that ranks need no exchange to agree on it.
        crystals = make_crystals(kind, pool, seed=12345 + rank, dtype=torch.float32)
        ds = DeviceDataset(crystals, device)
        pool_nmax = int(max(c["x"].shape[0] for c in crystals))
        rng = np.random.default_rng(777 + rank)
        order = {"perm": rng.permutation(len(ds)), "pos": 0}

        def next_indices():
            if order["pos"] + B > len(ds):            # next epoch: reshuffle
                order["perm"], order["pos"] = rng.permutation(len(ds)), 0
            idx = order["perm"][order["pos"]:order["pos"] + B]
            order["pos"] += B
            N, E = int(ds.n_nodes[idx].sum()), int(ds.n_edges[idx].sum())
            real_dims.append((N, E, pool_nmax))
            n_pad, e_pad = bucket_sizes(N, E, *bucket)
            ops.REAL_ROWS.clear()
            ops.REAL_ROWS.update({n_pad: N, e_pad: E})
            return idx

        def next_batch():                             # (instrumented eager pass only)
            return ds.collate(next_indices(), n_max=pool_nmax)

        def do_step():
            # collate straight into the bucket's static buffers (dosx_collate_padded) + replay
            return trainer.step_dataset(ds, next_indices(), n_global, n_max=pool_nmax)
    else:
        # device-resident, pre-collated shards of N_DISTINCT_BATCHES global batches (global n_max per batch);
        # in graph / replay mode each is padded (exactly: ghost nodes/edges) to its (N, E) shape bucket
        batches = []
        for k in range(N_DISTINCT_BATCHES):
            crystals = make_crystals(kind, B * world, seed=k, dtype=torch.float32)
            g = shard_batch(crystals, world, rank)
            real_dims.append((g.meta.num_nodes, g.meta.num_edges, g.meta.n_max))
            if use_graph:
                g = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, *bucket))
            batches.append(g.to(device))
        it = {"i": 0}

        def next_batch():
            k = it["i"] % len(batches)
            g = batches[k]
            it["i"] += 1
            ops.REAL_ROWS.clear()                     # (work records are evaluated while a bucket is RECORDED)
            if use_graph:
                ops.REAL_ROWS.update({g.meta.num_nodes: real_dims[k][0], g.meta.num_edges: real_dims[k][1]})
            return g

        def do_step():
            return trainer.step(next_batch(), n_global)

    def sync():
        torch.cuda.synchronize()
        if dp is not None:
            td.barrier()

    # prepare (NOT warm-up, reported separately as `prepare_steps`): every fixed bucket is RECORDED once (an eager step that
    # builds the bucket's launch list) and REPLAYED once - the first replay of a recorded program still pays one-time costs
    # (20 timed steps: 1.316 ms after 8 such steps, 1.292 after 16).  Then exactly `warmup` untimed steps, as asked.
    n_prep = 2 * N_DISTINCT_BATCHES if (use_graph and not shuffle) else 0
    for i in range(n_prep):
        do_step()
    sync()
    for i in range(warmup):
        do_step()
    sync()
    # eager mode: per-kernel HIP-event timing over the timed region itself (same stream as the launches);
    # graph / replay mode: the timed region re-issues recorded launches (no per-kernel events possible), so the
    # kernel timing comes from an instrumented pass of the same steps right after it.
    ops.KERNEL_TIMER.reset(enabled=not use_graph)
    hits0, miss0, prom0 = trainer.slot_hits, trainer.slot_misses, trainer.slot_promoted
    if shuffle:
        real_dims.clear()
    # `check` (what the timed steps did, read AFTER the timed region): the parameters as they stand now ...
    params_before = trainer._fp.flat.clone() if trainer._fp is not None else None
    torch.cuda.synchronize()
    loss_first = loss = None
    t0 = time.perf_counter()
    for i in range(steps):
        loss = do_step()
        if i == 0:
            loss_first = loss.detach().clone()       # (the loss lives in the bucket's static buffer: one 4-byte device copy)
    host = time.perf_counter() - t0                  # the host is done enqueueing here; the GPU may still be running
    torch.cuda.synchronize()
    if dp is not None:
        td.barrier()
    elapsed = time.perf_counter() - t0
    # ... the device loss of the first and the last timed step, every parameter finite, the parameters moved
    fp_now = trainer._fp.flat
    check = {"loss_first": _round_sig(float(loss_first), 6), "loss_last": _round_sig(float(loss), 6),
             "finite": bool(torch.isfinite(fp_now).all()) and bool(torch.isfinite(loss_first)) and bool(torch.isfinite(loss)),
             "params_changed_frac": None if params_before is None else
             round(float((fp_now != params_before).float().mean()), 4),
             "replay_eq_eager": None}
    del params_before
    ops.KERNEL_TIMER.enabled = False
    hits, misses, promoted = trainer.slot_hits - hits0, trainer.slot_misses - miss0, trainer.slot_promoted - prom0
    n_slots = len(trainer._slots)
    # host time to ENQUEUE one step with an empty queue in front of it (inside the timed loop the host also blocks on the
    # HIP queue's back pressure whenever it runs ahead of the GPU, so `host` above is an upper bound, not the enqueue cost)
    enq = []
    for i in range(9):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        do_step()
        enq.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    enq.sort()
    host_enqueue = enq[len(enq) // 2]
    if shuffle:
        del real_dims[-9:]
    # ... and ONE step on the same batch from the same state issued both ways - the replayed launch list (three streams) and the
    # eager, single-stream marshalling of every call from Python - must leave the same parameters (state restored afterwards)
    if mode == "replay" and not shuffle and dp is None:
        fp = trainer._fp
        saved = (fp.flat.clone(), trainer._m.clone(), trainer._v.clone(), trainer.step_count)

        def restore():
            fp.flat.copy_(saved[0]); trainer._m.copy_(saved[1]); trainer._v.copy_(saved[2])
            trainer.step_count = saved[3]
        ops.REAL_ROWS.clear()
        g0 = batches[0]
        loss_r = trainer.step(g0, n_global).detach().clone()
        p_r = fp.flat.clone()
        restore()
        trainer.replay = False
        try:
            loss_e = trainer.step(g0, n_global).detach().clone()
        finally:
            trainer.replay = True
        p_e = fp.flat.clone()
        restore()
        torch.cuda.synchronize()
        check["replay_eq_eager"] = bool(torch.equal(p_r, p_e)) and bool(torch.equal(loss_r, loss_e))
        check["replay_eager_max_abs_diff"] = _round_sig(float((p_r - p_e).abs().max()), 3)
        del saved, p_r, p_e
    dims = list(real_dims) if shuffle else [real_dims[i % len(real_dims)] for i in range(steps)]
    n_inst = 0
    if not instrument:
        pass
    elif mode == "replay":
        # per-launch durations INSIDE the replayed three-stream step: the recorded programs are re-issued through
        # dosx_replay_timed (a HIP event pair around every entry, on the stream it launches on); the few launches outside
        # the recordings (slot copy, AdamW) are bracketed by ops._call
        trainer.kernel_timer = ops.KERNEL_TIMER
        ops.KERNEL_TIMER.reset(enabled=True)
        n_inst = min(steps, 24)
        for i in range(n_inst):
            do_step()
            torch.cuda.synchronize()
        ops.KERNEL_TIMER.enabled = False
        trainer.kernel_timer = None
    elif use_graph:
        trainer.graph = trainer.replay = False
        ops.KERNEL_TIMER.reset(enabled=True)
        n_inst = min(steps, 24)
        for i in range(n_inst):
            # An eager step is host-bound (~17 us of Python per launch): without a head start the GPU
            # would idle between a site's start event and its kernel and the bracket would time the host.
            # A device-side spin lets the host enqueue the whole step first, so every bracket times
            # back-to-back GPU execution (a bracket then adds ~1 us to a kernel: tools/event_overhead.py).
            g = next_batch()
            torch.cuda._sleep(int(1.5e7))
            trainer.step(g, n_global)
            torch.cuda.synchronize()
        ops.KERNEL_TIMER.enabled = False
    else:
        n_inst = steps
    ops.REAL_ROWS.clear()
    roof = ops.KERNEL_TIMER.roofline(HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS) if instrument else {"dominant": None, "all": []}
    ops.KERNEL_TIMER.reset(enabled=False)
    flops_step = 3.0 * sum(algorithmic_flops(kind, L, T, H, n, e, B, nm) for n, e, nm in dims) / max(len(dims), 1)
    dp_info = None
    if dp is not None and trainer._fp is not None:
        fp = trainer._fp              # the two gradient buckets of the data-parallel step and the replay plan
        plan = next((s.plan for s in trainer._slots.values() if s.plan), [])
        has_mid = any(k == "mid" for k, _ in plan)
        # early: transformers / heads / embeddings (all-reduced under the GNN backward); mid: GN_decoder + layers L-1 .. 1 (under
        # layer 0's backward); late: encoders + layer 0 - the only bytes reduced BEHIND the backward pass (`exposed_bytes`)
        dp_info = {"grad_bucket_bytes": {"early": 4 * int(fp.total - fp.n_late), "mid": 4 * int(fp.n_late - fp.n_last) if has_mid else 0,
                                         "late": 4 * int(fp.n_last if has_mid else fp.n_late)},
                   "exposed_bytes": 4 * int(fp.n_last if has_mid else fp.n_late),
                   "collectives_per_step": sum(1 for k, _ in plan if k != "prog") + 1,
                   "plan": [k for k, _ in plan] + ["late", "adamw"], "backend": td.get_backend(),
                   "staged_through_host": bool(dp.staged)}
    res = {"kind": kind, "L": L, "T": T, "H": H, "B": B, "n_global": n_global, "elapsed": elapsed, "host": host, "host_enqueue": host_enqueue,
           "roof": roof, "n_inst": n_inst, "flops_step": flops_step, "prepare_steps": n_prep, "dp_info": dp_info, "check": check,
           "slots": {"hits": hits, "misses": misses, "hit_rate": round(hits / max(hits + misses, 1), 4),
                     "promoted": promoted, "live": n_slots, "max": trainer.max_slots} if use_graph else None}
    del trainer, model
    torch.cuda.empty_cache()
    return res


def eval_throughput(device, B=64, n_batches=4, iters=100):
    """crystals/s of replayed inference on B-crystal batches with per-crystal key counts (= B batch-1 forwards of the reference's
    evaluation loop), and of the batch-1 loop itself for comparison."""
    from dostransformer_amd.batch import collate
    from dostransformer_amd.predict import Predictor
    kind, L, T, H, _ = CONFIGS["phonon_h128_b64"]
    model = build_model(kind, L, T, H, device).to(device).eval()
    pred = Predictor(model, per_crystal_keys=True).eval()
    cs = [make_crystals(kind, B, seed=900 + k, dtype=torch.float32) for k in range(n_batches)]
    batches = [collate(c).to(device) for c in cs]
    ones = [collate([c]).to(device) for c in cs[0][:16]]
    for g in batches + ones:
        pred(g); pred(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        pred(batches[i % n_batches])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(iters):
        pred(ones[i % len(ones)])
    torch.cuda.synchronize()
    el1 = time.perf_counter() - t0
    return {"value": round(B * iters / el, 1), "unit": "crystals/s", "ms_per_batch": round(1e3 * el / iters, 4),
            "batch1_loop": round(iters / el1, 1)}


def split_bf16_gemms(device, iters=30):
    """VERDICT r5 item 9 (secondary only; the headline and every program stay exact fp32): the split-bf16 GEMM (csrc/gemm_bf16x3.hip -
    three bf16 terms per fp32 operand element, six products on the bf16 matrix pipe, fp32 accumulation) next to dosx_gemm on the two
    largest GEMM shapes of the Electron-DOS step (fc1 forward, fc2 input gradient: M = 201 * 128 rows, N = 1024, K = 256).  Errors
    are max |C - C64| / max |C64| against float64 on the first 4096 rows, for both kernels on the same operands."""
    from dostransformer_amd import ops
    out = {"dtype": "fp32 in/out; bf16 x 3 split (6 products, fp32 accumulate) vs fp32 MFMA"}
    g = torch.Generator(device="cpu").manual_seed(5)
    for name, M, N, K, wl in (("fc1_fwd", 25728, 1024, 256, 0), ("fc2_dgrad", 25728, 1024, 256, 1)):
        a = torch.randn(M, K, generator=g).to(device)
        w = (torch.randn(N, K, generator=g) if wl == 0 else torch.randn(K, N, generator=g)).to(device)
        c3, c32 = torch.empty(M, N, device=device), torch.empty(M, N, device=device)

        def timed(fn):
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) * 1e3 / iters
        us3 = timed(lambda: ops.gemm_bf16x3(a, w, c3, w_layout=wl))
        us32 = timed(lambda: ops.gemm(M, N, [ops.seg(a)], w, c32, w_layout=wl))
        ref = a[:4096].double() @ (w.double().T if wl == 0 else w.double())
        sc = float(ref.abs().max())
        out[name] = {"M": M, "N": N, "K": K, "us": round(us3, 1), "us_fp32": round(us32, 1), "speedup": round(us32 / us3, 3),
                     "tflops_fp32_equiv": round(2.0 * M * N * K / us3 / 1e6, 1),
                     "err": float(f"{float((c3[:4096].double() - ref).abs().max()) / sc:.3g}"),
                     "err_fp32": float(f"{float((c32[:4096].double() - ref).abs().max()) / sc:.3g}")}
    return out


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    """Environment of rank `rank` of a one-node job of `world` processes (what torch.distributed.run would set)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's cross-process buffers need it on this stack
    return env


def self_launch(world: int, argv, *, grace_s: float = 30.0, script=None) -> int:
    """`python bench.py --gpus N` with no launcher: start the N ranks as FRESH child processes of this same script (one per
    GPU, rank r -> cuda:r), relay rank 0's single JSON line to stdout and return the worst exit code.  The calling process
    has not initialised the GPU and never does (no HIP call, no exec of a process that used the GPU); a rank that dies takes
    the others down after `grace_s` seconds (they would wait in a collective for ever) - by their exact PIDs."""
    import signal
    import subprocess
    import threading
    port = _free_port()
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(cmd, env=rank_env(r, world, port), cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def forward(signum, _frame):          # the driver's timeout / ^C reaches the ranks too
        for p in procs:
            if p.poll() is None:
                p.send_signal(signum)
    old = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT)}
    ended = set()                     # ranks this launcher ended itself: their codes say nothing
    try:
        failed_at = None
        while any(p.poll() is None for p in procs):
            if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
                failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > grace_s:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                        ended.add(p.pid)
            time.sleep(0.1)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    reader.join(timeout=10)
    rcs = [p.returncode for p in procs]
    text = (out0[0] if out0 else b"").decode(errors="replace").strip()
    if text:
        sys.stdout.write(text.splitlines()[-1] + "\n")
        sys.stdout.flush()
    own = [p.returncode for p in procs if p.pid not in ended] or rcs
    worst = max(own, key=lambda c: (c != 0, abs(c)))
    if worst != 0:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
    return worst if worst >= 0 else 128 - worst


class _SkipDp1(Exception):
    pass


def main():
    env_switches = refuse_tuning_env(sys.argv[1:])        # before anything can touch the GPU or read a switch
    if "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`, N > 1: become the launcher - decided on the command line alone, before anything
        # below (or any import side effect) can touch the GPU
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        n = pre.parse_known_args()[0].gpus
        if n > 1:
            raise SystemExit(self_launch(n, sys.argv[1:]))
    # Exactly ONE line may reach stdout (the JSON record): libraries print there too (RCCL emits a
    # version banner on fd 1), so fd 1 is pointed at stderr for the whole run and the record is
    # written to the saved descriptor at the end.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="phonon_h128_b64", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements of the default run (eDOS H256 line, --shuffle line)")
    ap.add_argument("--no-dp1", action="store_true",
                    help="skip the `dp1_nccl` secondary (the headline workload through the data-parallel step on a 1-rank RCCL "
                         "group: it initialises a process group inside this process)")
    ap.add_argument("--kernels-out", default=os.path.join(ROOT, "bench_kernels_last.json"),
                    help="side file for the full per-site kernel table (never on the stdout line)")
    ap.add_argument("--launch", choices=["replay", "eager", "graph"], default="replay",
                    help="replay: re-issue a recorded launch list on static buffers (3 HIP streams); eager: marshal "
                         "every launch from Python; graph: torch/HIP graph replay")
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--shuffle", action="store_true",
                    help="steady-state epoch mode: every step collates a FRESH random batch from a device-resident pool "
                         "(loader.DeviceDataset), pads it to its shape bucket and trains on it; reports the slot hit rate")
    ap.add_argument("--pool", type=int, default=1536, help="--shuffle: crystals in the device-resident pool per GPU")
    ap.add_argument("--bucket", type=int, nargs=2, default=None, metavar=("NODES", "EDGES"),
                    help="shape-bucket granularity (default 8 128; --shuffle: 16 320)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (production); gloo = host-staged sums, only for running the N > 1 code path "
                         "on a box with fewer GPUs than ranks (together with --share-gpu; no scaling meaning)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (test harness; see --dist-backend)")
    ap.add_argument("--allow-env", action="store_true",
                    help="run although DOSX_* tuning switches are set in the environment (A/B experiments); they are listed in the "
                         "record's `env` field.  Without this flag bench.py aborts when any is set")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed/RCCL even for one rank (exercises the data-parallel code path on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda:0" if args.share_gpu else f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    import torch.distributed as td
    dp = None
    rccl_log = None
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.dist_backend == "nccl" and world > 1 and rank == 0 and "NCCL_DEBUG" not in os.environ:
            # SURVEY.md §5 "verify which algorithm RCCL selects": the tuner's decision per collective size goes to a side
            # file on rank 0 (one short line per collective; read back after the timed region, see rccl_choices)
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), f"dosx_rccl_tuning_{os.getpid()}.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="TUNING", NCCL_DEBUG_FILE=rccl_log)
        # (no device_id=: the eager communicator it creates costs every later kernel launch of this process
        #  ~3 us on this stack: 2.42 vs 1.92 ms/step measured with tools/dist_overhead.py; the device is
        #  already selected with torch.cuda.set_device above)
        td.init_process_group(args.dist_backend, rank=rank, world_size=world)
        from dostransformer_amd.dist import DataParallel
        dp = DataParallel()

    mode = "graph" if args.graph else args.launch
    bucket = tuple(args.bucket) if args.bucket else (SHUFFLE_BUCKET if args.shuffle else (8, 128))
    common = dict(device=device, world=world, rank=rank, dp=dp, mode=mode, pool=args.pool)
    r = run_workload(args.config, shuffle=args.shuffle, steps=args.steps, warmup=args.warmup, bucket=bucket, **common)
    kind, L, T, H, B, n_global = r["kind"], r["L"], r["T"], r["H"], r["B"], r["n_global"]

    t = torch.tensor([r["elapsed"], r["host"]], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
    if dp is not None:
        td.all_reduce(t, op=td.ReduceOp.MAX)
    elapsed, host = float(t[0]), float(t[1])

    # secondary measurements of the DEFAULT run (one GPU, default configuration): the other single-GPU BASELINE
    # configuration and the steady-state (fresh batch every step) figure, so that the driver's record holds them too
    secondary = {}
    if world == 1 and dp is None and not args.no_secondary and args.config == "phonon_h128_b64" and not args.shuffle \
            and mode == "replay":
        def brief(x, steps):
            ms = 1e3 * x["elapsed"] / steps
            return {"value": round(x["n_global"] * steps / x["elapsed"], 1), "unit": "crystals/s", "ms_per_step": round(ms, 4),
                    "steps": steps, "step_frac": round(x["flops_step"] / (ms * 1e-3) / (MFMA_F32_PEAK_TFLOPS * 1e12), 4),
                    "host_ms_per_step": round(1e3 * x["host_enqueue"], 4)}
        try:
            e = run_workload("edos_h256_b64", shuffle=False, steps=40, warmup=8, bucket=(8, 128), instrument=False, **common)
            secondary["edos_h256_b64"] = brief(e, 40)
            # BASELINE.json configs[4]'s per-GPU shard (Electron-DOS H256 T4, 32 crystals per GPU of the 8 x 32 job)
            e4 = run_workload("edos_h256_t4_b32", shuffle=False, steps=30, warmup=6, bucket=(8, 128), instrument=False, **common)
            secondary["edos_h256_t4_b32"] = brief(e4, 30)
            sh = run_workload("phonon_h128_b64", shuffle=True, steps=args.steps, warmup=max(args.warmup, 60),
                              bucket=SHUFFLE_BUCKET, instrument=False, **common)
            secondary["shuffle"] = dict(brief(sh, args.steps), hit_rate=sh["slots"]["hit_rate"], promoted=sh["slots"]["promoted"], live_buckets=sh["slots"]["live"])
            # the headline model at 128 / 256 / 512 crystals on this ONE GPU: where the step leaves the launch-latency regime
            # (what per-GPU batch a multi-GPU run should use); the headline itself stays on 64
            sweep = {"64": {"value": round(n_global * args.steps / elapsed, 1), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                            "step_frac": round(r["flops_step"] / (elapsed / args.steps) / (MFMA_F32_PEAK_TFLOPS * 1e12), 4)}}
            for bsz in (128, 256, 512):
                w = run_workload(f"phonon_h128_b{bsz}", shuffle=False, steps=30, warmup=6, bucket=(8, 128), instrument=False, **common)
                b_ = brief(w, 30)
                sweep[str(bsz)] = {k: b_[k] for k in ("value", "ms_per_step", "step_frac")}
            secondary["batch_sweep"] = sweep
        except Exception as ex:  # a secondary line must never take the headline down with it
            secondary["error"] = f"{type(ex).__name__}: {ex}"[:200]
        # batch-size-1-equivalent EVALUATION in one pass: Predictor(per_crystal_keys=True) on 64-crystal
        # batches of the headline model - the reference evaluates crystal by crystal (batch_size = 1, utils.py:61-143)
        try:
            secondary["eval_per_crystal_b64"] = eval_throughput(device)
        except Exception as ex:
            secondary["eval_per_crystal_b64"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        try:
            secondary["split_bf16"] = split_bf16_gemms(device)
            # ... and the Electron-DOS step with its plain feed-forward GEMMs on that kernel (opt-in functional._FFN_BF16X3; the
            # default programs - everything else on this line - are exact fp32)
            from dostransformer_amd import functional as _Fn
            _Fn._FFN_BF16X3 = True
            try:
                e3 = run_workload("edos_h256_b64", shuffle=False, steps=40, warmup=8, bucket=(8, 128), instrument=False, **common)
            finally:
                _Fn._FFN_BF16X3 = False
            secondary["split_bf16"]["edos_h256_b64_step_ms"] = round(1e3 * e3["elapsed"] / 40, 4)
        except Exception as ex:
            secondary["split_bf16"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        # the same headline workload through the DATA-PARALLEL step on a 1-rank RCCL group: the replay plan split around
        # the collectives (SSE pair, early gradient bucket, late bucket), i.e. what data parallelism costs a rank before
        # any byte crosses xGMI - the figure a 1-GPU box can give about the N > 1 path
        try:
            if args.no_dp1:
                raise _SkipDp1()
            from dostransformer_amd.dist import DataParallel
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ["MASTER_PORT"] = str(_free_port())
            td.init_process_group("nccl", rank=0, world_size=1)
            try:
                d1 = run_workload("phonon_h128_b64", shuffle=False, steps=args.steps, warmup=args.warmup, bucket=(8, 128),
                                  instrument=False, **dict(common, dp=DataParallel()))
                info = d1["dp_info"] or {}
                secondary["dp1_nccl"] = dict(brief(d1, args.steps), **{k: info.get(k) for k in
                                                                       ("grad_bucket_bytes", "exposed_bytes", "collectives_per_step", "backend")})
            finally:
                td.destroy_process_group()
        except _SkipDp1:
            pass
        except Exception as ex:
            secondary["dp1_nccl"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}

    if rank == 0:
        roof = r["roof"]
        n_inst = r["n_inst"]
        kernels, traffic_src = load_traffic(args.config) if (world == 1 and not args.shuffle) \
            else ({}, "traffic is profiled for the single-GPU fixed-batch runs only")
        kernel_only = load_kernel_only(args.config) if (world == 1 and not args.shuffle) else {}
        for rec in roof["all"] + ([roof["dominant"]] if roof["dominant"] else []):
            rec["traffic"] = traffic_of(kernels, rec["kernel"])
            rec["us_per_step"] = round(1e3 * rec["total_ms"] / max(n_inst, 1), 2)
            # launches_per_step: device KERNELS per step (rocprofv3's call count); brackets_per_step: the timed event pairs - a
            # dosx_grad_flush bracket holds one kernel per table of 8 jobs (6 brackets = 7 kernels)
            rec["launches_per_step"] = round(rec.get("kernel_launches", rec["launches"]) / max(n_inst, 1), 2)
            rec["brackets_per_step"] = round(rec["launches"] / max(n_inst, 1), 2)
            ko = kernel_only.get(rec["site"])
            rec["kernel_only_frac"] = ko["frac"] if ko else None
            rec["kernel_only_us_per_step"] = ko["us_per_step"] if ko else None
        ms_step = 1e3 * elapsed / args.steps
        flops_step = r["flops_step"]
        dom = roof["dominant"]
        if dom is not None:          # the roofline object of the contract: dominant site only, compact
            dom = {k: _round_sig(dom[k], 5) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "site", "kernel",
                                                       "avg_us", "launches_per_step", "brackets_per_step", "us_per_step", "work_per_launch",
                                                       "kernel_only_frac", "kernel_only_us_per_step")}
            dom["timing"] = "HIP event pair per launch inside the replayed step, on the launching stream" if mode == "replay" \
                else "HIP events per launch"
        out = {
            "metric": "crystals/sec training throughput (Phonon DOS, hidden=128)" if kind == "phonon" else
                      "crystals/sec training throughput (Electron DOS)",
            "value": round(n_global * args.steps / elapsed, 2),
            "unit": "crystals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prepare_steps": r["prepare_steps"],
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {kind} DOSTransformer layers={L} transformer={T} hidden={H}, "
                                   f"{B} crystals/GPU, full train step (fwd+loss+bwd+AdamW), " +
                                   (f"fresh random batch every step from a device-resident pool of {args.pool}, collated on the "
                                    f"GPU inside the timed region" if args.shuffle else
                                    f"{N_DISTINCT_BATCHES} distinct pre-collated HBM-resident batches"),
                       "global_batch": n_global, "parallelism": f"dp{world}", "launch": mode, "bucket": list(bucket)},
            # whole-step figure: algorithmic flops of a train step (SURVEY.md §8d formula on the real, un-padded rows,
            # x3 for fwd+bwd) / measured step time / fp32 MFMA dense peak
            "step_frac": round(flops_step / (ms_step * 1e-3) / (MFMA_F32_PEAK_TFLOPS * 1e12), 4),
            "step_gflop": round(flops_step / 1e9, 2),
            # host time to enqueue one replayed step into an EMPTY queue (median of 9): the margin to the GPU step; and the
            # per-step time of the timed loop's host side (includes blocking on the queue's back pressure)
            "host_ms_per_step": round(1e3 * r["host_enqueue"], 4),
            "host_loop_ms_per_step": round(1e3 * host / args.steps, 4),
            "roofline": dom,
            "traffic_source": traffic_src[:100],
            # the DOSX_* switches in effect ({} = the shipped defaults) and the sanity checks of the timed run (run_workload)
            "env": env_switches,
            "check": r["check"],
        }
        if world == 1 and args.config == "phonon_h128_b64":
            out["north_star"] = load_north_star()
        if r["slots"] is not None:
            out["slots"] = r["slots"]
        if r["dp_info"] is not None:
            out["dp"] = dict(r["dp_info"], rccl=rccl_choices(rccl_log, world))
            if rccl_log is not None:          # (the tuner's side file has been read: do not leave it in the temp directory)
                try:
                    os.unlink(rccl_log)
                except OSError:
                    pass
        if secondary:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, L, T, H, B, args.cpu_budget)
        try:
            with open(args.kernels_out, "w") as f:
                json.dump({"config": args.config, "ms_per_step": ms_step, "instrumented_steps": n_inst, "sites": roof["all"]}, f, indent=1)
            out["kernels_file"] = os.path.relpath(args.kernels_out, ROOT)
        except OSError as ex:
            print(f"bench.py: could not write {args.kernels_out}: {ex}", file=sys.stderr)
        line = json.dumps(compact_record(out, roof["all"]))
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())
    if dp is not None:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
