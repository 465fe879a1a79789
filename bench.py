#!/usr/bin/env python
"""Benchmark of the Point-GNN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic KITTI-shaped frame
(~20k points): device-side graph construction (keypoints + two radius graphs)
+ PointSetPooling + T GraphNetAutoCenter iterations + prediction heads for
`car_auto_T3` inference -- what run.py does per frame between "fetch input"
and "decode box" (run.py:219-263).  Inputs (xyz, intensity) are resident in
HBM before the timed region.  Frames are independent: rank r of N processes
takes frames r, r+N, ... (weak scaling, no collective on the data path).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: `roofline` (standalone scatter-max kernel, HBM-bound, timed with
events on the launch stream) and `cpu_baseline` (the oracle = CPU port of the
reference path, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: FP32 matrix peak
BF16_MFMA_PEAK_TF = 2500.0   # ... dense bf16 matrix peak (no sparsity)


def algorithmic_flops_per_frame(cfg, n_k, n_e0, n_e1):
    """2*in*out per row per FC layer, un-factorised."""
    lcs = cfg['model_kwargs']['layer_configs']
    total = 0
    dim = 1
    for lc in lcs[:-1]:
        kw = lc['kwargs']
        if lc['type'] == 'scatter_max_point_set_pooling':
            d = dim + 3
            for w in kw['point_MLP_depth_list']:
                total += 2 * d * w * n_e0
                d = w
            for w in kw['output_MLP_depth_list']:
                total += 2 * d * w * n_k
                d = w
            dim = d
        else:
            if kw['auto_offset']:
                d = dim
                for w in kw['auto_offset_MLP_depth_list']:
                    total += 2 * d * w * n_k
                    d = w
            d = dim + 3
            for w in kw['edge_MLP_depth_list']:
                total += 2 * d * w * n_e1
                d = w
            for w in kw['update_MLP_depth_list']:
                total += 2 * d * w * n_k
                d = w
    nc = cfg['num_classes']
    total += n_k * (2 * dim * 64 + 2 * 64 * nc)
    total += n_k * nc * (2 * dim * 64 + 2 * 64 * 64 + 2 * 64 * 7)
    return total


def time_kernel(fn, reps, torch):
    """Average duration (s) of fn() over `reps` back-to-back launches between
    ONE pair of HIP events on the current (launch) stream.  Kernels queued back
    to back start where the previous one ends (rocprofv3 traces show 0.0 us
    gaps), so this is the kernel's own duration as long as fn() launches that
    kernel alone -- roofline_edge_kernel does (aggregation buffer pre-filled,
    no fill launch).  (An event pair round EVERY launch was tried in round 4
    and reads ~55 us MORE per launch than the trace for the 1 ms edge kernel:
    the two event records are commands of their own.)"""
    for _ in range(3):
        fn()
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) * 1e-3 / reps


def live_pmc_scatter(preset, timeout_s=150):
    """HBM bytes of the standalone scatter-max measured in THIS run: two
    child processes under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE
    and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) each
    replay tools/kernel_bench.py scatter on seed 0 of `preset` -- the frame
    the roofline is timed on.  Returns the dict tools/pmc_scatter_json.py
    builds, or None (rocprofv3 missing / failed / timed out)."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_scatter_json as pj
    finally:
        sys.path.pop(0)
    tmp = tempfile.mkdtemp(prefix="pgnn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals, kernel, wl = {}, None, None
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, c)
            p = subprocess.run(
                [rocprof, "--kernel-trace", "--pmc", c, "-d", out_dir, "-o",
                 "pmc", "--", sys.executable,
                 os.path.join(ROOT, "tools", "kernel_bench.py"), "scatter",
                 "--reps", "3", "--preset", preset],
                cwd="/tmp", env=env, capture_output=True, text=True,
                timeout=timeout_s)
            for line in p.stdout.splitlines():
                if line.startswith("{"):
                    wl = json.loads(line).get("workload")
            db = None
            for root, _, files in os.walk(out_dir):
                for f in files:
                    if f.endswith(".db"):
                        db = os.path.join(root, f)
            if p.returncode != 0 or db is None:
                return None
            avg, n, kernel = pj.avg_counter(db, c)
            if avg is None:
                return None
            vals[c], vals[c + "_n"] = avg, n
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {
        "kernel": pj.short_kernel_name(kernel), "workload": wl,
        "FETCH_SIZE_KiB": vals["FETCH_SIZE"],
        "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
        "launches_averaged": vals["FETCH_SIZE_n"],
        "hbm_bytes_per_launch":
            (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
        "write_note": pj.WRITE_NOTE,
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in "
                  "separate child runs of tools/kernel_bench.py scatter taken "
                  "by this bench.py invocation; bytes = (2*FETCH_SIZE + "
                  "WRITE_SIZE)*1024: KiB units, and FETCH_SIZE counts half of "
                  "a wide coalesced read on gfx950 (MI355X_MICROARCH.md, HBM)",
    }


def live_kernel_trace(config, preset, edge_arith="f32", steps=8, warmup=2,
                      timeout_s=240):
    """Average kernel durations of ONE frame alone, measured in THIS run by a
    child `rocprofv3 --kernel-trace` of `bench.py --frames 1 --no-pipeline`
    (seed 0 of `preset`, frames strictly one after the other: the command
    behind profiles/*_infer_seed0_kernel_stats).  Returns {short kernel name:
    {"avg_us", "calls", "total_us"}} plus "_command", or None (rocprofv3
    missing / failed / timed out)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_scatter_json as pj
    finally:
        sys.path.pop(0)
    tmp = tempfile.mkdtemp(prefix="pgnn_trace_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config,
           "--preset", preset, "--edge-arith", edge_arith, "--steps",
           str(steps), "--warmup", str(warmup), "--frames", "1",
           "--frames-per-step", "1", "--repeats", "1", "--no-pipeline",
           "--no-cpu-baseline", "--no-secondary", "--no-live-pmc",
           "--no-roofline", "--no-capture"]
    try:
        p = subprocess.run([rocprof, "--kernel-trace", "-d", tmp, "-o", "run",
                            "--"] + cmd, cwd="/tmp", env=env,
                           capture_output=True, text=True, timeout=timeout_s)
        db = None
        for root, _, files in os.walk(tmp):
            for f in files:
                if f.endswith(".db"):
                    db = os.path.join(root, f)
        if p.returncode != 0 or db is None:
            return None
        con = sqlite3.connect(db)
        cols = [c[1] for c in con.execute("pragma table_info(kernels)")]
        c_name = "name" if "name" in cols else "kernel_name"
        c_s = "start" if "start" in cols else "start_time"
        c_e = "end" if "end" in cols else "end_time"
        out = {}
        for name, avg, n, tot in con.execute(
                "select %s, avg(%s - %s), count(*), sum(%s - %s) from kernels "
                "group by %s" % (c_name, c_e, c_s, c_e, c_s, c_name)):
            out[pj.short_kernel_name(name)] = {
                "avg_us": avg / 1e3, "calls": int(n), "total_us": tot / 1e3}
        out["_command"] = "rocprofv3 --kernel-trace -- python bench.py " + \
            " ".join(cmd[2:])
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def apply_kernel_trace(line, trace, prefix):
    """`line` (a roofline dict timed by HIP events) re-priced with the child
    pass's average duration of the kernel whose short name starts with
    `prefix`: avg_launch_us / achieved / frac follow rocprofv3 (what profiles/
    holds), the event figures stay beside them."""
    if line is None or not trace:
        return line
    if isinstance(prefix, (tuple, list)):
        # a stage of several launches: the sum of its kernels' averages
        parts = []
        for pre in prefix:
            hits = [(k, v) for k, v in trace.items()
                    if k.startswith(pre) and isinstance(v, dict)]
            if not hits:
                return line
            parts.append(max(hits, key=lambda kv: kv[1]["total_us"]))
        name = " + ".join(k for k, _ in parts)
        v = {"avg_us": sum(p["avg_us"] for _, p in parts),
             "calls": min(p["calls"] for _, p in parts)}
    else:
        hits = [(k, v) for k, v in trace.items()
                if k.startswith(prefix) and isinstance(v, dict)]
        if not hits:
            return line
        name, v = max(hits, key=lambda kv: kv[1]["total_us"])
    ev_us = line["avg_launch_us"]
    work = line["achieved"] * ev_us          # TFLOP/s x us
    line["avg_launch_us_events"] = ev_us
    line["frac_events"] = line["frac"]
    line["avg_launch_us"] = v["avg_us"]
    line["achieved"] = work / v["avg_us"]
    line["frac"] = line["achieved"] / line["peak"]
    line["duration_source"] = {
        "kernel": name, "launches_averaged": v["calls"],
        "method": "average kernel duration in a rocprofv3 --kernel-trace "
                  "child pass of this bench.py invocation (" +
                  trace.get("_command", "") + "); avg_launch_us_events / "
                  "frac_events: one HIP-event pair round 10 back-to-back "
                  "launches on the launch stream"}
    return line


def live_pmc_train_mfma_ops(config, preset, fpg, frames=8, steps=6, warmup=2,
                            timeout_s=240):
    """fp32-MFMA work of ONE training step measured in this run: a child
    `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32` replays
    `bench.py --train` (prebuilt batches: the graph build has no MFMA) for
    steps + warmup steps; FLOPs = sum of the counter over every dispatch x 512
    / steps run (one count = 512 FLOP: 2 * 304^2 * E of the edge kernel comes
    out exactly, profiles/r04_pmc_sq_car_600k.txt).  Returns a dict or None
    (rocprofv3 missing / failed / timed out)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    tmp = tempfile.mkdtemp(prefix="pgnn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    counter = "SQ_INSTS_VALU_MFMA_MOPS_F32"
    try:
        p = subprocess.run(
            [rocprof, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o",
             "pmc", "--", sys.executable, os.path.abspath(__file__), "--train",
             "--config", config, "--preset", preset, "--frames-per-gpu",
             str(fpg), "--frames", str(frames), "--steps", str(steps),
             "--warmup", str(warmup), "--train-loader", "prebuilt",
             "--no-live-pmc", "--no-bind"],
            cwd="/tmp", env=env, capture_output=True, text=True,
            timeout=timeout_s)
        db = None
        for root, _, files in os.walk(tmp):
            for f in files:
                if f.endswith(".db"):
                    db = os.path.join(root, f)
        if p.returncode != 0 or db is None:
            return None
        con = sqlite3.connect(db)
        total, n_disp = con.execute(
            "select sum(value), count(distinct dispatch_id) from "
            "counters_collection where counter_name = ?", (counter,)).fetchone()
        top = con.execute(
            "select kernel_name, sum(value) from counters_collection where "
            "counter_name = ? group by kernel_name order by 2 desc limit 6",
            (counter,)).fetchall()
        con.close()
        if not total:
            return None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    n = steps + warmup
    short = lambda k: k.replace("(anonymous namespace)::", "").replace(  # noqa: E731
        "pgnn::", "").split("(")[0][:60]
    return {
        "counter": counter, "steps_profiled": n, "dispatches": int(n_disp),
        "mfma_flops_per_step": float(total) * 512.0 / n,
        "by_kernel_gflop_per_step": {short(k): float(v) * 512.0 / n / 1e9
                                     for k, v in top},
        "method": "child run of `bench.py --train --train-loader prebuilt` "
                  "under rocprofv3 --kernel-trace --pmc %s taken by this "
                  "bench.py invocation; FLOPs = sum over all dispatches x 512 "
                  "/ %d steps" % (counter, n),
    }


def roofline_scatter_max(torch, edges1, n_k, width, reps=30, live_pmc=None):
    """Standalone scatter-max on an [E1, C] fp32 matrix resident in HBM, dst
    ids of the real level-1 graph (sorted).  Algorithmic bytes per launch =
    E*C*4 + E*4 + K*C*4."""
    from pointgnn_amd import gnn
    n_e = int(edges1.shape[0])
    # several distinct buffers so that the 256 MiB Infinity Cache cannot serve
    # the reads of the next launch
    bufs = [torch.randn((n_e, width), device=edges1.device) for _ in range(3)]
    dst = edges1[:, 1].contiguous()
    state = {"i": 0}

    def run():
        gnn.graph_scatter_max_fn(bufs[state["i"] % 3], dst, n_k,
                                 ids_sorted=True)
        state["i"] += 1
    dur = time_kernel(run, reps, torch)
    alg = n_e * width * 4 + n_e * 4 + n_k * width * 4
    # HBM traffic cannot be read from inside this process: `live_pmc` (two
    # rocprofv3 --pmc child runs of this same kernel on this same workload,
    # taken by the caller) supplies it; failing that, the committed PMC pass
    # (tools/pmc_scatter.sh) when it was taken on exactly this workload; else
    # it stays null.
    traffic, traffic_src = None, None
    if live_pmc is not None and live_pmc.get("workload") == {
            "E": n_e, "C": width, "K": n_k}:
        traffic = live_pmc["hbm_bytes_per_launch"]
        traffic_src = "live: " + live_pmc["method"]
    import glob
    for side in ([] if traffic is not None else sorted(glob.glob(os.path.join(
            ROOT, "profiles", "r*pmc_scatter_max*.json")), reverse=True)):
        with open(side) as fh:
            pm = json.load(fh)
        wl = pm.get("workload", {})
        if (wl.get("E"), wl.get("C"), wl.get("K")) == (n_e, width, n_k):
            traffic = pm["hbm_bytes_per_launch"]
            traffic_src = "profiles/%s: %s" % (os.path.basename(side),
                                               pm["method"])
            break
    return {
        "traffic_source": traffic_src,
        "kernel": "scatter_max_kernel (standalone, [E1,C] fp32, sorted dst)",
        "bound": "hbm", "achieved": alg / dur / 1e9, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": alg / dur / 1e9 / HBM_PEAK_GBS,
        "traffic": traffic, "algorithmic_bytes": alg,
        "traffic_detail": None if live_pmc is None else {
            k: live_pmc[k] for k in ("kernel", "FETCH_SIZE_KiB",
                                     "WRITE_SIZE_KiB", "launches_averaged",
                                     "write_note")},
        "avg_launch_us": dur * 1e6,
        "note": "duration includes the 4*K*C-byte lowest() fill memset",
        "scope": "the kernel BASELINE.json's metric and SURVEY 8(d) name "
                 "(standalone scatter-max, HBM-bound); the kernel with the "
                 "largest share of GPU time in the frame is the fused edge "
                 "kernel, whose MFMA roofline is in `roofline_mfma`",
    }


def roofline_edge_kernel(torch, engine, edges1, n_k, reps=10, frame=None):
    """Fused gather + edge-MLP layer 2 + scatter-max kernel (MFMA-bound)."""
    from pointgnn_amd import _lib, gnn
    lib = _lib.load()
    store = engine.model._store
    lc = [l for l in engine.config['model_kwargs']['layer_configs']
          if l['type'] == 'scatter_max_graph_auto_center_net']
    if not lc:
        return None
    key = [k for k in store._cache if k[0] == 'edge']
    if not key:
        return None
    c, p_chain, wx_dev, rest = store._cache[key[0]]
    wq = int(wx_dev.shape[1])
    dev = edges1.device
    # the first GNN iteration's real per-vertex inputs of this frame (MFMA
    # power, hence the clock, depends on the operand values: dense random
    # operands run ~7 % slower than a frame's own activations)
    if frame is not None:
        gnn.EDGE_INPUT_TAP = []
        engine.run_frame(*frame)
        p, q = gnn.EDGE_INPUT_TAP[0]
        gnn.EDGE_INPUT_TAP = None
        p, q = p.clone(), q.clone()
    else:
        p = torch.randn((n_k, wq), device=dev)
        q = torch.randn((n_k, wq), device=dev) * 0.1
        p[:, c:] = 0
        q[:, c:] = 0
    # the aggregation buffer holds lowest() before the first launch, as
    # vertex_pre_edge leaves it in a frame (flag bit 1: no fill launch inside
    # the call); later launches max the same values into it again
    agg = torch.full((n_k, gnn.padded_width(rest.n_out)),
                     float(np.finfo(np.float32).min), device=dev)
    n_e = int(edges1.shape[0])

    def run():
        _lib.check(lib.pgnn_edge_mlp_scatter_max_fwd(
            _lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(edges1),
            n_e, n_k, rest.array, rest.n, 1 | 2, _lib.ptr(agg), agg.stride(0),
            _lib.ptr(_lib.sched_ws()), _lib.stream_ptr()), "edge kernel")
    dur = time_kernel(run, reps, torch)
    widths = lc[0]['kwargs']['edge_MLP_depth_list']
    executed = sum(2 * a * b for a, b in zip(widths[:-1], widths[1:])) * n_e
    return {
        "kernel": "edge_ws_kernel (weights-stationary fused gather + edge "
                  "FC2 + scatter-max; csrc/edge_ws.h)",
        "bound": "mfma", "achieved": executed / dur / 1e12,
        "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
        "frac": executed / dur / 1e12 / FP32_MFMA_PEAK_TF,
        "executed_flops": executed, "avg_launch_us": dur * 1e6,
        "note": "fp32 MFMA (16x16x4); FLOPs = 2*E*sum(in*out) of the layers "
                "this kernel executes (first edge layer is factored per "
                "vertex); duration = one HIP-event pair round 10 "
                "back-to-back launches / 10, the call launches this kernel "
                "alone (aggregation buffer pre-filled, as in a frame): "
                "compare rocprofv3's average for edge_ws_kernel in profiles/",
    }


def roofline_edge_kernel_16bit(torch, engine, edges1, n_k, arith, reps=10,
                               frame=None):
    """The 16-bit matrix-pipe forms of the fused edge kernel
    (csrc/edge_ws_bf16.h: 'bf16x3', csrc/edge_ws_f16.h: 'f16x2') against THEIR
    bound: the bf16 / fp16 matrix pipe at six / three MFMAs per product."""
    from pointgnn_amd import _lib, gnn
    lib = _lib.load()
    store = engine.model._store
    key = [k for k in store._cache if k[0] == 'edge']
    img = [k for k in store._cache if k[0] == 'edge_' + arith]
    if not key or not img or frame is None:
        return None
    c, p_chain, wx_dev, rest = store._cache[key[0]]
    image = store._cache[img[0]]
    wq = int(wx_dev.shape[1])
    dev = edges1.device
    gnn.EDGE_INPUT_TAP = []
    engine.run_frame(*frame)
    p, q = [t.clone() for t in gnn.EDGE_INPUT_TAP[0]]
    gnn.EDGE_INPUT_TAP = None
    agg = torch.full((n_k, gnn.padded_width(rest.n_out)),
                     float(np.finfo(np.float32).min), device=dev)
    n_e = int(edges1.shape[0])
    head = (_lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(edges1),
            n_e, n_k, _lib.ptr(image), int(rest.n_out),
            int(rest.array[0].relu_from), 1 | 2, _lib.ptr(agg), agg.stride(0))
    status = torch.zeros(1, dtype=torch.int32, device=dev)

    def run():
        if arith == "bf16x3":
            rc = lib.pgnn_edge_mlp_scatter_max_bf16x3_fwd(
                *head, None, None, _lib.stream_ptr())
        else:
            rc = lib.pgnn_edge_mlp_scatter_max_f16x2_fwd(
                *head, _lib.ptr(status), None, None, _lib.stream_ptr())
        _lib.check(rc, arith + " edge kernel")
    dur = time_kernel(run, reps, torch)
    flops = 2.0 * int(rest.k_in) * int(rest.n_out) * n_e
    terms = 6 if arith == "bf16x3" else 3
    peak = BF16_MFMA_PEAK_TF / terms
    return {
        "kernel": "edge_ws_%s_kernel (weights-stationary fused gather + %s "
                  "+ edge FC2 as %d 16-bit MFMA products + scatter-max; "
                  "csrc/edge_ws_%s.h)" % (
                      arith, "3-way exact bf16 split" if terms == 6 else
                      "two-part fp16 representation", terms,
                      "bf16" if terms == 6 else "f16"),
        "bound": "mfma", "achieved": flops / dur / 1e12, "peak": peak,
        "unit": "TFLOP/s (fp32-equivalent)", "frac": flops / dur / 1e12 / peak,
        "executed_flops": flops, "avg_launch_us": dur * 1e6,
        "note": "peak = dense 16-bit MFMA peak (%.0f TFLOP/s) / %d products "
                "per fp32-equivalent product; FLOPs = 2*E*k_in*n_out of the "
                "layer (the K padding 300 -> 320 and the 16 zero columns are "
                "not counted); duration as for roofline_mfma: compare "
                "rocprofv3's average for edge_ws_%s_kernel in profiles/"
                % (BF16_MFMA_PEAK_TF, terms, arith),
    }


def roofline_pool_kernel(torch, engine, reps=10, frame=None):
    """Fused PointSetPooling kernel: per-edge gather [f, dxyz] -> point MLP ->
    segmented max (MFMA-bound; the kernel furthest below its roofline)."""
    from pointgnn_amd import _lib, gnn
    lib = _lib.load()
    store = engine.model._store
    lc = engine.config['model_kwargs']['layer_configs'][0]
    if lc['type'] != 'scatter_max_point_set_pooling' or frame is None:
        return None
    widths = list(lc['kwargs']['point_MLP_depth_list'])
    key = ('mlp', lc['scope'] + '/extract_vertex_features', tuple(widths),
           False)
    if key not in store._cache:
        return None
    chain = store._cache[key]
    x, f = frame
    engine.run_frame(x, f)
    coords, kps, edges = engine.last_graph
    e0 = edges[0]
    kp = kps[0].reshape(-1).to(torch.int32).contiguous()
    n_feat = int(f.shape[1])
    k = int(kp.shape[0])
    agg = torch.empty((k, gnn.padded_width(chain.n_out)), device=x.device)
    n_e = int(e0.shape[0])

    import ctypes
    ws_bytes = ctypes.c_size_t(0)
    _lib.check(lib.pgnn_point_set_pooling_workspace_bytes(
        chain.array, chain.n, n_feat, n_e, 0, ctypes.byref(ws_bytes)),
        "pooling workspace query")
    work = torch.empty(max(ws_bytes.value // 4, 1), device=x.device)

    def run():
        # (with a workspace of 0 bytes: exactly pgnn_point_set_pooling_fwd)
        _lib.check(lib.pgnn_point_set_pooling_fwd_ws(
            _lib.ptr(f), n_feat, _lib.ptr(x), _lib.ptr(kp), _lib.ptr(e0), n_e,
            k, chain.array, chain.n, 1, _lib.ptr(agg), agg.stride(0),
            _lib.ptr(_lib.sched_ws()), None, None,
            _lib.ptr(work) if ws_bytes.value else None, ws_bytes.value,
            _lib.stream_ptr()), "pooling kernel")
    dur = time_kernel(run, reps, torch)
    dims = [n_feat + 3] + widths
    flops = sum(2 * a * b for a, b in zip(dims[:-1], dims[1:])) * n_e
    return {
        "kernel": ("pool_hidden_kernel + edge_ws_kernel<16, 8, false, true> "
                   "(two launches, hidden rows [E0, 256] through a %.2f GB "
                   "workspace: gather + point MLP %s + scatter-max)"
                   % (ws_bytes.value / 1e9, "->".join(map(str, dims))))
        if ws_bytes.value else
                  "pool_ws_kernel / fused_mlp_kernel<POOL> (gather + point "
                  "MLP %s + scatter-max)" % "->".join(map(str, dims)),
        "launches": 2 if ws_bytes.value else 1,
        "bound": "mfma", "achieved": flops / dur / 1e12,
        "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
        "frac": flops / dur / 1e12 / FP32_MFMA_PEAK_TF,
        "algorithmic_flops": flops, "avg_launch_us": dur * 1e6,
        "workload": {"E0": n_e, "K": k},
    }


def roofline_pool_kernel_f16x2(torch, engine, reps=10, frame=None):
    """The pooling stage of an 'f16x2' model (csrc/pool_ws_f16.h: the 64->128
    and 128->300 layers as three fp16 products per block, the two narrowest
    layers fp32 MFMA), priced in fp32-equivalent FLOPs against the 16-bit MFMA
    peak / 3 like the edge kernel."""
    from pointgnn_amd import _lib, gnn
    lib = _lib.load()
    store = engine.model._store
    lc = engine.config['model_kwargs']['layer_configs'][0]
    if lc['type'] != 'scatter_max_point_set_pooling' or frame is None:
        return None
    widths = list(lc['kwargs']['point_MLP_depth_list'])
    scope = lc['scope'] + '/extract_vertex_features'
    chain = store._cache.get(('mlp', scope, tuple(widths), False))
    image = store._cache.get(('pool_f16x2', scope, tuple(widths)))
    hidden = store._cache.get(('pool_f16x2_h', scope, tuple(widths)))
    if chain is None or image is None:
        return None
    x, f = frame
    engine.run_frame(x, f)
    coords, kps, edges = engine.last_graph
    e0 = edges[0]
    kp = kps[0].reshape(-1).to(torch.int32).contiguous()
    n_feat = int(f.shape[1])
    k = int(kp.shape[0])
    agg = torch.empty((k, gnn.padded_width(chain.n_out)), device=x.device)
    n_e = int(e0.shape[0])
    status = store.range_status()

    def run():
        _lib.check(lib.pgnn_point_set_pooling_f16x2_fwd(
            _lib.ptr(f), n_feat, _lib.ptr(x), _lib.ptr(kp), _lib.ptr(e0), n_e,
            k, chain.array, chain.n, _lib.ptr(image), _lib.ptr(hidden), 1,
            _lib.ptr(agg), agg.stride(0), _lib.ptr(_lib.sched_ws()),
            _lib.ptr(status), None, None, _lib.stream_ptr()),
            "f16x2 pooling kernel")
    dur = time_kernel(run, reps, torch)
    dims = [n_feat + 3] + widths
    flops = sum(2 * a * b for a, b in zip(dims[:-1], dims[1:])) * n_e
    peak = BF16_MFMA_PEAK_TF / 3
    return {
        "kernel": "pool_ws_f16x2_kernel (gather + point MLP %s + scatter-max; "
                  "64->128 and 128->300 as 3 fp16 products, the rest fp32 "
                  "MFMA; csrc/pool_ws_f16.h)" % "->".join(map(str, dims)),
        "bound": "mfma", "achieved": flops / dur / 1e12, "peak": peak,
        "unit": "TFLOP/s (fp32-equivalent)",
        "frac": flops / dur / 1e12 / peak,
        "algorithmic_flops": flops, "avg_launch_us": dur * 1e6,
        "note": "peak = dense 16-bit MFMA peak (%d TFLOP/s) / 3 products per "
                "fp32-equivalent product; 2 %% of the FLOPs (the 4->32->64 "
                "layers) run as fp32 MFMA" % BF16_MFMA_PEAK_TF,
        "workload": {"E0": n_e, "K": k},
    }


def _host_description():
    """CPU model, core count and BLAS backend of this host (SURVEY.md 8d asks
    for them next to the CPU baseline)."""
    model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas, threads = None, os.cpu_count() or 1
    try:
        import threadpoolctl
        infos = threadpoolctl.threadpool_info()
        for p in infos:
            if p.get("user_api") == "blas":
                blas = "%s %s (%s)" % (p.get("internal_api"), p.get("version"),
                                       p.get("threading_layer", "?"))
        threads = max([p.get('num_threads', 1) for p in infos] + [1])
    except Exception:
        pass
    import platform
    return {"cpu_model": model or platform.processor() or platform.machine(),
            "host_cpu_count": os.cpu_count(), "blas": blas,
            "blas_threads": int(threads)}


def cpu_baseline(cfg, params, xyz, inten, budget_s=20.0):
    """The oracle (CPU port of the reference path: scikit-learn ball tree as
    the reference calls it + the torch-CPU GNN of oracle/gnn_oracle_torch.py:
    gather -> GEMM chain -> segment max in cache-sized row chunks, all cores)
    timed on this host with SURVEY
    8d's protocol: third-party imports before the clock, one warm-up, median
    of 5 for the graph build (single-threaded as the reference ships it,
    graph_gen.py:85,208); the GNN part (BLAS on all cores) is bounded: it runs
    on the sub-graph of the first K' keypoints, K' chosen from a short GEMM
    probe so that the whole leg stays near budget_s, one warm-up + median of
    3, scaled by the FLOP fraction."""
    from oracle import graph_oracle as go
    from oracle import gnn_oracle_torch as gn   # the GEMM-bound host port
    import sklearn.neighbors  # noqa: F401  (imported before the clock starts)
    import torch as _torch
    host = _host_description()
    kw = cfg['runtime_graph_gen_kwargs']
    coords, kps, edges = go.multi_level_graph(xyz, **kw)      # warm-up
    t_graphs = []
    for _ in range(5):
        t0 = time.perf_counter()
        go.multi_level_graph(xyz, **kw)
        t_graphs.append(time.perf_counter() - t0)
    t_graph = float(np.median(t_graphs))
    n_k = coords[1].shape[0]
    e0, e1 = np.asarray(edges[0]), np.asarray(edges[1])
    # GEMM probe -> sustained GFLOP/s of this host's BLAS
    a = np.random.default_rng(0).standard_normal((4096, 304)).astype(np.float32)
    b = np.random.default_rng(1).standard_normal((304, 304)).astype(np.float32)
    a @ b
    t = time.perf_counter()
    for _ in range(5):
        a @ b
    gflops = 5 * 2 * 4096 * 304 * 304 / (time.perf_counter() - t) / 1e9
    total = algorithmic_flops_per_frame(cfg, n_k, len(e0), len(e1))
    est = total / (gflops * 1e9) * 1.5
    # 1 warm-up + 3 timed runs of the GNN share the remaining budget
    frac = min(1.0, max(0.02, (budget_s - 6 * t_graph) / 4.0 / max(est, 1e-9)))
    k_sub = max(16, int(n_k * frac))
    m0 = e0[:, 1] < k_sub
    # level-1 sub-graph induced by the first k_sub keypoints
    m1 = (e1[:, 1] < k_sub) & (e1[:, 0] < k_sub)
    sub_coords = [coords[0], coords[1][:k_sub], coords[2][:k_sub]]
    sub_kps = [kps[0][:k_sub], kps[1][:k_sub]]
    sub_edges = [e0[m0], e1[m1]]
    # torch's intra-op threading stops scaling long before this host's core
    # count (car_600k frame, 256 hardware threads: 178 GFLOP/s at 32 intra-op
    # threads, 147 at 64 = 5-6 % of the sgemm probe): the port runs
    # chunk-parallel instead -- one worker thread per CPU the rank's affinity
    # mask allows, single-threaded ops, 2048-row chunks -- which reaches
    # 1.1-1.3 TFLOP/s = 0.39-0.48 of the probe, flat from 16 to 256 workers
    # (the ceiling: profiles/r06_cpu_baseline_sweep.txt)
    n_thr = _torch.get_num_threads()
    allowed = len(os.sched_getaffinity(0))
    _torch.set_num_threads(1)
    gn.CHUNK_ROWS, gn.WORKERS = 2048, min(256, allowed)
    try:
        gn.predict(params, cfg, inten, sub_coords, sub_kps, sub_edges)  # warm-up
        t_gnns = []
        for _ in range(3):
            t = time.perf_counter()
            gn.predict(params, cfg, inten, sub_coords, sub_kps, sub_edges)
            t_gnns.append(time.perf_counter() - t)
        threads = int(gn.WORKERS)
    finally:
        _torch.set_num_threads(n_thr)
        gn.WORKERS = 0
    t_gnn_sub = float(np.median(t_gnns))
    sub_flops = algorithmic_flops_per_frame(cfg, k_sub, int(m0.sum()),
                                            int(m1.sum()))
    t_gnn_full = t_gnn_sub * total / max(sub_flops, 1)
    out = {
        "value": 1.0 / (t_graph + t_gnn_full), "unit": "frames/s",
        "cores": int(threads), "kind": "port",
        "sample": "1 frame (%s seed of the headline pool): graph build with "
                  "the reference's sklearn calls, single-threaded as shipped, "
                  "1 warm-up + median of 5 (%.2f s; min %.2f, max %.2f); GNN "
                  "oracle (torch-CPU fp32, 2048-row chunks on %d worker "
                  "threads = every CPU of the rank's affinity mask; NumPy "
                  "sgemm probe %.0f GFLOP/s) on "
                  "the sub-graph of the first %d of %d keypoints (%.1f%% of "
                  "the frame's FLOPs), 1 warm-up + median of 3 (%.2f s), "
                  "scaled to the full frame (%.1f s)"
                  % ("first", t_graph, min(t_graphs), max(t_graphs), threads,
                     gflops, k_sub, n_k, 100.0 * sub_flops / total, t_gnn_sub,
                     t_gnn_full),
        "gen_graph_s": t_graph, "gnn_inference_s_scaled": t_gnn_full,
        # the GNN port against this host's own sgemm rate ("BLAS
        # threads = all host cores"); the thread-count x chunk sweep behind
        # the choice is profiles/r06_cpu_baseline_sweep.txt
        "achieved_gflops": sub_flops / t_gnn_sub / 1e9,
        "probe_gflops": gflops,
        "frac_of_probe": sub_flops / t_gnn_sub / 1e9 / gflops,
        "allowed_cpus": allowed,
    }
    out.update(host)
    return out


def roofline_graph(torch, cfg, coords):
    """Radius-graph kernels alone (count pass + scan + fill pass per level, no
    host read in between: the capacity is known from the frame's own graph):
    edges/s and algorithmic bytes 12(P+Q) + 8E over the launch time.  These
    kernels are latency-bound (~7 MB per frame), so the fraction of
    the HBM peak is small by construction; reported as the spec asks."""
    from pointgnn_amd import _lib
    lib = _lib.load()
    out = []
    for lc in cfg['runtime_graph_gen_kwargs']['level_configs']:
        lvl = lc['graph_level']
        r = float(lc['graph_gen_kwargs']['radius'])
        pts, ctr = coords[lvl].contiguous(), coords[lvl + 1].contiguous()
        n_p, n_c = int(pts.shape[0]), int(ctr.shape[0])
        ws_bytes = lib.pgnn_radius_graph_workspace_bytes(n_p, n_c)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pts.device)
        off = torch.empty(n_c + 1, dtype=torch.int32, device=pts.device)

        def count():
            _lib.check(lib.pgnn_radius_graph_count(
                _lib.ptr(pts), n_p, _lib.ptr(ctr), n_c, r, None, _lib.ptr(ws),
                ws_bytes, _lib.ptr(off), _lib.stream_ptr()), "count")
        count()
        n_e = int(off[-1].item())
        edges = torch.empty((n_e, 2), dtype=torch.int32, device=pts.device)

        def both():
            count()
            _lib.check(lib.pgnn_radius_graph_fill(
                _lib.ptr(pts), n_p, _lib.ptr(ctr), n_c, r, None, _lib.ptr(ws),
                ws_bytes, _lib.ptr(off), _lib.ptr(edges), n_e,
                _lib.stream_ptr()), "fill")
        dur = time_kernel(both, 20, torch)
        alg = 12 * (n_p + n_c) + 8 * n_e
        out.append({"level": lvl, "radius": r, "P": n_p, "Q": n_c, "E": n_e,
                    "us": dur * 1e6, "edges_per_s": n_e / dur,
                    "algorithmic_bytes": alg, "GB_per_s": alg / dur / 1e9,
                    "frac_of_hbm_peak": alg / dur / 1e9 / HBM_PEAK_GBS})
    tot_e = sum(o["E"] for o in out)
    tot_t = sum(o["us"] for o in out) * 1e-6
    return {"kernel": "radius graph: cell keys + radix sort + bounds + "
                      "radius_query<count> + scan + radius_query<fill> "
                      "(csrc/graph.hip, csrc/sort.hip), keypoints excluded",
            "bound": "latency (HBM-side bytes are ~MB)", "levels": out,
            "edges_per_s": tot_e / tot_t, "us": tot_t * 1e6,
            "GB_per_s": sum(o["algorithmic_bytes"] for o in out) / tot_t / 1e9}


def make_communicator(torch, dist):
    """The training step's communicator: RCCL behind the C ABI
    (pgnn_comm_* / pgnn_allreduce_step), its 128 id bytes handed out through
    the process group the launcher already formed.  None for one rank, and in
    the one-GPU test mode (two ranks on cuda:0: RCCL refuses two ranks per
    device; the step's collectives then ride on the gloo group)."""
    if dist is None or STUB or ONE_GPU:
        return None
    from pointgnn_amd.comm import Communicator
    return Communicator.from_torch(dist.group.WORLD)


def collective_fixed_cost(torch, dev, n_params, reps=30):
    """What the step's two collectives cost on THIS stack when nobody has to be
    waited for: a world-1 RCCL communicator (a valid one: unique id,
    ncclCommInitRank, the reduction kernels) reducing the flat gradient + the
    four loss sums as one group (pgnn_allreduce_step) and the two endpoint
    counts (pgnn_allreduce_sum_f64), timed by events round `reps` calls on the
    current stream.  At N ranks the call additionally moves 2(N-1)/N x the
    buffer over xGMI and waits for the slowest rank."""
    from pointgnn_amd.comm import Communicator
    comm = Communicator.single()
    grad = torch.randn(n_params, dtype=torch.float32, device=dev)
    sums = torch.ones(4, dtype=torch.float64, device=dev)
    counts = torch.ones(2, dtype=torch.float64, device=dev)

    def timed(fn):
        for _ in range(5):
            fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        comm.allreduce_step(grad, sums)
    host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    out = {
        "what": "world-1 RCCL communicator behind the C ABI: the call's fixed "
                "cost on this stack (no peer to wait for, nothing crosses "
                "xGMI); HIP events round %d back-to-back calls" % reps,
        "allreduce_ms": timed(lambda: comm.allreduce_step(grad, sums)),
        "counts_allreduce_ms": timed(lambda: comm.allreduce_sum(counts)),
        "host_enqueue_ms": host,
        "allreduce_bytes": int(n_params) * 4 + 32,
        "rccl_version": Communicator.rccl_version(),
        "rccl_library": Communicator.library(),
    }
    comm.destroy()
    return out


def train_measure(torch, dev, rank, world, dist, config_name, preset, steps,
                  warmup, frames, fpg=2, pipeline=True, deferred=True,
                  force_collective=False):
    """BASELINE config 4: `steps` timed training steps of `config_name` --
    per rank and step: training-mode graph build (voxel 0.8 m, random
    keypoints + origin jitter, level-1 fan-in capped at 256) for `fpg` frames,
    frame merge, forward, loss, backward, ONE all-reduce of the flat gradient,
    SGD.  `deferred`: each step's losses are read after the NEXT step is
    queued (Trainer.train_step(deferred=True)); all of them inside the timed
    region.  Returns (elapsed max over ranks, all-reduce ms, trainer, per-step
    shapes [(K, E0, E1)], last loss dict)."""
    from pointgnn_amd import configs, graph_gen, train
    from pointgnn_amd.synthetic import synthetic_cloud
    cfg = configs.get_config(config_name)
    # the gradient all-reduce: RCCL over xGMI through the C ABI's communicator
    # (pgnn_trainer_backward_sync enqueues it behind the last gradient kernel);
    # force_collective: a world-1 communicator, every collective issued
    pg = dist.group.WORLD if dist is not None else None
    comm = make_communicator(torch, dist)
    if comm is None and force_collective:
        from pointgnn_amd.comm import Communicator
        comm = Communicator.single()
    tr = train.Trainer(cfg, seed=0, device=dev, process_group=pg, comm=comm,
                       force_collective=force_collective)
    n_steps = steps + warmup
    pool = {}
    for s in range(frames):
        xyz, inten = synthetic_cloud(seed=s, preset=preset)
        pool[s] = (torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev))
    gen = torch.Generator(device='cpu').manual_seed(1234 + rank)
    np.random.seed(99 + rank)
    hints = graph_gen.CountHints()

    def make_frame(i):
        x, f = pool[i % frames]
        # the training graph in capacity form, ONE size read per frame (what
        # train.fetch_data does with its graph_hints)
        coords, kps, edges = graph_gen.gen_multi_level_local_graph_v3_one_read(
            x, hints, **cfg['graph_gen_kwargs'])
        k = int(coords[1].shape[0])
        lab = (torch.rand(k, generator=gen) < 0.2).to(torch.int32) * \
            torch.randint(1, 3, (k,), generator=gen, dtype=torch.int32)
        lab = lab.reshape(k, 1).to(dev)
        boxes = torch.randn((k, 1, 7), generator=gen).to(dev)
        valid = (lab > 0).to(torch.float32).reshape(k, 1, 1)
        return (f, coords, kps, edges, lab, boxes, valid)

    shapes = []
    # The data side (graph build + frame merge of step i+1) runs on its own
    # stream while step i's forward/backward occupy the compute stream -- the
    # reference hides it behind 16 loader processes (train.py:430-440).
    # (a stream on a hardware queue of its own: see engine.concurrent_streams)
    from pointgnn_amd.engine import concurrent_streams
    sg = concurrent_streams(1, dev)[0]
    cur = torch.cuda.current_stream()

    def make_batch(i):
        with torch.cuda.stream(sg):
            # every rank walks the whole frame pool (offset by its rank): the
            # same mix of graph sizes per GPU at every N
            fr = [make_frame((rank + i) * fpg + j) for j in range(fpg)]
            batch = train.batch_data(fr)
            nv = float(sum(float(x[6].sum().item()) for x in fr))
        return batch, nv

    def use(batch):
        cur.wait_stream(sg)
        for t in [batch[0]] + list(batch[1]) + list(batch[2]) + \
                list(batch[3]) + list(batch[4:]):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        shapes.append((int(batch[1][1].shape[0]), int(batch[3][0].shape[0]),
                       int(batch[3][1].shape[0])))

    state = {}
    mode = pipeline if isinstance(pipeline, str) else \
        ("thread" if pipeline else "off")
    prebuilt = []
    loader = None
    if mode == "thread":
        # the data side on a loader thread with the stream `sg` (the
        # reference: a 16-process loader pool, train.py:430-440): the stepping
        # thread only enqueues steps and reads their losses
        def make_in_thread(i):
            fr = [make_frame((rank + i) * fpg + j) for j in range(fpg)]
            batch = train.batch_data(fr)
            nv = float(sum(float(x[6].sum().item()) for x in fr))
            return batch, nv
        loader = train.BatchPrefetcher(make_in_thread, range(n_steps + 6),
                                       depth=2, device=dev, stream=sg)

    def step(i):
        """Queue step i; returns the loss dict of step i-1 when the losses are
        read one step late (`deferred`), of step i otherwise."""
        if mode == "prebuilt":   # diagnostic: the step without its data side
            if len(prebuilt) < max(1, frames // fpg):
                prebuilt.append(make_batch(i))
            batch, nv = prebuilt[i % len(prebuilt)]
            after = None
        elif mode == "thread":
            batch, nv = next(loader)
            after = None
            shapes.append((int(batch[1][1].shape[0]), int(batch[3][0].shape[0]),
                           int(batch[3][1].shape[0])))
            res = tr.train_step(batch, num_valid=nv, deferred=deferred)
            if not deferred:
                return res
            prev = state.get('result')
            state['result'] = res
            return prev.get() if prev is not None else None
        else:
            if 'next' not in state:
                state['next'] = make_batch(i)
            batch, nv = state.pop('next')
            after = None if mode == "off" else (
                lambda: state.__setitem__('next', make_batch(i + 1)))
        use(batch)
        res = tr.train_step(batch, num_valid=nv, after_enqueue=after,
                            deferred=deferred)
        if not deferred:
            return res
        prev = state.get('result')
        state['result'] = res
        return prev.get() if prev is not None else None

    def drain():
        prev = state.pop('result', None)
        return prev.get() if prev is not None else None

    for i in range(warmup):
        step(i)
    drain()
    _sync(torch, dist)
    del shapes[:]
    t0 = time.perf_counter()
    for i in range(warmup, n_steps):
        out = step(i)
    out = drain() or out     # every step's losses are read inside the region
    _sync(torch, dist)
    elapsed = time.perf_counter() - t0
    # the collective alone, AFTER the timed region (bracketing it with events
    # keeps it out of pgnn_trainer_backward_sync): a few more steps
    ar_ms = 0.0
    if tr._multi():
        tr.allreduce_events = []
        for i in range(n_steps, n_steps + 6):
            step(i)
        drain()
        _sync(torch, dist)
        ar_ms = (sum(a.elapsed_time(b) for a, b in tr.allreduce_events) /
                 max(1, len(tr.allreduce_events)))
        tr.allreduce_events = None
    elapsed, ar_ms = _max_over_ranks(torch, dist, dev, [elapsed, ar_ms])
    if loader is not None:
        loader.close()
    tr.collective = "none" if not tr._multi() else (
        "pgnn_allreduce_step (RCCL behind the C ABI, world %d)" % comm.world
        if comm is not None else "torch.distributed all_reduce (%s)"
        % dist.get_backend())
    if comm is not None:
        # (explicitly, while the runtime is alive; the trainer keeps no other
        # use for it after the measurement)
        torch.cuda.synchronize()
        tr.comm = None
        if tr.force_collective:
            tr.force_collective = False
        comm.destroy()
    return elapsed, ar_ms, tr, cfg, list(shapes), out


def train_roofline(cfg, shapes, fpg, elapsed, steps, live=None):
    """MFMA roofline of the training step, three ways (peak 157.3 TFLOP/s):
      frac_equivalent_dense  3 x the forward's ALGORITHMIC FLOPs (SURVEY 8d's
                             per-row figures; backward = dX and dW GEMMs of
                             every layer, each the forward's size) / step time:
                             what a dense, unfactorised implementation would
                             have to sustain -- not a utilisation;
      frac_dense_executed    3 x the forward's EXECUTED FLOPs (first edge layer
                             per vertex) / step time: still dense
                             (the sparse scatter-max adjoint executes a
                             fraction of the backward's edge GEMMs);
      frac_mfma_ops          the fp32-MFMA operations the step's kernels
                             actually issue (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512
                             summed over a profiled child run, `live`) / step
                             time: the utilisation of the matrix pipe.
    `frac` is the last one when it was measured (else null: the first two are
    not utilisations)."""
    a = np.asarray(shapes, dtype=np.float64)
    fwd = float(np.mean([algorithmic_flops_per_frame(cfg, *map(int, r))
                         for r in a]))
    exe = float(np.mean([executed_flops_per_frame(cfg, *map(int, r))
                         for r in a]))
    per_step = elapsed / steps
    out = {
        "kernel": "whole training step (forward + loss + backward + SGD)",
        "bound": "mfma", "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
        "achieved": None, "frac": None,
        "frac_equivalent_dense": 3 * fwd / per_step / 1e12 / FP32_MFMA_PEAK_TF,
        "frac_dense_executed": 3 * exe / per_step / 1e12 / FP32_MFMA_PEAK_TF,
        "frac_mfma_ops": None,
        "formula": "frac = frac_mfma_ops = sum(SQ_INSTS_VALU_MFMA_MOPS_F32) x "
                   "512 per step / step time / 157.3 TFLOP/s; "
                   "frac_equivalent_dense = 3 x algorithmic_forward_flops(K, "
                   "E0, E1 of the merged %d-frame batch) / step time / peak; "
                   "frac_dense_executed likewise on the executed forward" % fpg,
        "algorithmic_forward_gflop": fwd / 1e9,
        "dense_executed_forward_gflop": exe / 1e9,
        "batch_shape_mean": {"K": float(a[:, 0].mean()),
                             "E0": float(a[:, 1].mean()),
                             "E1": float(a[:, 2].mean())},
    }
    if live is not None:
        ops = live["mfma_flops_per_step"]
        out["achieved"] = ops / per_step / 1e12
        out["frac"] = out["frac_mfma_ops"] = \
            ops / per_step / 1e12 / FP32_MFMA_PEAK_TF
        out["mfma_gflop_per_step"] = ops / 1e9
        out["pmc"] = live
    return out


def run_train(args, torch, dev, rank, world, dist):
    fpg = args.frames_per_gpu
    elapsed, ar_ms, tr, cfg, shapes, out = train_measure(
        torch, dev, rank, world, dist, args.config, args.preset, args.steps,
        args.warmup, args.frames, fpg,
        "off" if args.no_pipeline else args.train_loader,
        not args.train_sync_loss)
    if rank == 0:
        res = {
            "metric": "training frames/sec (%s, fwd+loss+bwd+allreduce+SGD, "
                      "training graph kwargs)" % args.config,
            "value": world * fpg * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s training step, %d frames/GPU/step (global "
                            "batch %d), graph build included%s, synthetic labels"
                            % (args.config, fpg, world * fpg,
                               "" if args.no_pipeline else
                               " (batches built by a loader thread on a "
                               "second stream, two ahead)"
                               if args.train_loader == "thread" else
                               " (next batch built on a second stream by the "
                               "stepping thread)"
                               if args.train_loader == "stream" else
                               " -- DIAGNOSTIC: batches built once, no build "
                               "in the loop"),
                "losses": "read before the next step is queued"
                          if args.train_sync_loss else
                          "every step's losses read inside the timed region, "
                          "one step late (train_step(deferred=True))",
                "last_batch_shape": dict(zip(("K", "E0", "E1"), shapes[-1])),
                "params": int(tr.flat.numel()),
                "allreduce_bytes": int(tr.flat.numel()) * 4,
                # device time between the events that bracket the gradient
                # all-reduce (+ the loss sums, one RCCL group), measured on six
                # steps AFTER the timed region; includes any wait for the
                # slowest rank to arrive; at world 1 no call is made inside
                # the step: `collective_fixed_cost` is the forced world-1 call
                "allreduce_ms": ar_ms,
                "collective": tr.collective,
                "distributed": dist_info(dist, world),
                "last_loss": {k: out[k] for k in ('cls_loss', 'loc_loss',
                                                  'reg_loss')},
                "parallelism": "dp%d (frames sharded, one flat gradient "
                               "all-reduce per step)" % world},
        }
        live = None
        if world == 1:
            res["config"]["collective_fixed_cost"] = collective_fixed_cost(
                torch, dev, int(tr.flat.numel()))
        if world == 1 and not args.no_live_pmc:
            torch.cuda.synchronize()
            live = live_pmc_train_mfma_ops(args.config, args.preset, fpg,
                                           frames=args.frames)
        res["roofline"] = train_roofline(cfg, shapes, fpg, elapsed, args.steps,
                                         live)
        print(json.dumps(res), flush=True)


def executed_flops_per_frame(cfg, n_k, n_e0, n_e1):
    """FLOPs the kernels actually issue: like the algorithmic count, but the
    first edge layer is evaluated per vertex (P = [h,x]W1+b, Q = x'Wx:
    2*(C+3)*C + 2*3*C per vertex instead of 2*(C+3)*C per edge)."""
    total = algorithmic_flops_per_frame(cfg, n_k, n_e0, n_e1)
    dim = None
    for lc in cfg['model_kwargs']['layer_configs'][:-1]:
        kw = lc['kwargs']
        if lc['type'] == 'scatter_max_point_set_pooling':
            dim = kw['output_MLP_depth_list'][-1]
        else:
            w1 = kw['edge_MLP_depth_list'][0]
            total -= 2 * (dim + 3) * w1 * n_e1
            total += (2 * (dim + 3) * w1 + 2 * 3 * w1) * n_k
    return total


def pool_statistics(cfg, shapes):
    """mean/min/max of K, E0, E1 and the mean FLOPs per frame over exactly the
    frames of the timed region (`shapes` = engine.frame_shapes)."""
    a = np.asarray(shapes, dtype=np.float64)
    alg = [algorithmic_flops_per_frame(cfg, *map(int, r[:3])) for r in a]
    exe = [executed_flops_per_frame(cfg, *map(int, r[:3])) for r in a]
    stat = lambda c: {"mean": float(a[:, c].mean()), "min": int(a[:, c].min()),
                      "max": int(a[:, c].max())}
    return {"frames": int(a.shape[0]), "K": stat(0), "E0": stat(1),
            "E1": stat(2), "alg_flops_mean": float(np.mean(alg)),
            "exe_flops_mean": float(np.mean(exe))}


# frame schedule per inference config (A/B on one MI355X): whole
# frames on 3 streams for car_auto_T3 (307 vs 300 frames/s with the split);
# one GNN stream + two builder streams for ped_cyl_auto_T3 (135 vs 130: its
# LDS-tile pooling kernel and the C = 256 edge kernel of different frames
# start on each other's tails when GNN streams overlap)
SCHEDULES = {"car_auto_T3": 3, "car_auto_T2": 3, "car_auto_T1": 3,
             "car_auto_T0": 3, "ped_cyl_auto_T3": 0}


def frame_streams_for(args, config_name):
    if args.frame_streams >= 0:
        return args.frame_streams
    return SCHEDULES.get(config_name, 3)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="car_auto_T3")
    ap.add_argument("--preset", default=None,
                    help="synthetic cloud preset (default: car_600k, the "
                         "north-star ~20k-point / ~600k-edge shape; ped_dense "
                         "for ped_cyl configs)")
    ap.add_argument("--frames", type=int, default=8,
                    help="distinct synthetic frames in the pool")
    ap.add_argument("--frames-per-step", type=int, default=8,
                    help="inference: frames every GPU processes per step "
                         "(the same at every N: weak scaling).  8 makes the "
                         "timed region of a 20-step run ~0.5 s per rank, so "
                         "a few ms of rank skew at the closing barrier is "
                         "<1 %% of it")
    ap.add_argument("--repeats", type=int, default=5,
                    help="barrier-bracketed timed regions of exactly --steps "
                         "steps; the FIRST is the headline `value`, the "
                         "spread of all of them is reported beside it")
    ap.add_argument("--edge-arith", choices=("f32", "bf16x3", "f16x2"), default="f32",
                    help="arithmetic of the per-edge 300x300 product: f32 = "
                         "fp32 MFMA (the headline); bf16x3 = the SECONDARY "
                         "split-bf16 kernel (csrc/edge_ws_bf16.h): the line's "
                         "dtype and metric then say so")
    ap.add_argument("--no-bind", action="store_true",
                    help="do not pin this rank to the CPUs of its GPU's "
                         "NUMA node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not spawn the two rocprofv3 --pmc child runs "
                         "that measure roofline.traffic (uses the committed "
                         "PMC pass of the same workload instead, if any)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra `car`-preset measurement")
    ap.add_argument("--train-loader", default="thread",
                    choices=["thread", "stream", "prebuilt"],
                    help="--train: 'thread' = batches come from a loader "
                         "thread with its own stream, two ahead "
                         "(train.BatchPrefetcher); 'stream' = the stepping "
                         "thread builds the next batch on a second stream once "
                         "the step is enqueued; 'prebuilt' = diagnostic, no "
                         "build in the loop")
    ap.add_argument("--train-sync-loss", action="store_true",
                    help="--train: read every step's losses before the next "
                         "step is queued (default: one step late)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run frames strictly sequentially on one stream")
    ap.add_argument("--no-capture", action="store_true",
                    help="skip the hipGraph capture / replay latency entry")
    ap.add_argument("--gnn-priority", type=int, default=0,
                    help="experiment: +1 = message passing of the frame "
                         "streams on high-priority partner streams, -1 = the "
                         "graph builds on them")
    ap.add_argument("--no-capture-overlap", action="store_true",
                    help="skip the capture of the frame with the overlapped "
                         "graph build (side streams inside the hipGraph)")
    ap.add_argument("--host-sized", action="store_true",
                    help="build the graphs with host-read sizes (two host "
                         "waits per frame) instead of the capacity form")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--e2e", action="store_true",
                    help="only the files-to-files frame loop (secondary_e2e)")
    ap.add_argument("--e2e-bias", type=float, default=None)
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 4: training step instead of inference")
    ap.add_argument("--frames-per-gpu", type=int, default=2)
    ap.add_argument("--compute-streams", type=int, default=1,
                    help="GNN streams of the frame pipeline (1 or 2)")
    ap.add_argument("--frame-streams", type=int, default=-1,
                    help="capacity form: streams that take whole frames "
                         "(build + GNN) round-robin; 0 = the builder / "
                         "compute split (--graph-streams, --compute-streams); "
                         "-1 = the schedule measured best for the config "
                         "(SCHEDULES)")
    ap.add_argument("--graph-streams", type=int, default=2,
                    help="streams the graphs of consecutive frames are built "
                         "on (capacity form only; 1 = a single builder)")
    ap.add_argument("--lookahead", type=int, default=0,
                    help="frames the graph-builder thread may run ahead of the "
                         "GNN (0 = build inline on the calling thread)")
    ap.add_argument("--graph-cus", type=int, default=0,
                    help="CUs reserved for the graph-build stream of the frame "
                         "pipeline (CU-masked streams; 0 = shared device)")
    ap.add_argument("--tune", action="append", default=[],
                    help="key=value library tunable (experiments; repeatable)")
    args = ap.parse_args(argv)
    if args.preset is None:
        args.preset = "ped_dense" if args.config.startswith("ped") else (
            "car" if args.train else "car_600k")
    return args


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher around it: become the
    launcher.  One process per GPU via torch.distributed.run on 127.0.0.1;
    rank 0 of the children prints the JSON line."""
    import subprocess
    if not STUB:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and not (ONE_GPU and have >= 1):
            raise SystemExit(
                "bench.py: --gpus %d but only %d GPU(s) visible; refusing to "
                "report an n_gpus it did not run on" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


# PGNN_BENCH_STUB=1: the launcher / rank plumbing / JSON contract exercised on
# CPU with gloo and a sleep in place of the engine (tests/test_sharding_cpu.py).
# The line it prints says "data": "stub" and carries no rates.
STUB = os.environ.get("PGNN_BENCH_STUB") == "1"
# PGNN_BENCH_ONE_GPU=1 (tests/test_gpu_multirank.py on a one-GPU box): every
# rank runs the REAL engine on cuda:0 and the rank plumbing uses gloo.  The
# line says so in "data"; its value is not a multi-GPU measurement.
ONE_GPU = os.environ.get("PGNN_BENCH_ONE_GPU") == "1"


class _StubEngine(object):
    def __init__(self):
        self.frame_shapes = []

    def run_frames(self, frames):
        for f in frames:
            time.sleep(0.002)
            self.frame_shapes.append((100 + f, 1000 + f, 2000 + f))


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


# where bind_rank_to_numa reads PCI / NUMA topology (tests point it at a tree
# of their own: tests/test_sharding_cpu.py fakes an 8-GPU, 2-socket node)
SYS_ROOT = os.environ.get("PGNN_SYS_ROOT", "/sys")


def bind_rank_to_numa(torch, local_rank, local_world):
    """Pin this process to the CPUs next to its GPU: the cpulist of the NUMA
    node /sys reports for the GPU's PCI function; when the platform reports
    no node (-1: one-socket hosts, VMs) the allowed CPUs are split evenly by
    local rank instead, so 8 ranks' enqueue threads (~110 launches per frame
    each) never share cores.  Returns what was done, for the JSON line."""
    info = {"bound": False}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        node = -1
        bus = None
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id,
                                        pr.pci_device_id)
            with open("%s/bus/pci/devices/%s/numa_node" % (SYS_ROOT, bus)) as fh:
                node = int(fh.read().strip())
        except Exception:
            pass
        info["pci"] = bus
        info["numa_node"] = node
        cpus = None
        if node >= 0:
            with open("%s/devices/system/node/node%d/cpulist"
                      % (SYS_ROOT, node)) as fh:
                cpus = sorted(_parse_cpulist(fh.read()) & set(allowed))
            # ranks whose GPUs share the node split its CPUs
            peers = []
            for r in range(local_world):
                try:
                    q = torch.cuda.get_device_properties(r)
                    with open("%s/bus/pci/devices/%04x:%02x:%02x.0/numa_node"
                              % (SYS_ROOT, q.pci_domain_id, q.pci_bus_id,
                                 q.pci_device_id)) as fh:
                        if int(fh.read().strip()) == node:
                            peers.append(r)
                except Exception:
                    pass
            if local_rank in peers and len(peers) > 1 and \
                    len(cpus) >= len(peers):
                i, n = peers.index(local_rank), len(peers)
                cpus = cpus[i * len(cpus) // n:(i + 1) * len(cpus) // n]
            info["policy"] = "cpulist of the GPU's NUMA node / ranks on it"
        if not cpus:
            n = max(1, local_world)
            cpus = allowed[local_rank * len(allowed) // n:
                           (local_rank + 1) * len(allowed) // n]
            info["policy"] = "no NUMA node reported: allowed CPUs split " \
                             "evenly by local rank"
        if cpus:
            os.sched_setaffinity(0, cpus)
            info.update(bound=True, n_cpus=len(cpus), first_cpu=cpus[0],
                        last_cpu=cpus[-1])
    except Exception as exc:      # never fatal to the measurement
        info["error"] = repr(exc)
    return info


def init_ranks(args, torch):
    """(rank, world, local_rank, device, dist).  Ranks exist before this
    point: created by the driver's torch.distributed.run or by self_launch."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if STUB:
        dev = torch.device("cpu")
        backend = "gloo"
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the hot path has no CPU "
                             "fallback)")
        if ONE_GPU:
            local_rank_dev = 0
            backend = "gloo"
        else:
            local_rank_dev = local_rank
            backend = "nccl"
        torch.cuda.set_device(local_rank_dev)
        dev = torch.device("cuda", local_rank_dev)
        args.cpu_affinity = {"bound": False, "policy": "--no-bind"} \
            if args.no_bind else bind_rank_to_numa(
                torch, local_rank,
                int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world > 1:
        import torch.distributed as dist
        if STUB or ONE_GPU:
            dist.init_process_group(backend)
        else:
            dist.init_process_group(backend, device_id=dev)
        assert dist.get_world_size() == world
    return rank, world, local_rank, dev, dist


def _sync(torch, dist):
    if not STUB:
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if not STUB:
        torch.cuda.synchronize()


def _max_over_ranks(torch, dist, dev, values):
    if dist is None:
        return values
    t = torch.tensor(values, dtype=torch.float64,
                     device="cpu" if ONE_GPU else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def _gather_shapes(torch, dist, dev, shapes, world):
    """All ranks' per-frame (K, E0, E1) rows on rank 0 (sizes only, after the
    timed region)."""
    a = torch.tensor(shapes, dtype=torch.int64,
                     device="cpu" if ONE_GPU else dev).reshape(-1, 3)
    if dist is None:
        return a.tolist()
    out = [torch.zeros_like(a) for _ in range(world)]
    dist.all_gather(out, a)
    return torch.cat(out).tolist()


def dist_info(dist, world):
    if dist is None:
        return {"world_size": 1, "backend": None}
    return {"world_size": int(dist.get_world_size()),
            "backend": dist.get_backend() + (
                " (RCCL)" if dist.get_backend() == "nccl" else "")}


def run_stub(args, torch, dev, rank, world, dist):
    eng = _StubEngine()
    fps = max(1, args.frames_per_step)
    reps = max(1, args.repeats)
    per_region = args.steps * fps
    ids = list(range(rank, world * (args.warmup * fps + reps * per_region),
                     world))
    eng.run_frames(ids[:args.warmup * fps])
    regions, shapes = [], None
    for r in range(reps):
        lo = args.warmup * fps + r * per_region
        eng.frame_shapes = []
        _sync(torch, dist)
        t0 = time.perf_counter()
        eng.run_frames(ids[lo:lo + per_region])
        _sync(torch, dist)
        el, = _max_over_ranks(torch, dist, dev, [time.perf_counter() - t0])
        regions.append(el)
        if r == 0:
            shapes = _gather_shapes(torch, dist, dev, eng.frame_shapes, world)
    elapsed = regions[0]
    rep_ms = [r / args.steps * 1e3 for r in regions]
    if rank == 0:
        print(json.dumps({
            "metric": "stub frames/sec (launcher self-test, no kernels)",
            "value": world * per_region / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "none", "data": "stub",
            "config": {"workload": "stub", "frames_timed": len(shapes),
                       "frames_per_gpu_per_step": fps,
                       "timed_region_s": elapsed,
                       "repeat_ms_per_step": {
                           "n": len(rep_ms), "min": min(rep_ms),
                           "median": float(np.median(rep_ms)),
                           "max": max(rep_ms), "all": rep_ms},
                       "distributed": dist_info(dist, world)}}), flush=True)


def secondary_ped(args, torch, dev, measure):
    """BASELINE config 5 (`ped_cyl_auto_T3`, dense scan: ~50k points, small
    radii, C = 256): >= 10 pipelined frames of preset `ped_dense` plus the MFMA
    rooflines of its edge and pooling kernels on the pool's first frame."""
    from pointgnn_amd import configs, weights
    from pointgnn_amd.engine import InferenceEngine
    cfg = configs.get_config("ped_cyl_auto_T3")
    params = weights.init_params(cfg, seed=0, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    eng.config_name = "ped_cyl_auto_T3"
    # (the pool of `--config ped_cyl_auto_T3`: 8 seeded frames; 24 frames in
    # flight through the pipeline = 0.17 s, its fill and drain under 5 %)
    steps = max(16, min(24, args.steps))
    elapsed, shapes, pool = measure("ped_dense", steps, 3, engine=eng,
                                    n_frames=8)
    st = pool_statistics(cfg, shapes)
    out = {
        "workload": "ped_cyl_auto_T3 inference, preset 'ped_dense' (8 seeded "
                    "frames cycled), 1 frame/step; schedule: %s" % (
                        "%d whole-frame streams" % frame_streams_for(
                            args, "ped_cyl_auto_T3")
                        if frame_streams_for(args, "ped_cyl_auto_T3") > 0 and
                        not args.host_sized else
                        "%d GNN + %d builder stream(s)" % (
                            args.compute_streams,
                            1 if args.host_sized else args.graph_streams)),
        "steps": steps, "frames_per_sec": steps / elapsed,
        "ms_per_frame": elapsed / steps * 1e3, "N": 50000,
        "K": st["K"], "E0": st["E0"], "E1": st["E1"],
        "algorithmic_gflop_per_frame": st["alg_flops_mean"] / 1e9,
        "executed_gflop_per_frame": st["exe_flops_mean"] / 1e9,
        "algorithmic_tflops": st["alg_flops_mean"] * steps / elapsed / 1e12,
        "executed_frac_of_fp32_mfma_peak":
            st["exe_flops_mean"] * steps / elapsed / 1e12 / FP32_MFMA_PEAK_TF,
    }
    if not args.no_roofline:
        first = sorted(pool)[0]
        x, f = pool[first][:2]
        eng.run_frame(x, f)
        coords, kps, edges = eng.last_graph
        n_k = int(coords[1].shape[0])
        mf = roofline_edge_kernel(torch, eng, edges[1], n_k, frame=(x, f))
        if mf is not None:
            mf["workload"] = {"frame_seed": first,
                              "E": int(edges[1].shape[0]), "K": n_k}
            out["roofline_mfma"] = mf
        pl = roofline_pool_kernel(torch, eng, frame=(x, f))
        if pl is not None:
            out["roofline_pool"] = pl
    # the same frames with the edge stage on the 16-bit matrix-pipe kernels
    # (C = 256 instances) -- SECONDARY, as in the headline
    for arith in ("bf16x3", "f16x2"):
        eng.model.edge_arith = arith
        try:
            eng.frame_shapes = []
            e16, _, _ = measure("ped_dense", steps, 3, engine=eng, n_frames=8)
            eng.check_edge_range()
            b16 = {"frames_per_sec": steps / e16,
                   "ms_per_frame": e16 / steps * 1e3,
                   "vs_f32": elapsed / e16,
                   "dtype": arith + " products, f32 accumulate (edge stage "
                                    "only)"}
            if not args.no_roofline:
                eng.run_frame(x, f)   # host-sized: the image is cached by now
                rf = roofline_edge_kernel_16bit(torch, eng, edges[1], n_k,
                                                arith, frame=(x, f))
                if rf is not None:
                    rf["workload"] = out["roofline_mfma"]["workload"]
                    b16["roofline"] = rf
            out[arith] = b16
        finally:
            eng.model.edge_arith = "f32"
    return out


def secondary_e2e(args, torch, dev, headline_fps=None, n_files=8, passes=20,
                  background_bias=None):
    """The drop-in frame loop (run.py:203-433 -> pointgnn_amd.run.run_dataset's
    FramePipeline) from FILES to FILES: a synthetic KITTI-object tree (the
    headline preset's scenes as 115 k-point velodyne sweeps + calib + image
    headers, written by pointgnn_amd.synthetic.write_kitti_frames) -> crop ->
    graph -> GNN -> softmax -> decode + NMS -> KITTI rows -> txt, frames kept in
    flight.  Timed: `passes` passes over the `n_files` files (page cache) by
    one FramePipeline.run(), wall clock, after one warm-up pass.

    Synthetic weights classify at random; a trained model marks a few percent
    of the vertices as objects.  `background_bias` is added to the background
    logit's bias so that the candidate rate is a trained model's (the line
    reports candidates / kept boxes / rows per frame): the NMS and the
    host-side rows stage then do a real frame's amount of work."""
    import shutil
    import tempfile
    from pointgnn_amd import configs, kitti_dataset, run as RUN, weights
    from pointgnn_amd.synthetic import write_kitti_frames
    cfg = configs.get_config("car_auto_T3")
    params = weights.init_params(cfg, seed=0, bias_scale=0.05)
    bias_name = "output/predictor/cls/fully_connected_1/biases"
    if background_bias is None:
        background_bias = E2E_BACKGROUND_BIAS
    params[bias_name] = params[bias_name].copy()
    params[bias_name][0] += background_bias
    root = tempfile.mkdtemp(prefix="pgnn_e2e_")
    try:
        dirs = write_kitti_frames(root, range(n_files), preset="car_600k")
        ds = kitti_dataset.KittiDataset(*dirs)
        model = RUN.build_model(cfg, params=params)
        out_dir = os.path.join(root, "out")

        def run(indices, seq=False):
            td = {}
            t0 = time.perf_counter()
            if seq:
                for i in indices:
                    RUN._frame_sequential(ds, i, model, cfg, out_dir, True,
                                          True, td, None, None)
                pipe = None
            else:
                pipe = RUN.FramePipeline(ds, cfg, model, out_dir, True, True,
                                         None, None, 3, 4, td)
                pipe.run(list(indices))
            torch.cuda.synchronize()
            return time.perf_counter() - t0, td, pipe
        files = list(range(n_files))
        run(files)                                  # warm-up pass
        n = n_files * passes
        wall, td, pipe = run(files * passes)
        seq_wall, seq_td, _ = run(files * 2, seq=True)
        names = ('fetch input', 'gen graph', 'gnn inference',
                 'decode box + nms', 'kitti rows', 'write txt')
        res = {
            "workload": "run_dataset's frame loop, files to files: %d "
                        "synthetic KITTI frames (car_600k scenes as 115 k-point "
                        "velodyne sweeps, %.1f MB each) x %d passes, "
                        "car_auto_T3, 3 frames in flight, loader + writer "
                        "threads; background-logit bias %+.1f on the synthetic "
                        "weights" % (n_files, 115000 * 16 / 1e6, passes,
                                     background_bias),
            "frames": n, "frames_per_sec": n / wall,
            "ms_per_frame": wall / n * 1e3,
            "vs_headline": (n / wall / headline_fps) if headline_fps else None,
            "phase_ms_per_frame": {k: td.get(k, 0.0) / n * 1e3 for k in names},
            "phase_note": "device stages: time between events on the frame's "
                          "stream (frames overlap: the sum exceeds "
                          "ms_per_frame); host stages: wall time in their "
                          "thread",
            "per_frame": {k: v / max(1, n - 1) for k, v in pipe.stats.items()},
            "sequential_fallbacks": pipe.fallbacks,
            "sequential_loop": {
                "frames_per_sec": 2 * n_files / seq_wall,
                "ms_per_frame": seq_wall / (2 * n_files) * 1e3,
                "phase_ms_per_frame": {
                    k: seq_td.get(k, 0.0) / (2 * n_files) * 1e3
                    for k in names}},
        }
        return res
    finally:
        shutil.rmtree(root, ignore_errors=True)


# added to the background logit's bias of the synthetic weights in the
# files-to-files loop (secondary_e2e): ~3 % of the vertices (~190 candidates, ~14 kept boxes, ~10 rows per frame) then pass
# run.py:266-290's prob > 1/nc test, a trained model's rate
E2E_BACKGROUND_BIAS = 2.0


def secondary_train(args, torch, dev):
    """BASELINE config 4 on one GPU (`car_auto_T3` training step, 2 frames per
    step, training graph kwargs): >= 10 timed steps and the whole-step MFMA
    roofline.  The 8-GPU form is `bench.py --train --gpus 8`."""
    # (48-64 steps = 0.15-0.2 s: a 20-step region is 60 ms, in which one slow
    # phase of the loader thread against the stepping thread reads as 5 %)
    steps = max(48, min(64, args.steps))
    fpg = 2
    # (8 warm-up steps: the first few size the trainer's workspace and the
    # allocator's pools for the graph tensors)
    elapsed, ar_ms, tr, cfg, shapes, out = train_measure(
        torch, dev, 0, 1, None, "car_auto_T3", "car", steps, 8, 4, fpg, True)
    res = {
        "workload": "car_auto_T3 training step, %d frames/step, training "
                    "graph kwargs (voxel 0.8, random keypoints + jitter, "
                    "fan-in cap 256), graph build included (batches built "
                    "by a loader thread on a second stream, two ahead), every step's losses read inside the "
                    "timed region one step late, synthetic labels, preset "
                    "'car'" % fpg,
        "steps": steps, "ms_per_step": elapsed / steps * 1e3,
        "training_frames_per_sec": fpg * steps / elapsed,
        "params": int(tr.flat.numel()),
        "last_loss": {k: out[k] for k in ('cls_loss', 'loc_loss', 'reg_loss')},
    }
    n_params = int(tr.flat.numel())
    del tr
    torch.cuda.empty_cache()
    # the same loop over the headline's 8-frame pool (frames 4-7 carry smaller
    # training graphs than 0-3: what `bench.py --train` runs by default)
    el8, _, tr8, _, _, _ = train_measure(
        torch, dev, 0, 1, None, "car_auto_T3", "car", steps, 8, 8, fpg, True)
    res["ms_per_step_8frame_pool"] = el8 / steps * 1e3
    del tr8
    torch.cuda.empty_cache()
    # the step's collectives on one GPU: their fixed cost alone, and the same
    # timed loop with a world-1 RCCL communicator and every collective issued
    # (counts before the forward, gradient + sums behind the backward)
    fixed = collective_fixed_cost(torch, dev, n_params)
    el_f, ar_f, tr_f, _, _, out_f = train_measure(
        torch, dev, 0, 1, None, "car_auto_T3", "car", steps, 8, 4, fpg, True,
        force_collective=True)
    res.update({
        "allreduce_ms": fixed["allreduce_ms"],
        "counts_allreduce_ms": fixed["counts_allreduce_ms"],
        "collective_fixed_cost": fixed,
        "collectives_forced_world1": {
            "collective": tr_f.collective,
            "ms_per_step": el_f / steps * 1e3,
            "allreduce_ms_inside_step": ar_f,
            "last_loss": {k: out_f[k] for k in ('cls_loss', 'loc_loss',
                                                'reg_loss')}},
    })
    del tr_f
    torch.cuda.empty_cache()
    live = None
    if not args.no_live_pmc:
        torch.cuda.synchronize()
        live = live_pmc_train_mfma_ops("car_auto_T3", "car", fpg, frames=4)
    res["roofline"] = train_roofline(cfg, shapes, fpg, elapsed, steps, live)
    return res


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))

    import torch
    rank, world, local_rank, dev, dist = init_ranks(args, torch)
    if STUB:
        run_stub(args, torch, dev, rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import pointgnn_amd  # noqa: F401
    from pointgnn_amd import configs, weights
    from pointgnn_amd.engine import InferenceEngine, shard_frames
    from pointgnn_amd.synthetic import synthetic_cloud
    for kv in args.tune:
        from pointgnn_amd import _lib as _pg_lib
        key, val = kv.split("=")
        _pg_lib.set_tunable(key, int(val))
    if args.train:
        run_train(args, torch, dev, rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.e2e:
        print(json.dumps({"secondary_e2e": secondary_e2e(
            args, torch, dev, background_bias=args.e2e_bias)}), flush=True)
        return

    cfg = configs.get_config(args.config)
    params = weights.init_params(cfg, seed=0, bias_scale=0.05)
    engine = InferenceEngine(cfg, params, device=dev,
                             edge_arith=args.edge_arith)
    engine.config_name = args.config

    def measure(preset, steps, warmup, engine=engine, n_frames=None, fps=1,
                repeats=1):
        """Time `steps` steps of `fps` frames per rank of `preset`, `repeats`
        times (each region bracketed by barrier + synchronize); returns
        (elapsed of the FIRST region, max over ranks; all ranks' per-frame
        shapes of that region; this rank's pool; every region's elapsed)."""
        n_frames = n_frames or args.frames
        per_region = steps * fps
        total = warmup * fps + repeats * per_region
        # frame stream: rank r owns frames r, r+W, ...
        my_ids = shard_frames(world * total, rank, world)

        def seed_of(i):
            # every rank cycles through ALL pool frames (rank r starts at
            # frame r), so the per-GPU work is the same mix at every N (weak
            # scaling); with `id % frames` rank 0 of 8 would see frame 0 only
            return (my_ids[i] // world + rank) % n_frames
        pool = {}
        for s in sorted({seed_of(i) for i in range(len(my_ids))}):
            xyz, inten = synthetic_cloud(seed=s, preset=preset)
            pool[s] = (torch.from_numpy(xyz).to(dev),
                       torch.from_numpy(inten).to(dev), xyz, inten)

        def run(lo, hi):
            """Frames lo..hi-1 of this rank's stream.  Default: software
            pipeline on HIP streams (graph build of frame i+1 overlaps the GNN
            of frame i); --no-pipeline runs them strictly one after the other."""
            fr = [pool[seed_of(i)][:2] for i in range(lo, hi)]
            deferred = not args.host_sized
            if args.no_pipeline:
                if deferred:
                    outs = [engine.run_frame_deferred(x, f) for x, f in fr]
                    return [d.result() for d in outs][-1]
                out = None
                for x, f in fr:
                    out = engine.run_frame(x, f)
                return out
            n_fs = frame_streams_for(args, engine.config_name)
            if deferred and n_fs > 0 and args.graph_cus <= 0:
                return engine.run_frames_on_streams(
                    fr, n_fs, gnn_priority=args.gnn_priority)[-1]
            return engine.run_frames_pipelined(
                fr, compute_streams=args.compute_streams,
                graph_cus=args.graph_cus, lookahead=args.lookahead,
                deferred=deferred, graph_streams=args.graph_streams)[-1]

        if warmup:
            run(0, warmup * fps)
        regions, shapes = [], None
        for r in range(repeats):
            lo = warmup * fps + r * per_region
            engine.frame_shapes = []
            _sync(torch, dist)
            t0 = time.perf_counter()
            out = run(lo, lo + per_region)
            _sync(torch, dist)
            el, = _max_over_ranks(torch, dist, dev,
                                  [time.perf_counter() - t0])
            regions.append(el)
            assert torch.isfinite(out[0]).all()
            assert len(engine.frame_shapes) == per_region
            if r == 0:
                shapes = _gather_shapes(
                    torch, dist, dev, [s[:3] for s in engine.frame_shapes],
                    world)
        measure.regions = regions
        # Python time to ENQUEUE one frame (~110 ctypes launches: graph build
        # + GNN), nothing read back: 16 frames onto an idle device -- far
        # below any queue depth, so the host never waits for the device here
        measure.enqueue_ms = None
        if not args.no_pipeline and not args.host_sized and \
                frame_streams_for(args, engine.config_name) > 0 and \
                args.graph_cus <= 0:
            n_enq = min(16, per_region)
            samples = []
            for _ in range(3):
                _sync(torch, None)
                run(warmup * fps, warmup * fps + n_enq)
                samples.append(engine.last_enqueue_s / n_enq * 1e3)
            measure.enqueue_ms = _max_over_ranks(
                torch, dist, dev, [float(np.median(samples))])[0]
            engine.frame_shapes = []
        return regions[0], shapes, pool

    fps_h = max(1, args.frames_per_step)
    elapsed, shapes, pool = measure(args.preset, args.steps, args.warmup,
                                    fps=fps_h, repeats=max(1, args.repeats))
    regions_h, enqueue_ms_h = list(measure.regions), measure.enqueue_ms
    second = None
    if world == 1 and not args.no_secondary and args.preset == "car_600k":
        s2 = max(8, args.steps // 2)
        e2, sh2, _ = measure("car", s2, 4)
        st2 = pool_statistics(cfg, sh2)
        second = {
            "workload": "%s inference, preset 'car' (round-1 headline shape)"
                        % args.config,
            "steps": s2, "frames_per_sec": s2 / e2, "ms_per_frame": e2 / s2 * 1e3,
            "K": st2["K"], "E0": st2["E0"], "E1": st2["E1"],
            "algorithmic_gflop_per_frame": st2["alg_flops_mean"] / 1e9,
            "algorithmic_tflops": st2["alg_flops_mean"] * s2 / e2 / 1e12}

    # BASELINE configs 5 and 4 in the same line (single-GPU runs of the
    # headline command only; `--config ped_cyl_auto_T3` / `--train` are the
    # full-length forms): the ped_cyl dense-scan stress and the training step
    ped = trn = e2e = None
    if world == 1 and not args.no_secondary and args.preset == "car_600k" \
            and args.config == "car_auto_T3":
        ped = secondary_ped(args, torch, dev, measure)
        trn = secondary_train(args, torch, dev)
        e2e = secondary_e2e(args, torch, dev,
                            headline_fps=args.steps * fps_h / elapsed)

    # SECONDARY arithmetics (not the headline, whose dtype is f32): the same
    # frames with the per-edge product on the matrix pipe's 16-bit formats --
    # 'bf16x3' (both operands split exactly into three bf16 parts, six
    # products; csrc/edge_ws_bf16.h) and 'f16x2' (two fp16 parts, 22
    # significand bits, three products; csrc/edge_ws_f16.h) -- next to how far
    # their logits are from the fp32-MFMA path's on the pool's first frame
    sec16 = {}
    if world == 1 and not args.no_secondary and args.edge_arith == "f32" \
            and args.preset == "car_600k" and args.config == "car_auto_T3":
        first = sorted(pool)[0]
        x0_, f0_ = pool[first][:2]
        lg32, bx32 = [t.clone() for t in engine.run_frame(x0_, f0_)]
        what = {
            "bf16x3": ("split-bf16 kernel (3 x 3 bf16 parts, the 6 products of "
                       "order <= 2, fp32 accumulation; fp32-MFMA everywhere "
                       "else)",
                       "bf16x3 split products, f32 accumulate (edge stage only)",
                       "its distance is not larger than the fp32 path's"),
            "f16x2": ("two-part fp16 kernel (both operands as x0 + x1 / 2^11 "
                      "in fp16 = 22 significand bits, 3 products, fp32 "
                      "accumulation) and the pooling stage's 64->128->300 "
                      "layers in the same representation (fp32-MFMA "
                      "everywhere else)",
                      "f16x2 two-part products, f32 accumulate (edge stage + "
                      "the wide layers of the pooling stage)",
                      "same bars: its distance grows by a few per cent; "
                      "activations are clamped at 65504 and flagged from "
                      "32768 on"),
        }
        for arith in ("bf16x3", "f16x2"):
            engine.frame_shapes = []
            engine.model.edge_arith = arith
            try:
                lg16, bx16 = [t.clone() for t in engine.run_frame(x0_, f0_)]
                s3 = max(8, args.steps // 2)
                e3, sh3, _ = measure(args.preset, s3, 2, fps=fps_h)
                engine.check_edge_range()
                rf16 = rp16 = None
                if not args.no_roofline:
                    # (a host-sized frame: after `measure` last_graph holds
                    # the capacity form, whose rows behind the counts are not
                    # a graph)
                    engine.run_frame(x0_, f0_)
                    coords_, _, edges_ = engine.last_graph
                    rf16 = roofline_edge_kernel_16bit(
                        torch, engine, edges_[1], int(coords_[1].shape[0]),
                        arith, frame=(x0_, f0_))
                    if rf16 is not None:
                        rf16["workload"] = {"frame_seed": first,
                                            "E": int(edges_[1].shape[0]),
                                            "K": int(coords_[1].shape[0])}
                    if arith == "f16x2":
                        rp16 = roofline_pool_kernel_f16x2(torch, engine,
                                                          frame=(x0_, f0_))
            finally:
                engine.model.edge_arith = "f32"
            engine.frame_shapes = []
            sec16[arith] = {
                "workload": "%s inference, preset '%s', edge stage on the %s"
                            % (args.config, args.preset, what[arith][0]),
                "dtype": what[arith][1],
                "steps": s3, "frames_per_gpu_per_step": fps_h,
                "frames_per_sec": s3 * fps_h / e3,
                "ms_per_frame": e3 / (s3 * fps_h) * 1e3,
                "vs_f32_headline": (s3 * fps_h / e3) /
                                   (args.steps * fps_h / elapsed),
                "max_abs_dlogit_vs_f32_path_frame_seed%d" % first:
                    float((lg16 - lg32).abs().max()),
                "max_abs_dbox_vs_f32_path_frame_seed%d" % first:
                    float((bx16 - bx32).abs().max()),
                "note": "SECONDARY: the headline `value` and `dtype` are the "
                        "fp32-MFMA path; tests/test_gpu_bf16x3.py and the "
                        "edge_arith-parametrised parity tests hold this path "
                        "to the float64 oracle and the reference's TF graphs "
                        "(%s)" % what[arith][2],
            }
            if rf16 is not None:
                sec16[arith]["roofline"] = rf16
            if rp16 is not None:
                sec16[arith]["roofline_pool"] = rp16

    if rank == 0:
        # per-phase wall clock of the pool's first frame (outside the timed
        # region; run.py's phase names) and that frame's graph for the
        # standalone kernel measurements
        first = sorted(pool)[0]
        x, f, xyz_np, inten_np = pool[first]
        engine.time_dict = {}
        for _ in range(3):
            engine.run_frame(x, f, timed=True)
        coords, kps, edges = engine.last_graph
        n_k = int(coords[1].shape[0])
        n_e1 = int(edges[1].shape[0])
        frames = engine.time_dict['frames']
        # one frame alone, enqueue to results on the host: host-sized graph
        # (the builder waits twice for sizes) vs capacity form (one read, at
        # the end)
        lat = {"host-sized": [], "capacity form": [],
               "capacity form, overlapped build": []}
        captured = captured_ov = capture_note = None
        if not args.no_capture:
            lat["capacity form, one hipGraph"] = []
            captured = engine.capture_frame(x, f)
            if not args.no_capture_overlap:
                try:
                    captured_ov = engine.capture_frame(x, f,
                                                       overlap_build=True)
                    lat["capacity form, overlapped build, one hipGraph"] = []
                except Exception as exc:   # reported, never fatal to the line
                    capture_note = "overlapped-build capture failed: %r" % (
                        exc,)
        for _ in range(9):
            for key in lat:
                torch.cuda.synchronize()
                tp = time.perf_counter()
                if key == "host-sized":
                    engine.run_frame(x, f)
                elif key == "capacity form":
                    engine.run_frame_deferred(x, f).result()
                elif key == "capacity form, overlapped build":
                    engine.run_frame_deferred(x, f,
                                              overlap_build=True).result()
                elif key == "capacity form, one hipGraph":
                    captured.replay(x, f).result()
                else:
                    captured_ov.replay(x, f).result()
                torch.cuda.synchronize()
                lat[key].append((time.perf_counter() - tp) * 1e3)
        del captured, captured_ov
        lat = {k_: float(np.median(v[2:])) for k_, v in lat.items()}
        if capture_note:
            lat["note"] = capture_note
        # the graph build alone in capacity form (enqueue to idle device):
        # in order on one stream vs its independent parts on side streams
        build = {"capacity form": [], "capacity form, overlapped build": []}
        for _ in range(9):
            for key in build:
                torch.cuda.synchronize()
                tp = time.perf_counter()
                g_ = engine.build_graph_deferred(
                    x, overlap=key.endswith("overlapped build"))
                torch.cuda.synchronize()
                build[key].append((time.perf_counter() - tp) * 1e3)
                del g_
        build = {k_: float(np.median(v[2:])) for k_, v in build.items()}
        engine.run_frame(x, f)   # last_graph back to the host-sized form
        coords, kps, edges = engine.last_graph
        # run.py's last two phases ("decode box", "nms", run.py:264-326) on
        # this frame's outputs -- reported beside the metric, not part of it
        # (the metric excludes NMS)
        from pointgnn_amd import kitti_output, nms as pg_nms
        logits, box_enc = engine.run_frame(x, f)
        probs = torch.softmax(logits, dim=1)
        lmap = kitti_output.LABEL_MAPS[cfg.get('label_method', 'Car')]
        post_ms = []
        for _ in range(5):
            torch.cuda.synchronize()
            tp = time.perf_counter()
            det = pg_nms.detect_boxes(probs, box_enc, coords[-1], lmap,
                                      cfg.get('nms_overlapped_thres', 0.01))
            torch.cuda.synchronize()
            post_ms.append((time.perf_counter() - tp) * 1e3)
        post = {"ms": float(np.median(post_ms[1:])),
                "candidates": int(pg_nms.select_candidates(probs)[0].numel()),
                "kept": int(det[0].numel()),
                "note": "seeded weights give near-uniform class "
                        "probabilities: thousands of candidates, a stress "
                        "case for the NMS"}
        st = pool_statistics(cfg, shapes)
        assert st["frames"] == world * args.steps * fps_h
        n_builders = 1 if (args.host_sized or args.graph_cus > 0) else \
            max(1, args.graph_streams)
        n_fs = frame_streams_for(args, args.config)
        frame_sched = (not args.host_sized and n_fs > 0 and
                       args.graph_cus <= 0)
        fps = world * args.steps * fps_h / elapsed
        n_pts = int(x.shape[0])
        rep_ms = [r / args.steps * 1e3 for r in regions_h]
        res = {
            "metric": "KITTI-shaped frames/sec (%s inference: graph build + "
                      "GNN, %d pts/frame, mean E1 %.0fk)" % (
                          args.config, n_pts, st["E1"]["mean"] / 1e3),
            "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.edge_arith == "f32" else
                     "%s products with f32 accumulation in the edge stage "
                     "(SECONDARY arithmetic), f32 elsewhere" % args.edge_arith,
            "data": "synthetic" if not (ONE_GPU and world > 1) else
                    "synthetic; TEST MODE: %d ranks share ONE GPU under gloo "
                    "(not a multi-GPU measurement)" % world,
            "config": {
                "workload": "%s inference, %d frames/step/GPU, synthetic "
                            "HDL-64E-shaped cloud preset '%s' (%d seeded "
                            "frames cycled), seeded Xavier weights"
                            % (args.config, fps_h, args.preset, args.frames),
                # statistics over exactly the world*steps timed frames
                "N": n_pts, "K": st["K"], "E0": st["E0"], "E1": st["E1"],
                "frames_timed": st["frames"],
                "frames_per_gpu_per_step": fps_h,
                "ms_per_frame_per_gpu": elapsed / (args.steps * fps_h) * 1e3,
                "timed_region_s": elapsed,
                # every barrier-bracketed region of exactly --steps steps
                # (max over ranks each); the headline is the first
                "repeat_ms_per_step": {
                    "n": len(rep_ms), "min": min(rep_ms),
                    "median": float(np.median(rep_ms)), "max": max(rep_ms),
                    "all": rep_ms},
                "host_enqueue_ms_per_frame": enqueue_ms_h,
                "host_enqueue_note": "Python time to enqueue one frame "
                    "(graph build + GNN, ~110 C-ABI launches) with nothing "
                    "read back, max over ranks; the device needs "
                    "ms_per_frame_per_gpu, the ratio is the host's headroom",
                "cpu_affinity": getattr(args, "cpu_affinity", None),
                "sizes": "host-read (two waits per frame in the graph "
                         "builder)" if args.host_sized else
                         "capacity form: K, E0, E1 stay on the device, one "
                         "host read per batch of frames (results)",
                "schedule": "sequential, 1 stream" if args.no_pipeline else
                            ("%d HIP streams, frame i (graph build + GNN) "
                             "wholly on stream i %% %d; sizes are read once "
                             "per batch of frames" % (n_fs, n_fs)
                             if frame_sched else
                             "%d HIP streams: the graphs of frames i+1 .. i+%d "
                             "are built on %d builder stream(s) while %d GNN "
                             "stream(s) run frame i%s" % (
                                 args.compute_streams + n_builders,
                                 n_builders, n_builders, args.compute_streams,
                                 ("; graph stream on %d reserved CUs, GNN "
                                  "streams on the other CUs (CU-masked "
                                  "streams)" % args.graph_cus
                                  if args.graph_cus > 0 else ""))),
                "parallelism": "frame-parallel x%d (no collective)" % world,
                "distributed": dist_info(dist, world),
                "frames_per_sec_per_gpu": fps / world,
                # mean over the timed frames; x frames/s/GPU = the two rates
                "algorithmic_gflop_per_frame": st["alg_flops_mean"] / 1e9,
                "executed_gflop_per_frame": st["exe_flops_mean"] / 1e9,
                "algorithmic_tflops": st["alg_flops_mean"] * fps / world / 1e12,
                "executed_tflops": st["exe_flops_mean"] * fps / world / 1e12,
                "executed_frac_of_fp32_mfma_peak":
                    st["exe_flops_mean"] * fps / world / 1e12 / FP32_MFMA_PEAK_TF,
                "phase_ms_frame_seed%d" % first: {
                    "gen graph": engine.time_dict['gen graph'] / frames * 1e3,
                    "gnn inference":
                        engine.time_dict['gnn inference'] / frames * 1e3,
                    "decode box + nms": post["ms"]},
                "latency_ms_frame_seed%d" % first: lat,
                "graph_build_ms_frame_seed%d" % first: build,
                "capacity_overflow_rebuilds": engine.deferred_overflows,
                "postprocess": post,
            },
        }
        if second is not None:
            res["config"]["secondary"] = second
        for arith, line in sec16.items():
            res["config"]["secondary_" + arith] = line
        if ped is not None:
            res["config"]["secondary_ped"] = ped
        if trn is not None:
            res["config"]["secondary_train"] = trn
        if e2e is not None:
            res["config"]["secondary_e2e"] = e2e
        if not args.no_roofline:
            width = cfg['model_kwargs']['layer_configs'][1]['kwargs'][
                'edge_MLP_depth_list'][-1] if len(
                cfg['model_kwargs']['layer_configs']) > 2 else 300
            live = None
            if world == 1 and not args.no_live_pmc:
                torch.cuda.synchronize()
                live = live_pmc_scatter(args.preset)
            res["roofline"] = roofline_scatter_max(torch, edges[1], n_k, width,
                                                   live_pmc=live)
            res["roofline"]["workload"] = {
                "frame_seed": first, "E": n_e1, "C": width, "K": n_k}
            trace = None
            if world == 1 and not args.no_live_pmc:
                torch.cuda.synchronize()
                trace = live_kernel_trace(args.config, args.preset,
                                          args.edge_arith)
            mf = roofline_edge_kernel(torch, engine, edges[1], n_k,
                                      frame=(x, f))
            if mf is not None:
                mf["workload"] = {"frame_seed": first, "E": n_e1, "K": n_k}
                res["roofline_mfma"] = apply_kernel_trace(mf, trace,
                                                          "edge_ws_kernel")
            pl = roofline_pool_kernel(torch, engine, frame=(x, f))
            if pl is not None:
                res["roofline_pool"] = apply_kernel_trace(
                    pl, trace, ("pool_hidden_kernel",
                                "edge_ws_kernel<16, 8, false, true>")
                    if pl.get("launches") == 2 else "pool_ws_kernel")
            if trace:
                # one frame alone, per kernel family (us per frame)
                per_frame = {}
                n_gnn = sum(l['type'] == 'scatter_max_graph_auto_center_net'
                            for l in cfg['model_kwargs']['layer_configs'])
                n_tr = sum(v["calls"] for k, v in trace.items()
                           if isinstance(v, dict) and
                           k.startswith("edge_ws_kernel")) / max(1, n_gnn)
                n_tr = n_tr or 10.0
                for k, v in trace.items():
                    if isinstance(v, dict):
                        fam = k.split("<")[0]
                        per_frame[fam] = per_frame.get(fam, 0.0) + \
                            v["total_us"] / n_tr
                top = sorted(per_frame.items(), key=lambda kv: -kv[1])[:14]
                res["config"]["kernel_us_per_frame_seed%d" % first] = {
                    "source": trace["_command"] + " (%g frames)" % n_tr,
                    "total": sum(per_frame.values()),
                    "top": {k: round(v, 1) for k, v in top}}
            res["roofline_graph"] = roofline_graph(torch, cfg, coords)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, params, xyz_np, inten_np,
                                               args.cpu_budget)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
