#!/usr/bin/env python
"""Phase timeline of the weights-stationary edge kernel (csrc/edge_ws.h):
s_memtime stamps written by the kernel itself (pgnn_set_debug_buffer) at the
four phase boundaries of each wave's first 38 tiles, plus every wave's begin /
end on the shader clock and on the constant 100 MHz clock.

    python tools/ws_timeline.py [--preset car_600k] [--tune=key=value ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
import bench  # noqa: E402

preset = "car_600k"
for i, a in enumerate(sys.argv):
    if a == "--preset":
        preset = sys.argv[i + 1]
dev = torch.device("cuda", 0)
cfg = configs.get_config("car_auto_T3")
eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                      device=dev)
xyz, inten = synthetic_cloud(seed=0, preset=preset)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
eng.run_frame(x, f)
coords, kps, edges = eng.last_graph
n_k = int(coords[1].shape[0])
if "--local-src" in sys.argv:
    # experiment: what if keypoints were numbered spatially (src close to dst)?
    e1 = edges[1].clone()
    e1[:, 0] = torch.clamp(e1[:, 1] + (e1[:, 0] % 128) - 64, 0, n_k - 1)
    edges = [edges[0], e1.contiguous()]
frame = (x, f)
if "--train-graph" in sys.argv:
    # the training step's level-1 graph: two frames built with the TRAINING
    # graph kwargs (voxel 0.8 m, random keypoints, fan-in capped at 256) and
    # merged (train.batch_data) -- ~70k edges on ~1.6k vertices: a few tiles
    # per wave, where the kernel's fixed costs show
    from pointgnn_amd import graph_gen, train
    fn = graph_gen.get_graph_generate_fn(cfg['graph_gen_method'])
    np.random.seed(99)
    frs = []
    for sd in (0, 1):
        xx, ff = synthetic_cloud(seed=sd, preset="car")
        xx, ff = torch.from_numpy(xx).to(dev), torch.from_numpy(ff).to(dev)
        cs, ks, es = fn(xx, **cfg['graph_gen_kwargs'])
        kk = int(cs[1].shape[0])
        z = torch.zeros((kk, 1), device=dev)
        frs.append((ff, cs, ks, es, z, z, z))
    merged = train.batch_data(frs)
    coords, edges = merged[1], merged[3]
    n_k = int(coords[1].shape[0])
    frame = None
lib = _lib.load()
for a in sys.argv[1:]:
    if a.startswith("--tune="):
        k, v = a[len("--tune="):].split("=")
        _lib.set_tunable(k, int(v))
STAMP, STRIDE, WAVES, GRID = 38, 8 + 4 * 38, 8, 256
buf = torch.zeros(GRID * WAVES * STRIDE, dtype=torch.int64, device=dev)
r0 = bench.roofline_edge_kernel(torch, eng, edges[1], n_k, reps=10, frame=frame)
lib.pgnn_set_debug_buffer(_lib.ptr(buf))
r = bench.roofline_edge_kernel(torch, eng, edges[1], n_k, reps=1, frame=frame)
torch.cuda.synchronize()
lib.pgnn_set_debug_buffer(None)
ts = buf.cpu().numpy().reshape(GRID * WAVES, STRIDE)
hdr, tiles = ts[:, :8], ts[:, 8:].reshape(-1, STAMP, 4)
live = hdr[:, 0] > 0
print("E1 %d K %d; kernel %.1f us (%.1f us with stamps); waves stamped %d" % (
    edges[1].shape[0], n_k, r0["avg_launch_us"], r["avg_launch_us"], live.sum()))
if not live.any():
    sys.exit("no stamps: the weights-stationary kernel did not run")
hdr, tiles = hdr[live].copy(), tiles[live]
rt_entry = hdr[:, 6].copy()          # wave entered the kernel (100 MHz clock)
hdr[:, 6] = hdr[:, 5] // 100         # slice
hdr[:, 5] = hdr[:, 5] % 100          # column tiles of the group
cyc = hdr[:, 1] - hdr[:, 0]
rt0, rt1 = hdr[:, 2], hdr[:, 3]
ghz = cyc / np.maximum(rt1 - rt0, 1) * 0.1
print("shader clock during the kernel: %.3f GHz (p10 %.3f p90 %.3f)" % (
    ghz.mean(), np.percentile(ghz, 10), np.percentile(ghz, 90)))
t_begin = (rt0 - rt0.min()) / 100.0   # us
t_end = (rt1 - rt0.min()) / 100.0
print("wave begin (after weights->LDS) us: min %.1f p50 %.1f max %.1f" % (
    t_begin.min(), np.median(t_begin), t_begin.max()))
t_entry = (rt_entry - rt_entry.min()) / 100.0
print("wave entry us (first wave of the launch = 0): p50 %.1f max %.1f; "
      "weights->LDS (entry to begin) us: min %.1f p50 %.1f max %.1f; last "
      "wave end - first entry = %.1f us (rocprof kernel duration minus this = "
      "dispatch + drain)" % (
          np.median(t_entry), t_entry.max(),
          ((rt0 - rt_entry) / 100.0).min(), np.median((rt0 - rt_entry) / 100.0),
          ((rt0 - rt_entry) / 100.0).max(),
          (rt1.max() - rt_entry.min()) / 100.0))
print("wave end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (
    t_end.min(), np.percentile(t_end, 10), np.median(t_end),
    np.percentile(t_end, 90), t_end.max()))
for ntg in sorted(set(hdr[:, 5])):
    m = hdr[:, 5] == ntg
    nt = hdr[m, 4]
    print("-- column group of %d tiles: %d waves, %d..%d row tiles per wave, "
          "wave duration us p50 %.1f max %.1f" % (
              ntg, m.sum(), nt.min(), nt.max(),
              np.median(t_end[m] - t_begin[m]), (t_end[m] - t_begin[m]).max()))
    tl = tiles[m]
    ok = tl[:, :, 3] > 0

    def stat(name, a, ok=ok):
        a = a[ok]
        print("   %-30s mean %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (
            name, a.mean(), np.median(a), np.percentile(a, 90), a.max()))
    stat("gather (cycles)", tl[:, :, 1] - tl[:, :, 0])
    stat("MFMA phase", tl[:, :, 2] - tl[:, :, 1])
    stat("scatter-max", tl[:, :, 3] - tl[:, :, 2])
    per = tl[:, 1:, 0] - tl[:, :-1, 0]
    stat("tile period", per, ok[:, 1:] & ok[:, :-1])
    mf = 4 * 19 * ntg * 32
    print("   MFMA issue floor per tile: %d cycles alone, %d sharing the SIMD"
          % (mf, 2 * mf))

if "--balance" in sys.argv:
    # where does the spread of wave end times come from?
    blk = np.repeat(np.arange(GRID), WAVES)[live]
    wv = np.tile(np.arange(WAVES), GRID)[live]
    dur = t_end - t_begin
    wg_mean = np.array([dur[blk == b].mean() for b in range(GRID)])
    wg_std_in = np.array([dur[blk == b].std() for b in range(GRID)])
    print("wave duration us: overall mean %.1f std %.1f; std of workgroup "
          "means %.1f; mean std inside a workgroup %.1f" % (
              dur.mean(), dur.std(), wg_mean.std(), wg_std_in.mean()))
    sl = hdr[:, 6]
    print("by XCD slice (mean us): " + " ".join(
        "%d:%.0f" % (x, dur[sl == x].mean()) for x in sorted(set(sl))))
    print("by wave slot in the workgroup: " + " ".join(
        "%d:%.0f" % (w, dur[wv == w].mean()) for w in range(WAVES)))
    for ntg in sorted(set(hdr[:, 5])):
        m = hdr[:, 5] == ntg
        tl = tiles[m]
        ok = tl[:, :, 3] > 0
        per_wave = np.array([
            (t[o][:, 3] - t[o][:, 0]).mean() if o.any() else 0
            for t, o in zip(tl, ok)])
        nb = np.array([((t[o][:, 3] - t[o][:, 2]) > 5000).sum()
                       for t, o in zip(tl, ok)])
        print("group %d: corr(wave duration, slow scatter tiles among the "
              "stamped) = %.2f; per-wave mean tile cycles p10 %.0f p50 %.0f "
              "p90 %.0f" % (ntg, np.corrcoef(dur[m], nb)[0, 1],
                            np.percentile(per_wave, 10), np.median(per_wave),
                            np.percentile(per_wave, 90)))
if "--dump" in sys.argv:
    for w in (0, 333, 1500):
        tl = tiles[w]
        print("wave %d (group of %d tiles, %d row tiles): gather / MFMA / "
              "scatter-max cycles per tile" % (w, hdr[w, 5], hdr[w, 4]))
        print("  " + " ".join("%d/%d/%d" % (t[1] - t[0], t[2] - t[1], t[3] - t[2])
                              for t in tl if t[3] > 0))
if "--cost-model" in sys.argv:
    # Is the spread of SIMD end times explained by the segment boundaries in
    # each wave's static range?  dst is sorted, every vertex has its self
    # edge: the runs that END inside rows [e0, e1) number dst[e1] - dst[e0].
    E = int(edges[1].shape[0])
    dst = edges[1][:, 1].cpu().numpy().astype(np.int64)
    n_wt = (E + 15) // 16
    wg0 = {7: (0, 12), 6: None}
    blk = np.repeat(np.arange(GRID), WAVES)[live]
    wv = np.tile(np.arange(WAVES), GRID)[live]
    sl = blk % 8
    local = blk // 8
    # groups 7/6/6 own local workgroups [0,12), [12,22), [22,32)
    g_lo = np.where(local < 12, 0, np.where(local < 22, 12, 22))
    g_n = np.where(local < 12, 12, 10)
    s_first = n_wt * sl // 8
    s_last = n_wt * (sl + 1) // 8
    nw = g_n * WAVES
    wi = (local - g_lo) * WAVES + wv
    span = s_last - s_first
    t_first = s_first + span * wi // nw
    t_last = s_first + span * (wi + 1) // nw
    assert np.array_equal(t_last - t_first, hdr[:, 4]), "partition formula"
    e0 = np.minimum(t_first * 16, E - 1)
    e1 = np.minimum(t_last * 16, E - 1)
    closes = dst[e1] - dst[e0]
    tiles_n = (t_last - t_first).astype(np.float64)
    dur = t_end - t_begin
    ntg = hdr[:, 5].astype(np.float64)
    # SIMD = waves (w, w + 4) of a workgroup
    key = blk * 4 + (wv % 4)
    simd_end = np.zeros(GRID * 4)
    simd_beg = np.full(GRID * 4, 1e30)
    simd_tiles = np.zeros(GRID * 4)
    simd_close = np.zeros(GRID * 4)
    simd_ntg = np.zeros(GRID * 4)
    np.maximum.at(simd_end, key, t_end)
    np.minimum.at(simd_beg, key, t_begin)
    np.add.at(simd_tiles, key, tiles_n)
    np.add.at(simd_close, key, closes)
    simd_ntg[key] = ntg
    sd = simd_end - simd_beg
    print("SIMD busy span us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f "
          "(kernel %.1f)" % (sd.mean(), np.percentile(sd, 10), np.median(sd),
                             np.percentile(sd, 90), sd.max(),
                             r0["avg_launch_us"]))
    print("SIMD end us: p50 %.1f p90 %.1f max %.1f; all-equal bound = mean "
          "span %.1f + begin %.1f" % (np.median(simd_end),
                                      np.percentile(simd_end, 90),
                                      simd_end.max(), sd.mean(),
                                      simd_beg.mean()))
    for g in sorted(set(simd_ntg)):
        m = simd_ntg == g
        A = np.stack([simd_tiles[m], simd_close[m], np.ones(m.sum())], 1)
        coef, res, _, _ = np.linalg.lstsq(A, sd[m], rcond=None)
        pred = A @ coef
        print("group of %d column tiles: span ~ %.4f us/tile + %.4f us/closing"
              " + %.1f; closing = %.2f tiles; residual std %.1f us (span std "
              "%.1f); closings per SIMD p10 %.0f p50 %.0f p90 %.0f" % (
                  g, coef[0], coef[1], coef[2], coef[1] / coef[0],
                  (sd[m] - pred).std(), sd[m].std(),
                  np.percentile(simd_close[m], 10), np.median(simd_close[m]),
                  np.percentile(simd_close[m], 90)))
    print("by XCD slice: mean span " + " ".join(
        "%d:%.0f" % (x, sd[(np.arange(GRID * 4) // 4) % 8 == x].mean())
        for x in range(8)) + "; closings " + " ".join(
        "%d:%.0f" % (x, simd_close[(np.arange(GRID * 4) // 4) % 8 == x].mean())
        for x in range(8)))
