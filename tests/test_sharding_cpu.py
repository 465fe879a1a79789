"""Multi-process path on CPU (gloo, world_size 2): frames are sharded by rank
with no data-path collective; the only communication bench.py performs is the
barrier + MAX-reduce of the elapsed time, exercised here with gloo."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pointgnn_amd  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_frames, q):
    sys.path.insert(0, ROOT)
    import pointgnn_amd  # noqa: F401
    from pointgnn_amd.engine import shard_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_frames(num_frames, rank, world)
    # every rank "processes" its frames: here, a checksum of the frame ids
    local = torch.tensor([float(sum(mine)), float(len(mine))], dtype=torch.float64)
    dist.barrier()
    elapsed = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)     # bench.py's timing rule
    gathered = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, local)
    q.put((rank, mine, float(elapsed.item()), [g.tolist() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_frames_partition():
    from pointgnn_amd.engine import shard_frames
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 16, 100):
            parts = [shard_frames(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))                 # disjoint cover
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard_frames(4, 2, 2)


def test_two_process_gloo_sharding():
    world, n = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    frames = sorted(i for _, mine, _, _ in results for i in mine)
    assert frames == list(range(n))
    for rank, mine, elapsed, gathered in results:
        assert elapsed == 1.5                              # MAX over ranks
        assert sum(g[1] for g in gathered) == n
        assert sum(g[0] for g in gathered) == sum(range(n))


def test_eight_process_gloo_sharding():
    """The driver's widest form: 8 ranks, frames sharded with no data-path
    collective; every frame is processed exactly once and the timing rule sees
    the slowest rank."""
    world, n = 8, 61
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results.sort()
    frames = sorted(i for _, mine, _, _ in results for i in mine)
    assert frames == list(range(n))
    sizes = [len(mine) for _, mine, _, _ in results]
    assert max(sizes) - min(sizes) <= 1
    for rank, mine, elapsed, gathered in results:
        assert elapsed == 7.5                              # MAX over 8 ranks
        assert sum(g[1] for g in gathered) == n
        assert sum(g[0] for g in gathered) == sum(range(n))


class _StoreComm(object):
    """Stand-in for comm.Communicator on a box without GPUs: the same three
    methods the product's helpers call (`world`, allreduce_sum,
    allreduce_step), the reductions carried by the process group instead of
    RCCL -- so the `comm=` branch of the helpers runs at 2 and 8 ranks here,
    and only the fabric differs on the GPUs."""

    def __init__(self):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.calls = []

    def allreduce_sum(self, t):
        self.calls.append(("sum", t.dtype, t.numel()))
        dist.all_reduce(t)
        return t

    def allreduce_step(self, grads, sums=None):
        self.calls.append(("step", grads.numel(),
                           0 if sums is None else sums.numel()))
        dist.all_reduce(grads)
        if sums is not None:
            dist.all_reduce(sums)
        return grads


def _train_worker(rank, world, port, q, use_comm=False):
    """Each 'rank' holds one frame.  Gradients are computed by the CPU oracle
    with the GLOBAL normalisers obtained through the product's collective
    helpers (gloo here, RCCL on GPUs); their all-reduce SUM must equal the
    oracle's global gradient (train.py:264-297 + util/tf_util.py:3-43)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import pointgnn_amd  # noqa: F401
    from pointgnn_amd import configs, weights
    from pointgnn_amd.train import (allreduce_endpoint_counts,
                                    allreduce_gradients)
    from oracle import train_oracle as to
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = configs.car_auto_config(0)
    params = weights.init_params(cfg, seed=1, bias_scale=0.1)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "graph_tiny.npz")))
    k = g["kp_xyz"].shape[0]
    rng = np.random.default_rng(10 + rank)
    labels = rng.integers(0, 4, (k, 1)).astype(np.int32)
    boxes = rng.standard_normal((k, 1, 7)).astype(np.float32)
    valid = (rng.random((k, 1, 1)) < (0.3 + 0.4 * rank / max(1, world - 1))
             ).astype(np.float32)
    batch = (g["intensity"] * (1 + rank),
             [g["xyz"], g["kp_xyz"], g["kp_xyz"]],
             [g["kp_idx"], np.arange(k, dtype=np.int32).reshape(-1, 1)],
             [g["ref_edges0"], g["ref_edges1"]], labels, boxes, valid)
    comm = _StoreComm() if use_comm else None
    n_tot, nv_tot = allreduce_endpoint_counts(k, float(valid.sum()),
                                              torch.device("cpu"), comm=comm)
    tp = {kk: torch.tensor(v, dtype=torch.float64, requires_grad=True)
          for kk, v in params.items()}
    logits, pred = to.forward(tp, cfg, *batch[:4])
    ce, loc, n, nv = to.loss_terms(cfg, logits, pred, labels, boxes, valid)
    local = 0.1 * ce / n_tot + 10.0 * loc / nv_tot
    names = list(tp)
    grads = torch.autograd.grad(local, [tp[n_] for n_ in names],
                                allow_unused=True)
    flat = torch.cat([(torch.zeros_like(tp[n_]) if g_ is None else g_).reshape(-1)
                      for n_, g_ in zip(names, grads)])
    sums = torch.tensor([float(ce), float(loc), float(n), float(nv)],
                        dtype=torch.float64)
    allreduce_gradients(flat, sums if use_comm else None, comm=comm)
    if use_comm:
        # the Communicator branch: one f64[2] reduction for the counts, one
        # grouped (gradient, sums) reduction -- and nothing else
        assert comm.calls == [("sum", torch.float64, 2),
                              ("step", flat.numel(), 4)], comm.calls
        assert float(sums[2]) == n_tot and float(sums[3]) == nv_tot
    q.put((rank, batch, n_tot, nv_tot, flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,use_comm", [(2, False), (8, False), (8, True)])
def test_gradient_allreduce_equals_global_gradient(world, use_comm):
    """world 2, and 8 = the node's width (config 4: batch 16 over 8 ranks);
    use_comm: through the helpers' Communicator branch (pgnn_allreduce_* on
    the GPUs; here a stand-in whose reductions ride on gloo)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from pointgnn_amd import configs, weights
    from oracle import train_oracle as to
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker,
                         args=(r, world, port, q, use_comm))
             for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in range(world)],
                     key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = configs.car_auto_config(0)
    params = weights.init_params(cfg, seed=1, bias_scale=0.1)
    _, g_ref, _ = to.step_gradients(params, cfg, [r[1] for r in results])
    ref = np.concatenate([g_ref[n].reshape(-1) for n in params])
    assert all(r[2] == world * results[0][1][4].shape[0] for r in results)
    for _, _, _, _, flat in results:          # every rank holds the same sum
        np.testing.assert_allclose(flat, ref, atol=1e-12, rtol=1e-9)


def _metrics_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pointgnn_amd  # noqa: F401
    from pointgnn_amd.metrics import allreduce_state
    from oracle import metrics_oracle as mo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, l = mo.synthetic_step(40 + rank, 900 + 50 * rank, 4)
    state = torch.from_numpy(mo.state_counts(p, l, 4))
    allreduce_state(state)
    q.put((rank, state.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_metric_counters_add_up():
    """The class counters are integers: the all-reduce SUM of the per-rank
    states is the state of the concatenated batch."""
    import numpy as np
    from oracle import metrics_oracle as mo
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metrics_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    parts = [mo.synthetic_step(40 + r, 900 + 50 * r, 4) for r in range(world)]
    whole = mo.state_counts(np.concatenate([a for a, _ in parts]),
                            np.concatenate([b for _, b in parts]), 4)
    for _, st in results:
        assert np.array_equal(st, whole)


def _run_bench(argv, env_extra, timeout=240):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv,
                       env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, [json.loads(l) for l in lines]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it creates its two
    ranks itself (the driver's plain command line); rank 0 prints exactly one
    JSON line whose n_gpus is what the process group saw.  PGNN_BENCH_STUB
    swaps the engine for a sleep and RCCL for gloo -- the launcher, the rank
    plumbing, the barrier + MAX timing rule and the JSON contract are the
    real code."""
    p, lines = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2"],
                          {"PGNN_BENCH_STUB": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    r = lines[0]
    assert r["n_gpus"] == 2 and r["steps"] == 6 and r["warmup"] == 2
    assert r["config"]["distributed"] == {"world_size": 2, "backend": "gloo"}
    # 8 frames per GPU per step at every N (weak scaling), all ranks' frames
    assert r["config"]["frames_per_gpu_per_step"] == 8
    assert r["config"]["frames_timed"] == 2 * 6 * 8
    assert r["data"] == "stub" and r["scaling"] == "weak"
    # whole-job value: 2 ranks x 6 steps x 8 frames over the slowest rank's
    # time of the FIRST barrier-bracketed region; the other regions are spread
    assert abs(r["value"] - 2 * 6 * 8 / (r["ms_per_step"] * 6e-3)) \
        < 1e-6 * r["value"]
    rep = r["config"]["repeat_ms_per_step"]
    assert rep["n"] == 5 and len(rep["all"]) == 5
    assert rep["all"][0] == pytest.approx(r["ms_per_step"])
    assert rep["min"] <= rep["median"] <= rep["max"]
    assert r["config"]["timed_region_s"] == pytest.approx(
        r["ms_per_step"] * 6e-3)


def test_bench_rank_binding_splits_cpus_without_numa_info():
    """bench.bind_rank_to_numa: with no NUMA node reported for the GPU (this
    container has none) the allowed CPUs are split evenly by local rank --
    disjoint, covering sets -- and the process really is bound."""
    import bench

    class _NoGpu(object):
        class cuda(object):
            @staticmethod
            def get_device_properties(i):
                raise RuntimeError("no GPU here")
    before = os.sched_getaffinity(0)
    try:
        seen = []
        for r in range(2):
            os.sched_setaffinity(0, before)
            info = bench.bind_rank_to_numa(_NoGpu, r, 2)
            if len(before) < 2:
                pytest.skip("one CPU")
            assert info["bound"] and info["numa_node"] == -1
            got = os.sched_getaffinity(0)
            assert len(got) == info["n_cpus"] and got <= before
            seen.append(got)
        assert not (seen[0] & seen[1])
        assert (seen[0] | seen[1]) == before
        assert bench._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    finally:
        os.sched_setaffinity(0, before)


def test_bench_eight_ranks_json_contract():
    """`python bench.py --gpus 8` (stub engine, gloo): the JSON line of the
    driver's 8-GPU run -- n_gpus, world size, frames timed over all ranks,
    whole-job value over the slowest rank's time."""
    p, lines = _run_bench(["--gpus", "8", "--steps", "4", "--warmup", "1"],
                          {"PGNN_BENCH_STUB": "1", "OMP_NUM_THREADS": "1"},
                          timeout=480)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    r = lines[0]
    assert r["n_gpus"] == 8 and r["steps"] == 4 and r["warmup"] == 1
    assert r["config"]["distributed"] == {"world_size": 8, "backend": "gloo"}
    assert r["config"]["frames_per_gpu_per_step"] == 8
    assert r["config"]["frames_timed"] == 8 * 4 * 8
    assert r["scaling"] == "weak" and r["higher_is_better"] is True
    assert abs(r["value"] - 8 * 4 * 8 / (r["ms_per_step"] * 4e-3)) \
        < 1e-6 * r["value"]


def test_bench_rank_binding_on_a_faked_eight_gpu_node(tmp_path, monkeypatch):
    """bench.bind_rank_to_numa on a faked /sys tree: 8 GPUs, four per NUMA
    node, each node listing half of the allowed CPUs -> 8 disjoint CPU sets,
    every rank inside its GPU's node, the node's CPUs split over its four
    ranks."""
    import bench
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 8:
        pytest.skip("fewer than 8 CPUs")
    half = len(allowed) // 2
    nodes = {0: allowed[:half], 1: allowed[half:]}
    sys_root = tmp_path / "sys"
    for node, cpus in nodes.items():
        d = sys_root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    bus_of = {r: 0x10 + 0x11 * r for r in range(8)}
    for r in range(8):
        d = sys_root / "bus" / "pci" / "devices" / ("0000:%02x:00.0" % bus_of[r])
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % (r // 4))
    monkeypatch.setattr(bench, "SYS_ROOT", str(sys_root))

    class _Prop(object):
        def __init__(self, r):
            self.pci_domain_id, self.pci_bus_id, self.pci_device_id = \
                0, bus_of[r], 0

    class _Eight(object):
        class cuda(object):
            @staticmethod
            def get_device_properties(i):
                return _Prop(i)
    before = os.sched_getaffinity(0)
    try:
        seen = []
        for r in range(8):
            os.sched_setaffinity(0, before)
            info = bench.bind_rank_to_numa(_Eight, r, 8)
            assert info["bound"] and info["numa_node"] == r // 4, info
            got = os.sched_getaffinity(0)
            assert got and got <= set(nodes[r // 4])
            assert len(got) == info["n_cpus"]
            seen.append(got)
        for i in range(8):
            for j in range(i + 1, 8):
                assert not (seen[i] & seen[j]), (i, j)
        assert set().union(*seen[:4]) == set(nodes[0])
        assert set().union(*seen[4:]) == set(nodes[1])
    finally:
        os.sched_setaffinity(0, before)


def test_bench_under_an_external_launcher_and_mismatch():
    """The driver's multi-GPU form: ranks exist already (torch.distributed.run);
    bench.py must not launch again, and a WORLD_SIZE that differs from --gpus is
    refused instead of reported."""
    import json
    import subprocess
    env = dict(os.environ, PGNN_BENCH_STUB="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "4", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
    p, lines = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "0"],
                          {"PGNN_BENCH_STUB": "1", "WORLD_SIZE": "2",
                           "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines
    assert "WORLD_SIZE=2" in p.stderr


def test_bench_refuses_gpus_it_does_not_have():
    import torch as _t
    if _t.cuda.is_available() and _t.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    p, lines = _run_bench(["--gpus", "2", "--steps", "2"], {})
    assert p.returncode != 0 and not lines
    assert "refusing" in p.stderr


def test_bench_flop_accounting_matches_survey():
    """97 536 FLOP/edge(E0), 361 800 FLOP/edge(E1) per iteration,
    360 000 / 38 784 / 228 864 per vertex; pool statistics are means over the
    timed frames."""
    sys.path.insert(0, ROOT)
    import bench
    from pointgnn_amd import configs
    cfg = configs.car_auto_config(3)
    k, e0, e1 = 3213, 323939, 415621
    want = 97536 * e0 + 360000 * k + 3 * (38784 * k + 361800 * e1 + 360000 * k) \
        + 228864 * k
    assert bench.algorithmic_flops_per_frame(cfg, k, e0, e1) == want
    exe = bench.executed_flops_per_frame(cfg, k, e0, e1)
    assert exe == want - 3 * (2 * 303 * 300 * e1) + 3 * (2 * 303 * 300 + 1800) * k
    st = bench.pool_statistics(cfg, [(k, e0, e1), (k, e0, e1 + 2000)])
    assert st["frames"] == 2 and st["E1"] == {"mean": e1 + 1000.0, "min": e1,
                                              "max": e1 + 2000}
    assert abs(st["alg_flops_mean"] - (want + 3 * 361800 * 1000)) < 1
