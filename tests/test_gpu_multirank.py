"""Two `Trainer`s under nccl (= RCCL): the gradient all-reduce of config 4 on
real hardware (train.py:178-181, 264-288, 397-405; util/tf_util.py:3-43).

Needs two visible GPUs: with one it SKIPS (the gloo world-2 run of the same
helpers on oracle gradients is tests/test_sharding_cpu.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(t_config, backend):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PGNN_TEST_T"] = str(t_config)
    env["PGNN_TEST_BACKEND"] = backend
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(HERE, "_multirank_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    # both ranks write to one pipe: their records can land on one line, so
    # decode each record where its tag starts instead of splitting by lines
    dec = json.JSONDecoder()
    rows = [dec.raw_decode(chunk.lstrip())[0]
            for chunk in p.stdout.split("MULTIRANK ")[1:]]
    assert sorted(r["rank"] for r in rows) == [0, 1]
    for r in rows:
        assert r["backend"] == backend and r["world"] == 2
        assert r["same_on_all"] and r["weights_same"]
        # float32 sums in a different order (two partial gradients added by
        # the collective vs one pass over the merged batch; the adjoint
        # scatter kernels use float atomics): rounding-level agreement
        assert r["rel_fro_err"] < 2e-5, r
        for k in ("cls_loss", "loc_loss", "reg_loss"):
            assert abs(r["loss"][k] - r["loss_ref"][k]) <= \
                1e-5 * max(1.0, abs(r["loss_ref"][k])), (k, r)
        assert r["loss"]["num_endpoint"] == r["loss_ref"]["num_endpoint"]
        assert r["loss"]["num_valid_endpoint"] == \
            r["loss_ref"]["num_valid_endpoint"]


@pytest.mark.parametrize("t_config", [1, 3])
def test_two_rank_nccl_gradient_equals_merged_batch(t_config):
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (have %d)" % (
            torch.cuda.device_count() if torch.cuda.is_available() else 0))
    _run(t_config, "nccl")


def test_two_ranks_on_one_gpu_gloo_dry_run():
    """The same worker with both ranks on cuda:0 and gloo collectives: what a
    one-GPU box can execute of the two-rank step (everything but RCCL)."""
    _run(1, "gloo")


def test_bench_two_ranks_real_engine_on_one_gpu():
    """`bench.py --gpus 2` with the REAL engine (PGNN_BENCH_ONE_GPU=1: both
    ranks on cuda:0, gloo for the rank plumbing): frame sharding, the barrier
    + max-over-ranks timing, the repeat spread, the host-enqueue figure and
    the per-rank CPU binding all execute on hardware; only RCCL is absent
    (inference needs no collective on the data path).  The line marks itself
    as a test-mode line."""
    env = dict(os.environ, PGNN_BENCH_ONE_GPU="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
           "--steps", "3", "--warmup", "1", "--frames-per-step", "4",
           "--repeats", "2", "--preset", "car", "--frames", "2",
           "--no-roofline", "--no-cpu-baseline", "--no-capture"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = lines[0]
    c = r["config"]
    assert r["n_gpus"] == 2 and "TEST MODE" in r["data"]
    assert c["distributed"] == {"world_size": 2, "backend": "gloo"}
    assert c["frames_per_gpu_per_step"] == 4
    assert c["frames_timed"] == 2 * 3 * 4          # both ranks' frames
    assert c["repeat_ms_per_step"]["n"] == 2
    assert c["host_enqueue_ms_per_frame"] > 0
    assert c["cpu_affinity"]["bound"]
    assert r["value"] > 0 and r["dtype"] == "f32"
    assert abs(r["value"] - 2 * 3 * 4 / (r["ms_per_step"] * 3e-3)) \
        < 1e-6 * r["value"]


def test_bench_train_two_ranks_on_one_gpu():
    """`bench.py --train --gpus 2` in the same test mode: the loader thread,
    the counts' and the gradient's collectives (through the gloo group here:
    RCCL refuses two ranks per device), the max-over-ranks timing and the
    post-region all-reduce timing of the training line."""
    env = dict(os.environ, PGNN_BENCH_ONE_GPU="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--train", "--gpus",
           "2", "--steps", "4", "--warmup", "3", "--frames", "4",
           "--no-live-pmc"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = lines[0]
    c = r["config"]
    assert r["n_gpus"] == 2 and r["steps"] == 4
    assert c["distributed"] == {"world_size": 2, "backend": "gloo"}
    assert "all_reduce (gloo)" in c["collective"]
    assert c["allreduce_ms"] > 0 and c["allreduce_bytes"] == 4 * c["params"]
    assert abs(r["value"] - 2 * 2 * 4 / (r["ms_per_step"] * 4e-3)) \
        < 1e-6 * r["value"]
    assert all(np.isfinite(v) for v in c["last_loss"].values())
