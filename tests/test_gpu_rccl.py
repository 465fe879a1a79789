"""RCCL behind the C ABI (include/pointgnn_hip.h "collectives", csrc/comm.hip)
on the ONE GPU a test box has: a world of one rank is a valid communicator and
enters the library -- unique id, ncclCommInitRank, the all-reduce kernels on
the step's stream, destroy.  What it replaces: util/tf_util.py:3-43
(average_gradients), train.py:264-288 (unify_copies), train.py:397-405.

Bars: an all-reduce(sum) over one rank is the identity, BIT for bit -- on
random data and on the step's real gradient; a training step with its
collectives forced at world 1 equals the same step with the fabric taken out
(bit for bit where the box repeats a step bit for bit: the adjoint's scatter
kernels use float atomics) and the plain step to rounding (2e-5 Frobenius)."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N_PARAMS_CAR_T3 = 1489609       # the flat gradient, 5.96 MB


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pointgnn_amd import _lib
    _lib.load()
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def comm(dev):
    from pointgnn_amd.comm import Communicator
    c = Communicator.single()
    yield c
    c.destroy()


def test_rccl_is_bound_and_reports_itself(comm):
    from pointgnn_amd.comm import Communicator
    assert (comm.world, comm.rank) == (1, 0)
    assert Communicator.rccl_version() >= 20000
    assert "rccl" in Communicator.library()
    w, r = ctypes.c_int32(-1), ctypes.c_int32(-1)
    assert comm.lib.pgnn_comm_info(comm.handle, ctypes.byref(w),
                                   ctypes.byref(r), None) == 0
    assert (w.value, r.value) == (1, 0)
    # the process holds ONE RCCL image (the one PyTorch brought)
    with open("/proc/self/maps") as f:
        images = {l.split()[-1] for l in f if "librccl" in l}
    assert len(images) == 1, images
    comm.check_async_error()


def test_world1_allreduce_is_the_identity_bit_for_bit(dev, comm):
    import torch
    g = torch.Generator(device="cpu").manual_seed(5)
    flat = (torch.randn(N_PARAMS_CAR_T3, generator=g) *
            torch.logspace(-20, 20, N_PARAMS_CAR_T3)).to(dev)
    flat[:4] = torch.tensor([0.0, -0.0, float("inf"), 1e-42], device=dev)
    want = flat.clone()
    comm.allreduce_sum(flat)
    torch.cuda.synchronize()
    assert torch.equal(flat.view(torch.int32), want.view(torch.int32))
    counts = torch.tensor([6703.0, 1234.0], dtype=torch.float64, device=dev)
    comm.allreduce_sum(counts)
    assert counts.tolist() == [6703.0, 1234.0]
    sums = torch.tensor([1.5, 2.5, 3.0, 4.0], dtype=torch.float64, device=dev)
    comm.allreduce_step(flat, sums)
    comm.allreduce_step(flat, None)
    comm.broadcast(flat, 0)
    torch.cuda.synchronize()
    assert torch.equal(flat.view(torch.int32), want.view(torch.int32))
    assert sums.tolist() == [1.5, 2.5, 3.0, 4.0]
    # ordered on a side stream like any other entry
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x = torch.arange(1000, dtype=torch.float32, device=dev)
        x.mul_(2.0)
        comm.allreduce_sum(x)
        x.add_(1.0)
    side.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float32) * 2 + 1)
    # empty buffers are legal
    comm.allreduce_sum(torch.empty(0, dtype=torch.float32, device=dev))
    comm.check_async_error()


def test_communicator_argument_errors(dev, comm, tmp_path):
    import torch
    from pointgnn_amd import _lib
    from pointgnn_amd.comm import Communicator, ID_BYTES
    lib = _lib.load()
    with pytest.raises(ValueError):
        Communicator(b"short", 1, 0)
    h = ctypes.c_void_p()
    uid = Communicator.unique_id()
    assert len(uid) == ID_BYTES
    assert lib.pgnn_comm_init_rank(uid, 2, 2, ctypes.byref(h)) == -1
    assert b"rank" in lib.pgnn_last_error()
    assert lib.pgnn_comm_init_rank(None, 1, 0, ctypes.byref(h)) == -1
    assert lib.pgnn_comm_unique_id(None) == -1
    x = torch.zeros(8, dtype=torch.float32, device=dev)
    # not a communicator handle
    fake = (ctypes.c_char * 64)()
    assert lib.pgnn_allreduce_sum_f32(fake, _lib.ptr(x), 8, None) == -1
    assert lib.pgnn_allreduce_sum_f32(comm.handle, None, 8, None) == -1
    assert lib.pgnn_allreduce_sum_f32(comm.handle, _lib.ptr(x), -1, None) == -1
    assert lib.pgnn_broadcast_f32(comm.handle, _lib.ptr(x), 8, 1, None) == -1
    with pytest.raises(ValueError):
        comm.allreduce_sum(x.to(torch.int32))
    with pytest.raises(ValueError):
        comm.allreduce_sum(torch.zeros(8))          # a CPU tensor
    # a destroyed handle is refused, not dereferenced into RCCL
    c2 = Communicator.from_file(str(tmp_path / "id"), 1, 0)
    h2 = c2.handle
    c2.allreduce_sum(x)
    c2.destroy()
    c2.destroy()                                     # idempotent
    assert lib.pgnn_comm_destroy(None) == 0


class _NullComm(object):
    """The Communicator's interface with the fabric taken out (world 1: every
    reduction is the identity): the step takes exactly the code path of a
    multi-rank step -- counts on the device, pgnn_trainer_backward_sync -- and
    only RCCL is missing.  What the RCCL step is compared against."""
    world, rank, handle = 1, 0, None

    def allreduce_sum(self, t):
        return t

    def allreduce_step(self, grads, sums=None):
        return grads


@pytest.mark.parametrize("t_config,native", [(1, True), (3, True), (1, False)])
def test_trainer_step_with_world1_rccl_equals_step_without(dev, comm, t_config,
                                                           native):
    """The real flat gradient and the counts all-reduced in place by RCCL
    (pgnn_trainer_backward_sync for the native step, pgnn_allreduce_step
    behind the Python composition) against the same step with the fabric
    taken out (_NullComm) and against the plain single-rank step.  The
    adjoint's scatter kernels add with float atomics, so two runs of ONE step
    agree to rounding only; where this box happens to repeat a step bit for
    bit, the RCCL step must too.  The all-reduce itself is checked exactly:
    the step's real gradient reduced once more, in place, does not change."""
    import torch
    sys.path.insert(0, HERE)
    from _multirank_worker import make_frame
    from pointgnn_amd import train
    cfg = configs.car_auto_config(t_config)
    batch = train.batch_data([make_frame(cfg, i, dev) for i in range(2)])
    nv = float(batch[6].sum().item())
    keys = ("cls_loss", "loc_loss", "reg_loss")

    def run(num_valid, **kw):
        tr = train.Trainer(cfg, seed=3, device=dev, **kw)
        tr.native = native
        out = tr.train_step(batch, apply=False, num_valid=num_valid)
        g = tr.grad.clone()
        tr.train_step(batch, apply=True, num_valid=num_valid)
        return tr, out, g

    def close(a, b):
        return float((a - b).norm() / a.norm())

    plain, o0, g0 = run(None)
    assert not plain._multi() and float(g0.norm()) > 0
    for num_valid in (None, nv):      # counts reduced after / before forward
        null, on, gn = run(num_valid, comm=_NullComm(), force_collective=True)
        null2, _, gn2 = run(num_valid, comm=_NullComm(), force_collective=True)
        tr, o1, g1 = run(num_valid, comm=comm, force_collective=True)
        assert tr._multi() and null._multi()
        repeatable = torch.equal(gn, gn2) and torch.equal(null.flat, null2.flat)
        if repeatable:
            assert torch.equal(gn, g1)
            assert torch.equal(null.flat, tr.flat)
            assert all(on[k] == o1[k] for k in keys)
        assert close(gn, g1) < 2e-5 and close(g0, g1) < 2e-5
        assert close(plain.flat, tr.flat) < 1e-6
        for k in keys:
            assert abs(o0[k] - o1[k]) <= 1e-6 * max(1.0, abs(o0[k])), k
        assert o0["num_endpoint"] == o1["num_endpoint"]
        assert o0["num_valid_endpoint"] == o1["num_valid_endpoint"]
        # the collective on the step's own gradient: exact identity
        again = g1.clone()
        sums = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64,
                            device=dev)
        comm.allreduce_step(again, sums)
        torch.cuda.synchronize()
        assert torch.equal(again.view(torch.int32), g1.view(torch.int32))
    # a Communicator without force_collective at world 1: no collective at all
    tr, o2, g2 = run(None, comm=comm)
    assert not tr._multi() and close(g0, g2) < 2e-5
    comm.check_async_error()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world1_nccl_process_group_and_communicator_from_it():
    """init_process_group('nccl', world_size=1, device_id=cuda:0) in its own
    process: the torch.distributed variant of the step (world-1 short-circuits
    bypassed) and a Communicator whose id travelled through that group, both
    equal to the plain step (bit for bit when a step repeats bit for bit)."""
    env = dict(os.environ, PGNN_PORT=str(_free_port()), PGNN_TEST_T="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable,
                        os.path.join(HERE, "_rccl_world1_worker.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    rec = json.JSONDecoder().raw_decode(
        p.stdout.split("RCCLW1 ")[1].lstrip())[0]
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["rccl_version"] >= 20000 and "rccl" in rec["rccl_library"]
    assert rec["grad_norm"] > 0
    assert rec["pg_grad_rel_err"] < 2e-5 and rec["comm_grad_rel_err"] < 2e-5
    assert rec["pg_weights_rel_err"] < 1e-6
    assert rec["comm_weights_rel_err"] < 1e-6
    assert rec["real_gradient_allreduce_is_identity"]
    for k in ("cls_loss", "loc_loss", "reg_loss"):
        for other in ("loss_pg", "loss_comm"):
            assert abs(rec["loss_plain"][k] - rec[other][k]) <= \
                1e-6 * max(1.0, abs(rec["loss_plain"][k])), (k, other)
    for other in ("loss_pg", "loss_comm"):
        assert rec[other]["num_endpoint"] == rec["loss_plain"]["num_endpoint"]
