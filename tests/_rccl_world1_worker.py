"""Worker of tests/test_gpu_rccl.py: ONE rank on cuda:0 with a real `nccl`
(= RCCL) process group and a pgnn Communicator built from it -- what a 1-GPU
box can execute of the multi-rank training step INCLUDING the fabric library.
Prints one `RCCLW1 {json}` record."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import comm as pcomm, configs, train  # noqa: E402
from _multirank_worker import make_frame  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(
        "nccl", init_method="tcp://127.0.0.1:%s" % os.environ["PGNN_PORT"],
        world_size=1, rank=0, device_id=dev)
    assert dist.get_backend() == "nccl"
    t_config = int(os.environ.get("PGNN_TEST_T", "1"))
    cfg = configs.car_auto_config(t_config)
    frames = [make_frame(cfg, i, dev) for i in range(2)]
    batch = train.batch_data(frames)

    def run(**kw):
        tr = train.Trainer(cfg, seed=3, device=dev, **kw)
        out = tr.train_step(batch, apply=False)
        g = tr.grad.clone()
        out2 = tr.train_step(batch, apply=True)
        return tr, out, g, out2

    def rel(a, b):
        return float((a - b).norm() / a.norm())

    plain, o0, g0, _ = run()
    plain2, _, g0b, _ = run()
    assert not plain._multi()
    # torch.distributed's nccl all_reduce, the world-1 short-circuit bypassed
    tpg, o1, g1, _ = run(process_group=dist.group.WORLD, force_collective=True)
    assert tpg._multi()
    # the C ABI's communicator, its id handed out through the process group
    c = pcomm.Communicator.from_torch(dist.group.WORLD)
    tc, o2, g2, _ = run(comm=c, force_collective=True)
    assert tc._multi()
    again = g2.clone()
    c.allreduce_step(again, None)
    dist.all_reduce(again)
    torch.cuda.synchronize()
    keys = ("cls_loss", "loc_loss", "reg_loss", "num_endpoint",
            "num_valid_endpoint")
    res = {
        "backend": dist.get_backend(), "world": dist.get_world_size(),
        "rccl_version": pcomm.Communicator.rccl_version(),
        "rccl_library": pcomm.Communicator.library(),
        "grad_norm": float(g0.norm()),
        "plain_step_repeats_bit_for_bit": bool(
            torch.equal(g0, g0b) and torch.equal(plain.flat, plain2.flat)),
        "pg_grad_rel_err": rel(g0, g1), "comm_grad_rel_err": rel(g0, g2),
        "pg_weights_rel_err": rel(plain.flat, tpg.flat),
        "comm_weights_rel_err": rel(plain.flat, tc.flat),
        "real_gradient_allreduce_is_identity": bool(
            torch.equal(again.view(torch.int32), g2.view(torch.int32))),
        "loss_plain": {k: o0[k] for k in keys},
        "loss_pg": {k: o1[k] for k in keys},
        "loss_comm": {k: o2[k] for k in keys},
    }
    c.check_async_error()
    c.destroy()
    sys.stdout.write("RCCLW1 " + json.dumps(res) + "\n")
    sys.stdout.flush()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
