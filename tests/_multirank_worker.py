"""Worker of tests/test_gpu_multirank.py: launched by torch.distributed.run,
one process per GPU, backend nccl (= RCCL on ROCm).

Every rank holds the same seeded weights and steps on ITS frame(s) of a
global batch; the all-reduced flat gradient must equal the gradient one
process computes on the merged batch (util/tf_util.py:3-43 average_gradients
over towers built with train.py:264-288's unify_copies weights), and the
loss values must be the global ones on every rank."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, graph_gen, train  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402


def make_frame(cfg, frame_id, dev):
    """A training-mode frame that any rank can rebuild bit for bit: the
    graph's random draws come from numpy's global RNG, seeded per frame."""
    xyz, inten = synthetic_cloud(seed=frame_id, preset="small")
    np.random.seed(1000 + frame_id)
    fn = graph_gen.get_graph_generate_fn(cfg['graph_gen_method'])
    coords, kps, edges = fn(torch.from_numpy(xyz).to(dev),
                            **cfg['graph_gen_kwargs'])
    k = int(coords[1].shape[0])
    rng = np.random.default_rng(77 + frame_id)
    lab = rng.integers(1, 3, (k, 1)).astype(np.int32)
    lab[rng.random((k, 1)) < 0.7] = 0
    boxes = rng.standard_normal((k, 1, 7)).astype(np.float32)
    valid = (lab > 0).astype(np.float32).reshape(k, 1, 1)
    return (torch.from_numpy(inten).to(dev), coords, kps, edges,
            torch.from_numpy(lab).to(dev), torch.from_numpy(boxes).to(dev),
            torch.from_numpy(valid).to(dev))


def main():
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("PGNN_TEST_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
    else:
        # one-GPU dry run of the same code: both ranks on cuda:0, the
        # collectives through gloo (RCCL refuses two ranks on one device)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group(backend)
    assert dist.get_backend() == backend
    solo = [dist.new_group([r]) for r in range(world)][rank]  # world-1 groups
    t_config = int(os.environ.get("PGNN_TEST_T", "1"))
    cfg = configs.car_auto_config(t_config)
    fpg = 2
    frames = {i: make_frame(cfg, i, dev) for i in range(world * fpg)}
    mine = [frames[rank * fpg + j] for j in range(fpg)]

    tr = train.Trainer(cfg, seed=3, device=dev, process_group=dist.group.WORLD)
    out = tr.train_step(train.batch_data(mine), apply=False)
    g_dist = tr.grad.clone()

    ref = train.Trainer(cfg, seed=3, device=dev, process_group=solo)
    assert torch.equal(ref.flat, tr.flat)
    out_ref = ref.train_step(
        train.batch_data([frames[i] for i in range(world * fpg)]), apply=False)
    g_ref = ref.grad.clone()

    # every rank computed the same global gradient (bitwise: one all-reduce)
    gathered = [torch.empty_like(g_dist) for _ in range(world)]
    dist.all_gather(gathered, g_dist)
    same_on_all = all(torch.equal(gathered[0], g) for g in gathered)

    err = float((g_dist - g_ref).norm() / g_ref.norm())
    worst = float((g_dist - g_ref).abs().max())
    # the SGD step applied to the all-reduced gradient leaves identical weights
    tr.train_step(train.batch_data(mine), apply=True)
    w = [torch.empty_like(tr.flat) for _ in range(world)]
    dist.all_gather(w, tr.flat)
    weights_same = all(torch.equal(w[0], x) for x in w)
    res = {"rank": rank, "world": world, "backend": dist.get_backend(),
           "rel_fro_err": err, "max_abs_err": worst,
           "grad_norm": float(g_ref.norm()), "same_on_all": bool(same_on_all),
           "weights_same": bool(weights_same),
           "loss": {k: out[k] for k in ("cls_loss", "loc_loss", "reg_loss",
                                        "num_endpoint", "num_valid_endpoint")},
           "loss_ref": {k: out_ref[k] for k in ("cls_loss", "loc_loss",
                                                "reg_loss", "num_endpoint",
                                                "num_valid_endpoint")}}
    # one write per record, ranks in turn, so the records do not interleave
    for r in range(world):
        if r == rank:
            sys.stdout.write("MULTIRANK " + json.dumps(res) + "\n")
            sys.stdout.flush()
        dist.barrier()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
