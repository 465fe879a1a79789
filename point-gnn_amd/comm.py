"""The training step's collectives: an RCCL communicator behind the C ABI
(include/pointgnn_hip.h "collectives", csrc/comm.hip).

One process per GPU replaces the reference's in-process towers
(train.py:178-181): `average_gradients` (util/tf_util.py:3-43) behind
`unify_copies` (train.py:264-288) is one all-reduce(sum) of the flat gradient
buffer, the endpoint counts one more of two scalars.  PyTorch is the launcher
and the device-array container only: the 128 id bytes travel through
`torch.distributed`'s store / a broadcast (any backend, `gloo` included), a
file, or the caller's own channel; the collectives themselves are
`pgnn_allreduce_*` calls enqueued on the step's stream.

    comm = Communicator.from_env()            # RANK / WORLD_SIZE / MASTER_*
    comm = Communicator.from_torch(group)     # an initialised process group
    comm = Communicator.from_file(path, world, rank)
    comm = Communicator.single()              # one rank: still enters RCCL
"""
import ctypes
import os
import time

import torch

from . import _lib

ID_BYTES = 128   # PGNN_COMM_ID_BYTES

__all__ = ["Communicator", "ID_BYTES"]


class Communicator(object):
    """An RCCL communicator of `world` ranks bound to the CURRENT device."""

    def __init__(self, id_bytes, world, rank):
        if len(id_bytes) != ID_BYTES:
            raise ValueError("a communicator id is %d bytes" % ID_BYTES)
        if not torch.cuda.is_available():
            raise _lib.PointGnnHipError(
                "Communicator needs a GPU (RCCL; there is no CPU fallback)")
        self.lib = _lib.load()
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device("cuda", torch.cuda.current_device())
        buf = ctypes.create_string_buffer(bytes(id_bytes), ID_BYTES)
        h = ctypes.c_void_p(0)
        _lib.check(self.lib.pgnn_comm_init_rank(
            buf, self.world, self.rank, ctypes.byref(h)),
            "pgnn_comm_init_rank")
        self.handle = h

    # ---- construction -------------------------------------------------------
    @staticmethod
    def unique_id():
        """PGNN_COMM_ID_BYTES bytes identifying a new communicator (rank 0
        draws them, every rank passes the same bytes to the constructor)."""
        buf = ctypes.create_string_buffer(ID_BYTES)
        _lib.check(_lib.load().pgnn_comm_unique_id(buf), "pgnn_comm_unique_id")
        return buf.raw

    @classmethod
    def single(cls):
        return cls(cls.unique_id(), 1, 0)

    @classmethod
    def from_torch(cls, group=None):
        """Ranks of an initialised torch.distributed group (any backend): the
        id bytes go out by one broadcast_object_list from the group's rank 0."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
        return cls(box[0], world, rank)

    @classmethod
    def from_file(cls, path, world, rank, timeout=120.0):
        """Rank 0 writes the id to `path` (atomically), the others wait for
        it: a launcher with a shared directory and nothing else."""
        if rank == 0:
            uid = cls.unique_id()
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)
        else:
            t0 = time.time()
            while not (os.path.exists(path) and
                       os.path.getsize(path) == ID_BYTES):
                if time.time() - t0 > timeout:
                    raise TimeoutError("no communicator id at %s" % path)
                time.sleep(0.01)
            with open(path, "rb") as f:
                uid = f.read()
        return cls(uid, world, rank)

    @classmethod
    def from_env(cls):
        """torchrun's environment (RANK, WORLD_SIZE, MASTER_ADDR / _PORT):
        the id goes through a TCPStore on MASTER_PORT + 1 -- no process group."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        if world == 1:
            return cls.single()
        import datetime
        from torch.distributed import TCPStore
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"),
                         int(os.environ.get("MASTER_PORT", "29500")) + 1,
                         world, rank == 0,
                         timeout=datetime.timedelta(seconds=120))
        if rank == 0:
            store.set("pgnn_comm_id", cls.unique_id())
        return cls(bytes(store.get("pgnn_comm_id")), world, rank)

    # ---- collectives (in place, on torch's current stream) --------------------
    def _check(self, t, dtype):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and
                t.dtype == dtype and t.is_contiguous()):
            raise ValueError("collectives take contiguous %s CUDA tensors"
                             % (dtype,))

    def allreduce_sum(self, t):
        if t.dtype == torch.float32:
            fn, name = self.lib.pgnn_allreduce_sum_f32, "pgnn_allreduce_sum_f32"
        elif t.dtype == torch.float64:
            fn, name = self.lib.pgnn_allreduce_sum_f64, "pgnn_allreduce_sum_f64"
        else:
            raise ValueError("allreduce_sum: float32 / float64 only")
        self._check(t, t.dtype)
        _lib.check(fn(self.handle, _lib.ptr(t), t.numel(), _lib.stream_ptr()),
                   name)
        return t

    def allreduce_step(self, grads, sums=None):
        """The flat gradient (fp32) and the loss sums (float64) as one RCCL
        group."""
        self._check(grads, torch.float32)
        if sums is not None:
            self._check(sums, torch.float64)
        _lib.check(self.lib.pgnn_allreduce_step(
            self.handle, _lib.ptr(grads), grads.numel(), _lib.ptr(sums),
            sums.numel() if sums is not None else 0, _lib.stream_ptr()),
            "pgnn_allreduce_step")
        return grads

    def broadcast(self, t, root=0):
        self._check(t, torch.float32)
        _lib.check(self.lib.pgnn_broadcast_f32(
            self.handle, _lib.ptr(t), t.numel(), int(root),
            _lib.stream_ptr()), "pgnn_broadcast_f32")
        return t

    def check_async_error(self):
        _lib.check(self.lib.pgnn_comm_async_error(self.handle),
                   "pgnn_comm_async_error")

    # ---- info -----------------------------------------------------------------
    @staticmethod
    def rccl_version():
        v = ctypes.c_int32(0)
        _lib.check(_lib.load().pgnn_comm_info(None, None, None,
                                              ctypes.byref(v)),
                   "pgnn_comm_info")
        return int(v.value)

    @staticmethod
    def library():
        return _lib.load().pgnn_comm_library().decode()

    def destroy(self):
        h, self.handle = self.handle, None
        if h is not None and h.value:
            _lib.check(self.lib.pgnn_comm_destroy(h), "pgnn_comm_destroy")

    def __del__(self):
        # (not at interpreter shutdown: the HIP runtime and RCCL may be gone
        # by then, and a communicator dies with its process anyway)
        import sys
        if sys is None or sys.is_finalizing():
            return
        try:
            self.destroy()
        except Exception:
            pass
