#!/usr/bin/env python
"""Physical placement of CU-masked streams (pgnn_stream_create_cu_mask): for a
few masks, which (XCC, SE, CU) do the workgroups land on?"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    probe = ctypes.CDLL(os.path.join(ROOT, "ab", "libcuprobe.so"))
    probe.cu_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p]
    lib = _lib.load()
    dev = torch.device("cuda")
    blocks = 4096
    out = torch.zeros(2 * blocks, dtype=torch.int32, device=dev)

    def run(stream_handle, label):
        out.zero_()
        torch.cuda.synchronize()
        probe.cu_probe_launch(out.data_ptr(), blocks, 256, 20000, stream_handle)
        torch.cuda.synchronize()
        v = out.view(-1, 2).cpu().numpy().astype("uint32")
        xcc = v[:, 0] & 0xF
        hw = v[:, 1]
        # HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
        cu = (hw >> 8) & 0xF
        sh = (hw >> 12) & 0x1
        se = (hw >> 13) & 0x7
        places = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(),
                                         cu.tolist()))
        per_xcc = collections.Counter()
        for (x, _, _, _) in places:
            per_xcc[x] += 1
        print("%-28s distinct CUs %3d  per XCC %s" % (
            label, len(places), [per_xcc[i] for i in range(8)]))
        return set(places)

    run(None, "null stream")
    sets = {}
    for cu_first, count, comp in ((0, 8, 0), (0, 8, 1), (0, 16, 0), (0, 16, 1),
                                  (0, 32, 0), (0, 1, 0), (1, 1, 0), (8, 1, 0),
                                  (32, 1, 0)):
        p = ctypes.c_void_p()
        _lib.check(lib.pgnn_stream_create_cu_mask(cu_first, count, comp,
                                                  ctypes.byref(p)), "mask")
        s = run(p.value, "mask first=%d count=%d comp=%d" % (cu_first, count, comp))
        sets[(cu_first, count, comp)] = s
        if count == 1:
            print("     ->", sorted(s))
        torch.cuda.synchronize()
        _lib.check(lib.pgnn_stream_destroy(p.value), "destroy")
    for n in (8, 16):
        a, b = sets[(0, n, 0)], sets[(0, n, 1)]
        print("mask %d vs its complement: overlap %d CUs, union %d" % (
            n, len(a & b), len(a | b)))


if __name__ == "__main__":
    main()
