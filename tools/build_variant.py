#!/usr/bin/env python
"""Build ab/lib<name>.so = the tree's library with one (or a comma-separated
list of) translation unit(s) recompiled under extra -D flags (kernel A/B
variants; select at run time with PGNN_LIB=ab/lib<name>.so).  The other
objects come from csrc/build/ (run `python -m pointgnn_amd.build` first).

    python tools/build_variant.py kd1536 kdtree.hip -DKD_TOP_LEN=1536
    python tools/build_variant.py diag core.hip,gnn.hip,graph.hip -DPGNN_DIAG

The second form is the diagnostic build: the only one in which the timing
ablations with wrong results (mlp_debug bits 1/2/4, graph_debug bit 1) exist.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import build as B  # noqa: E402


def main():
    name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build(verbose=False)
    out_dir = os.path.join(ROOT, "ab")
    os.makedirs(out_dir, exist_ok=True)
    units = unit.split(",")
    hipcc = B._hipcc()
    new = []
    for u in units:
        obj = os.path.join(out_dir, "%s_%s.o" % (name, u[:-4]))
        subprocess.check_call([hipcc] + B.CXXFLAGS + flags +
                              ["-c", os.path.join(B.CSRC, u), "-o", obj])
        new.append(obj)
    objs = [os.path.join(B.CSRC, "build", os.path.basename(s)[:-4] + ".o")
            for s in B._sources() if os.path.basename(s) not in units] + new
    lib = os.path.join(out_dir, "lib%s.so" % name)
    subprocess.check_call([hipcc, "-shared", "-fPIC",
                           "--offload-arch=" + B.ARCH, "-o", lib] + objs)
    for obj in new:
        os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main()
