#!/usr/bin/env python3
"""EXPERIMENT libraries: libtriton_hip_<name>.so = the product's objects with ntt.hip recompiled under extra -D flags (timing
experiments of the LDE kernels: what bounds pass 2?).  Usage: python tools/build_ntt_variants.py name=-DFLAG[,-DFLAG] ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triton_vm_amd import build as B  # noqa: E402


def main(specs):
    B.build()
    objdir = os.path.join(B.HERE, "build")
    flags = open(os.path.join(objdir, ".flags")).read().split(" ")
    procs = []
    for spec in specs:
        name, defs = spec.split("=", 1)
        vdir = os.path.join(B.HERE, f"build_{name}")
        os.makedirs(vdir, exist_ok=True)
        obj = os.path.join(vdir, "ntt.o")
        procs.append((name, obj, subprocess.Popen([B._hipcc(), *flags, *defs.split(","), "-c", os.path.join(B.CSRC, "ntt.hip"), "-o", obj])))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        objs = [obj if os.path.basename(o) == "ntt.o" else o for o in
                (os.path.join(objdir, os.path.basename(s).replace(".hip", ".o")) for s in B.sources())]
        lib = os.path.join(B.HERE, f"libtriton_hip_{name}.so")
        subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib, *objs])
        print(lib)


if __name__ == "__main__":
    main(sys.argv[1:])
