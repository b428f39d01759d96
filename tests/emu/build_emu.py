"""Build the TEST-ONLY CPU emulation of the kernels: the same csrc/*.hip sources compiled by g++
against tests/emu/hip_emu.h.  Never loaded by the product package (see hip_emu.h)."""
import fcntl
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "triton_vm_amd", "csrc")
LIB = os.path.join(HERE, "libtriton_hip_emu.so")


def build(force=False):
    """Serialised by a file lock: the ranks of a multi-process test may all find the library stale at once."""
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force)


def _build(force):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "hip_emu.cpp")]
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        [os.path.join(HERE, "hip_emu.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in srcs:
        obj = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(obj)
        cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-pthread", "-DTVM_EMU", "-x", "c++", "-I", HERE, "-I", CSRC, "-I",
               os.path.join(ROOT, "include"), "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-c", s, "-o", obj]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"g++ failed on {s}:\n{out}")
    subprocess.check_call(["g++", "-shared", "-fPIC", "-pthread", "-o", LIB + ".tmp", *objs])
    os.replace(LIB + ".tmp", LIB)  # never a half-written library under the final name
    return LIB


if __name__ == "__main__":
    print(build(force=True))
