"""Build libtriton_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m triton_vm_amd.build          # or: from triton_vm_amd.build import build; build()

The built library is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtriton_hip.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libtriton_hip.so can only be built with the ROCm toolchain")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=False, extra_flags=(), variant=None, recompile=()):
    """Compile the .hip translation units for gfx950 (those whose object is older than its source or any header; all of them
    with force) and link the shared library.  `recompile`: translation units (file names) that are compiled again even when
    up to date, followed by a relink -- __graft_entry__.build() names the row-hashing unit, so that the library a machine
    loads always contains a hot kernel compiled and linked BY THAT MACHINE's toolchain, not only prebuilt objects.
    `variant` (a name) builds an experiment library libtriton_hip_<variant>.so with extra flags.
    Serialised by a file lock: the ranks of a multi-process launch may all find the library stale at once."""
    import fcntl

    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force, verbose, extra_flags, variant, tuple(recompile))


def _build(force, verbose, extra_flags, variant, recompile=()):
    lib = LIB if variant is None else os.path.join(HERE, f"libtriton_hip_{variant}.so")
    if variant is None and not force and not recompile and not _stale():
        return LIB
    objdir = os.path.join(HERE, "build" if variant is None else f"build_{variant}")
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "-mllvm", "-amdgpu-mfma-vgpr-form",  # hash.hip: matrix-core results stay in VGPRs (no accvgpr moves)
             "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-Wno-unused-variable",
             *extra_flags]
    procs, objs = [], []
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    stamp = os.path.join(objdir, ".flags")   # objects are reused only if they were compiled with the same flags
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.basename(src) not in recompile and same_flags and os.path.exists(obj) and \
                os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            continue   # this translation unit is up to date
        cmd = [_hipcc(), *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib + ".tmp", *objs])
    os.replace(lib + ".tmp", lib)  # never a half-written library under the final name
    return lib


HOST_LIB = os.path.join(HERE, "libtriton_host.so")


def build_host(backend_lib=None, out=None):
    """The C++ host side above the C ABI (triton_vm_amd/host/): plain g++, no device code.  It is linked against the
    backend library it is to drive (the product library by default; the test-only emulation passes its own)."""
    backend_lib = backend_lib or build()
    out = out or HOST_LIB
    assert os.path.dirname(os.path.abspath(out)) == os.path.dirname(os.path.abspath(backend_lib)), "host library next to its backend"
    srcs = [os.path.join(HERE, "host", "triton_host.cpp"), os.path.join(HERE, "host", "sharded_host.cpp")]
    # (not the backend library's mtime: the host binds to it dynamically through the C ABI, which include/triton_hip.h states --
    # a relinked backend with the same header does not make the host stale)
    deps = srcs + [os.path.join(HERE, "host", "triton_host.hpp"), os.path.join(HERE, "host", "host_internal.hpp"),
                   os.path.join(ROOT, "include", "triton_hip.h")]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    import fcntl

    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        name = os.path.basename(backend_lib)
        assert name.startswith("lib") and name.endswith(".so")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"),
                               "-o", out + ".tmp", *srcs, "-L", os.path.dirname(backend_lib), "-l" + name[3:-3],
                               "-Wl,-rpath,$ORIGIN"])  # the backend sits next to it
        os.replace(out + ".tmp", out)
    return out


RCCL_LIB = os.path.join(HERE, "libtriton_rccl.so")


class RcclUnavailable(RuntimeError):
    """this machine has no RCCL headers / library: libtriton_rccl.so (the multi-GPU communicator only) cannot be built"""


def rccl_available(rocm=None):
    rocm = rocm or os.environ.get("ROCM_PATH", "/opt/rocm")
    return os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")) and any(
        os.path.exists(os.path.join(rocm, "lib", n)) for n in ("librccl.so", "librccl.so.1"))


def build_rccl(backend_lib=None):
    """The RCCL communicator of the multi-GPU prover (triton_vm_amd/host/rccl_comm.cpp): host code that enqueues RCCL
    collectives on the context's stream.  g++ against the ROCm headers; linked with the product library, librccl and
    libamdhip64 (in a process that has imported torch, torch's bundled copies of those two are the ones already loaded)."""
    backend_lib = backend_lib or build()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    src = os.path.join(HERE, "host", "rccl_comm.cpp")
    deps = [src, os.path.join(HERE, "host", "triton_host.hpp"), os.path.join(ROOT, "include", "triton_hip.h")]
    if not rccl_available(rocm):
        raise RcclUnavailable(f"no RCCL under {rocm} (include/rccl/rccl.h, lib/librccl.so): the multi-GPU communicator is not built; "
                              "the single-GPU product (libtriton_hip.so, libtriton_host.so) does not need it")
    if os.path.exists(RCCL_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(RCCL_LIB) for d in deps):
        return RCCL_LIB
    import fcntl

    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(HERE, "host"), "-I", os.path.join(rocm, "include"), "-o", RCCL_LIB + ".tmp", src,
                               "-L", os.path.dirname(backend_lib), "-ltriton_hip", "-L", os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64",
                               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")])
        os.replace(RCCL_LIB + ".tmp", RCCL_LIB)
    return RCCL_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host())
    print(build_rccl() if rccl_available() else "libtriton_rccl.so: skipped (no RCCL on this machine; only the multi-GPU communicator needs it)")
