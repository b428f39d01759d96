#!/usr/bin/env python3
"""Static properties of the kernels in the built product library: VGPRs, scratch bytes per lane, LDS, code size -- read from the
code object's metadata notes (llvm-readelf --notes of the gfx950 code object bundled in libtriton_hip.so).
usage: python tools/kernel_static_properties.py [filter-substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "triton_vm_amd", "libtriton_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def main(filters):
    with tempfile.TemporaryDirectory() as tmp:
        # the fat binary sits in the .hip_fatbin section: one offload bundle per translation unit, concatenated
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, f"{tmp}/fatbin"])
        blob = open(f"{tmp}/fatbin", "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        notes = ""
        for k, st in enumerate(starts):
            part = blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)]
            open(f"{tmp}/bundle{k}", "wb").write(part)
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={tmp}/bundle{k}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/co{k}"])
            notes += subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f"{tmp}/co{k}"], text=True)
    rows = []
    for block in notes.split("- .agpr_count:")[1:]:
        get = lambda key: re.search(r"\." + key + r":\s+(\S+)", block)
        name = get("name").group(1).replace(".kd", "")
        try:
            name = subprocess.check_output([os.path.join(LLVM, "llvm-cxxfilt"), name], text=True).strip()
        except Exception:
            pass
        rows.append((name.split("(")[0], int(get("vgpr_count").group(1)), int(get("private_segment_fixed_size").group(1)),
                     int(get("group_segment_fixed_size").group(1)), int(get("sgpr_count").group(1))))
    print("# kernel, VGPRs, scratch B/lane, static LDS B, SGPRs   (from the gfx950 code object of triton_vm_amd/libtriton_hip.so)")
    for r in sorted(rows):
        if not filters or any(f in r[0] for f in filters):
            print(f"  {r[0]:70s} {r[1]:5d} {r[2]:6d} {r[3]:8d} {r[4]:5d}")


if __name__ == "__main__":
    main(sys.argv[1:])
