#!/usr/bin/env python3
"""profiles/lde_traffic.json from a tools/pmc.sh summary of `tools/probe.py 20 <cols> 0 1` (one chunk of <cols> <= 96
columns; round 1: two chunks of 32): fabric-side bytes per trace cell of the LDE kernel family = (2 * FETCH_SIZE +
WRITE_SIZE) KiB per dispatch, summed over the three kernels, / (columns per dispatch * 2^20 rows).  FETCH_SIZE is doubled per the gfx950 correction of
MI355X_MICROARCH.md (it reports half the bytes of a coalesced read; checked here on pass 1 and pass 3, whose reads
are known: 8 B and 64 B per cell).  Usage: python tools/lde_traffic.py <summary> [columns per dispatch = 32]"""
import json
import re
import sys


def main(path, cols=32):
    txt = open(path).read()
    total, per_kernel = 0.0, {}
    for head, kernel in re.findall(r"^((?:void )?tvm::(k_ntt2_pass1|k_lde_pass1\w*|k_lde_pass2\w*|k_lde_pass3\w*)(?:<[^>\n]*>)?)$", txt, re.M):
        block = txt.split(head + "\n", 1)[1]
        kernel = head.replace("void ", "").replace("tvm::", "")
        get = lambda c: float(re.search(r"^\s+" + c + r"\s+avg\s+([0-9.]+)", block, re.M).group(1))  # noqa: E731
        f, w = 2 * get("FETCH_SIZE") * 1024, get("WRITE_SIZE") * 1024
        per_kernel[kernel] = {"fetch_bytes_per_cell": round(f / (cols << 20), 1), "write_bytes_per_cell": round(w / (cols << 20), 1)}
        total += f + w
    out = {"source": f"{path} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc.sh over "
                     f"tools/probe.py 20 {cols if cols != 32 else 64} 0 1; FETCH_SIZE doubled per the gfx950 correction; Infinity-Cache hits are "
                     "counted, so this is fabric-side traffic, an upper bound on HBM bytes)",
           "method": "rocprofv3 --kernel-trace --pmc, FETCH_SIZE and WRITE_SIZE in separate passes (tools/pmc.sh); bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB "
                     "per dispatch of one 96-column chunk / (96 * 2^20 trace cells), summed over the kernels of the launch family",
           "measured_on": "the shipped kernels of this commit's tvm_lde_table at 2^20 rows (kernel names below)",
           "hbm_bytes_per_trace_cell": round(total / (cols << 20), 1), "kernels": per_kernel}
    json.dump(out, open("profiles/lde_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32)
