#!/usr/bin/env python3
"""profiles/kernel_counters.json from a tools/pmc.sh summary of `tools/probe.py 20 <cols> 0 1` (the LDE of one chunk of <cols> <= 96
main-table columns at 2^20 rows, then row hashing + Merkle tree of that table): per-dispatch hardware counters of the SHIPPED hot
kernels, turned into the per-unit figures bench.py's roofline objects multiply up --

  lde         fabric bytes per trace cell = (2 * FETCH_SIZE + WRITE_SIZE) KiB summed over the kernels of the family / cells
              (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; Infinity-Cache hits are counted, so an upper
              bound on HBM bytes), and wave-level VALU instructions per trace cell = sum SQ_INSTS_VALU / cells
  hash_rows   SQ_INSTS_VALU per row and Tip5 permutation (the probe's table has <cols> words per row: cols // 10 + 1 permutations),
              and the share of LDS-active cycles lost to bank conflicts

Usage: python tools/kernel_counters.py <summary file> [columns per dispatch = 96] [log2 rows = 20] [commit of the profiled tree]
       [expansion = 8] [shape key: merge as a further shape, e.g. 2p22_x8]"""
import json
import re
import sys


def blocks(txt):
    out, name = {}, None
    for line in txt.splitlines():
        if line and not line.startswith((" ", "#")):
            name = line.strip().replace("void ", "").replace("tvm::", "")
            out[name] = {}
        elif name and line.startswith("    "):
            m = re.match(r"\s+(\w+)\s+avg\s+([0-9.]+)\s+dispatches\s+(\d+)", line)
            if m:
                out[name][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


def main(path, cols=96, log2_rows=20, expansion=8, commit=None, shape=None):
    b = blocks(open(path).read())
    cells = cols << log2_rows
    src = f"{path}: rocprofv3 --kernel-trace --pmc in separate passes (tools/pmc.sh) over `tools/probe.py {log2_rows} {cols} 0 1`" + \
        (f" with TVM_PROBE_EXPANSION={expansion}" if expansion != 8 else "")
    lde, total_bytes, total_valu = {}, 0.0, 0.0
    for name, c in b.items():
        if not re.match(r"k_lde_pass[123]|k_ntt2_pass1", name) or "FETCH_SIZE" not in c:   # (k_ntt2_pass1: pass 1 where no row kernel applies)
            continue
        n_disp = c["SQ_INSTS_VALU"][1]
        f, w = 2 * c["FETCH_SIZE"][0] * 1024 * n_disp, c["WRITE_SIZE"][0] * 1024 * n_disp
        v = c["SQ_INSTS_VALU"][0] * n_disp
        lde[name] = {"fetch_bytes_per_cell": round(f / cells, 1), "write_bytes_per_cell": round(w / cells, 1),
                     "wave_valu_instructions_per_dispatch": int(c["SQ_INSTS_VALU"][0]), "dispatches": n_disp,
                     "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"][0] / max(c["SQ_LDS_IDX_ACTIVE"][0], 1), 3)}
        total_bytes += f + w
        total_valu += v
    out = {"commit": commit, "lde": {"source": src, "columns_per_dispatch": cols, "hbm_bytes_per_trace_cell": round(total_bytes / cells, 1),
                   "wave_valu_instructions_per_trace_cell": round(total_valu / cells, 3), "kernels": lde}}
    h = b.get("k_hash_rows_mfma")
    if h and "SQ_INSTS_VALU" in h:
        rows, perms = expansion << log2_rows, cols // 10 + 1
        out["hash_rows"] = {"source": src, "rows": rows, "permutations_per_row": perms,
                            "wave_valu_instructions_per_row_permutation": round(h["SQ_INSTS_VALU"][0] * h["SQ_INSTS_VALU"][1] / (rows * perms), 3),
                            "lds_instructions_per_row_permutation": round(h["SQ_INSTS_LDS"][0] / (rows * perms), 3) if "SQ_INSTS_LDS" in h else None,
                            "lds_bank_conflict_share": round(h["SQ_LDS_BANK_CONFLICT"][0] / max(h["SQ_LDS_IDX_ACTIVE"][0], 1), 3)}
    if shape:   # a further shape (e.g. "2p22_x8", "2p20_x32"), merged into the file of the default shape (2^20 rows, expansion 8)
        whole = json.load(open("profiles/kernel_counters.json"))
        whole.setdefault("shapes", {})[shape] = out
        out = whole
    else:
        try:
            out["shapes"] = json.load(open("profiles/kernel_counters.json")).get("shapes", {})
        except (OSError, ValueError):
            pass
    json.dump(out, open("profiles/kernel_counters.json", "w"), indent=1)
    print(json.dumps(out if not shape else out["shapes"][shape], indent=1))


if __name__ == "__main__":
    # [4]: the commit of the tree the counters were taken on (the kernels that were profiled), recorded in the file
    # [5] [6]: expansion of the probe's evaluation domain and a shape key -- further shapes are merged into the file under "shapes"
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 96, int(sys.argv[3]) if len(sys.argv) > 3 else 20,
         commit=sys.argv[4] if len(sys.argv) > 4 else None, expansion=int(sys.argv[5]) if len(sys.argv) > 5 else 8,
         shape=sys.argv[6] if len(sys.argv) > 6 else None)
