"""The Rust side of the drop-in boundary is committed as source (triton-vm-hip/; no cargo in this image): check what can
be checked without a compiler -- the generated `extern "C"` block covers every symbol of include/triton_hip.h with the
right arity, it is up to date with the generator, and the `hip` feature patch applies to the reference tree."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_ffi_block_covers_the_header():
    from tools import gen_rust_ffi as gen
    from triton_vm_amd.capi import EXPORTED_SYMBOLS

    decls = gen.declarations(open(gen.HEADER).read())
    assert {name for _, name, _ in decls} == set(EXPORTED_SYMBOLS)
    ffi = open(gen.OUT).read()
    for ret, name, params in decls:
        m = re.search(r"pub fn " + name + r"\((.*?)\)( -> [^;]+)?;", ffi, re.S)
        assert m, name
        args = [a for a in m.group(1).replace("\n", " ").split(",") if a.strip()]
        assert len(args) == len(params), name
        assert (m.group(2) is None) == (ret == "void"), name


def test_ffi_block_is_up_to_date(tmp_path):
    from tools import gen_rust_ffi as gen

    committed = open(gen.OUT).read()
    out = gen.OUT
    try:
        gen.OUT = str(tmp_path / "ffi.rs")
        gen.main()
        assert open(gen.OUT).read() == committed, "run python tools/gen_rust_ffi.py"
    finally:
        gen.OUT = out


@pytest.mark.skipif(not os.path.isdir("/root/reference/triton-vm"), reason="reference tree not present")
def test_hip_feature_patch_applies_to_the_reference(tmp_path):
    shutil.copytree("/root/reference/triton-vm", tmp_path / "triton-vm")
    subprocess.check_call(["git", "init", "-q", "."], cwd=tmp_path)
    patch = os.path.join(ROOT, "triton-vm-hip", "patches", "triton-vm-hip.patch")
    subprocess.check_call(["git", "apply", "--check", "-p1", patch], cwd=tmp_path)
    text = open(patch).read()
    for seam in ("maybe_low_degree_extend_all_columns", "hash_all_ldt_domain_rows", "hip_all_quotients_combined",
                 "split_and_fold", 'hip = ["dep:triton-vm-hip"]'):
        assert seam in text
