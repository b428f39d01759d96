"""Test helper: a Context backed by the TEST-ONLY fiber emulation of the kernels (tests/emu/).
The product package never loads this library."""
from tests.emu.build_emu import build as build_emu


def emu_context():
    from triton_vm_amd.capi import Context, load_library

    return Context(device=0, lib=load_library(build_emu()))
