"""TEST INFRASTRUCTURE (oracle): a CPU restatement of the part of Triton VM that lies *before* the prover's hot path
-- ISA encoding, the VM's trace execution, the algebraic execution trace, and the master tables' fill / pad / extend --
in plain python integers.  It exists so that (a) the restated AIR can be checked on VALID traces (all constraints
vanish, the reference's own strongest AIR test, stark.rs:4186-4255), (b) the device-side `extend` / `pad` / `fill`
kernels have an oracle, and (c) reference-held snapshots (program digests, proof digests) can be
reproduced.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it."""
