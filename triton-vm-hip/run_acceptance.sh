#!/bin/bash
# One command for the first machine that has cargo AND an MI355X: apply the `hip` feature patch to a checkout of
# TritonVM/triton-vm, build the backend, and run the reference's own acceptance tests over it --
#   * the proof-hash snapshots (stark.rs `supplying_prover_randomness_seed_fully_derandomizes_produced_proof`,
#     proof.rs `current_proof_version_is_still_current`): the backend must emit the reference's proof, word for word;
#   * every prove_and_verify_* test (stark.rs:4257-4317) incl. the STIR ones: the UNMODIFIED verifier must accept;
# once with stage 2 (the whole Prover::prove behind one call, tables resident on the device) and once with stage 1
# (TRITON_HIP_STAGE=1: the reference's control flow, four seams on the device).
# usage: triton-vm-hip/run_acceptance.sh <path to a triton-vm checkout>   (nothing here has been run: no cargo in the authoring image)
set -euo pipefail
CHECKOUT=${1:?path to a TritonVM/triton-vm checkout}
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(dirname "$HERE")
python3 -c "import sys; sys.path.insert(0, '$REPO'); from triton_vm_amd.build import build, build_host, build_rccl, rccl_available; build(); build_host(); print(build_rccl() if rccl_available() else 'no RCCL here: libtriton_rccl.so skipped (cargo feature rccl stays off)')"
cd "$CHECKOUT"
git apply --check -p1 "$HERE/patches/triton-vm-hip.patch" && git apply -p1 "$HERE/patches/triton-vm-hip.patch"
# point the dependency at this crate
sed -i "s#triton-vm-hip = { path = \"[^\"]*\"#triton-vm-hip = { path = \"$HERE\"#" triton-vm/Cargo.toml
export TRITON_HIP_LIB_DIR="$REPO/triton_vm_amd" LD_LIBRARY_PATH="$REPO/triton_vm_amd:${LD_LIBRARY_PATH:-}"
TESTS="supplying_prover_randomness_seed_fully_derandomizes_produced_proof current_proof_version_is_still_current prove_and_verify constraints_evaluate_to_zero"
for STAGE in 2 1; do
  echo "== feature hip, stage $STAGE"
  for T in $TESTS; do
    TRITON_HIP_STAGE=$STAGE cargo test --release -p triton-vm --features hip -- "$T"
  done
done
# the same acceptance tests through the sharded / coset-wise entry point (tvmh_prove_execution_sharded, comm = NULL): under the memory
# policy (0) and coset by coset in two passes (2) -- the proof must not change, so the snapshot tests pass unchanged
for PASSES in 0 2; do
  echo "== feature hip, stage 2 through tvmh_prove_execution_sharded, jit_passes = $PASSES"
  for T in $TESTS; do
    TRITON_HIP_SHARDED_ENTRY=$PASSES cargo test --release -p triton-vm --features hip,triton-vm-hip/acceptance-sharded-entry -- "$T"
  done
done
# STIR, the reference's automatic low-degree test from 2^16 padded rows on (stark.rs:1944-1951) -- the one part of the proof no
# reference-held value pins (both proof-hash snapshots are FRI-sized): prove_fib at 2^16 rows through the device backend, the
# UNMODIFIED verifier must accept.  FIBONACCI_INDEX 6500 -> ~65 000 cycles -> padded height 2^16; the bench's own
# `program.prove()` verifies nothing, so the example-style check runs as a doc-less integration test generated here.
mkdir -p triton-vm/tests
cat > triton-vm/tests/hip_stir_acceptance.rs <<'RS'
use triton_vm::prelude::*;
#[test]
fn prove_fib_at_2_pow_16_rows_with_the_automatic_stir_is_accepted_by_the_unmodified_verifier() {
    let program = dev_util::example_programs::fibonacci_sequence();
    let (stark, claim, proof) = triton_vm::prove_program(program, PublicInput::new(bfe_vec![6500_u32]), NonDeterminism::default()).unwrap();
    assert_eq!(1 << 16, proof.padded_height().unwrap());
    assert!(triton_vm::verify(stark, &claim, &proof));
}
RS
for STAGE in 2 1; do
  TRITON_HIP_STAGE=$STAGE cargo test --release -p triton-vm --features hip --test hip_stir_acceptance
done
TRITON_HIP_SHARDED_ENTRY=0 cargo test --release -p triton-vm --features hip,triton-vm-hip/acceptance-sharded-entry --test hip_stir_acceptance
# the headline benchmark, CPU vs device, same box; then the same at 2^16 rows, where Stark::default() is STIR
cargo bench -p triton-vm --bench prove_fib --no-default-features
cargo bench -p triton-vm --bench prove_fib --no-default-features --features hip
sed -i 's/const FIBONACCI_INDEX: u32 = 100;/const FIBONACCI_INDEX: u32 = 6500;/' triton-vm/benches/prove_fib.rs
cargo bench -p triton-vm --bench prove_fib --no-default-features
cargo bench -p triton-vm --bench prove_fib --no-default-features --features hip
