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
python3 -c "import sys; sys.path.insert(0, '$REPO'); from triton_vm_amd.build import build, build_host; build(); build_host()"
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
# the headline benchmark, CPU vs device, same box
cargo bench -p triton-vm --bench prove_fib --no-default-features
cargo bench -p triton-vm --bench prove_fib --no-default-features --features hip
