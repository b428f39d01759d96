export TMPDIR=/tmp
bash tools/r06_tip5_floor.sh r06_c
( timeout 2400 python -m pytest tests/test_kernels_hash.py tests/test_sharded_host.py tests/test_wider_pins.py tests/test_proof_snapshot.py tests/test_error_paths.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r06_c_pytest_gpu.log
cat gpurun_out/r06_c_pytest_gpu.log
