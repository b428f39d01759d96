# round 6: the row hashing's distance from the floor of its own instruction stream (tools/ubench/tip5_floor.hip), on one box
export TMPDIR=/tmp
mkdir -p gpurun_out tools/ubench/bin
[ -x tools/ubench/bin/tip5_floor ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I triton_vm_amd/csrc -mllvm -amdgpu-mfma-vgpr-form \
    tools/ubench/tip5_floor.hip -L triton_vm_amd -ltriton_hip -Wl,-rpath,$PWD/triton_vm_amd -o tools/ubench/bin/tip5_floor
for i in 1 2 3; do timeout 300 tools/ubench/bin/tip5_floor 20; done > gpurun_out/${1:-r06}_tip5_floor.txt 2>&1
cat gpurun_out/${1:-r06}_tip5_floor.txt
