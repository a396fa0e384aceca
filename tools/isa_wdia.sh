#!/bin/bash
# device ISA of spmv_wdia.hip -> /tmp/wdia.s, register / spill summary of the half-box kernel
cd /root/repo/spmv_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I/root/repo/include --cuda-device-only -S hip/spmv_wdia.hip -o /tmp/wdia.s $1 2>&1 | grep -v "hip-link" | head
grep -n "\.vgpr_count\|\.vgpr_spill_count\|\.sgpr_spill_count\|\.name:" /tmp/wdia.s | grep -A3 "box27_half" | grep -v "^--" | awk '{print $NF}' | paste - - - -
awk '/^_ZN12_GLOBAL__N_121csr_box27_half_kernelIddLb0ELb0EEE.*:/{f=1} f{print} /s_endpgm/{if(f)exit}' /tmp/wdia.s > /tmp/hb.s
