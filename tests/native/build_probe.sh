#!/bin/bash
# Build a cycle-stamped copy of the library + harness (UNITER_GEMM_PROBE) into probe_bin/ for phase-level GEMM timing:
#   bash tests/native/build_probe.sh && gpurun -- 'probe_bin/test_probe --one fwd 3072 3072 768 0 1 20'
set -e
cd "$(dirname "$0")/../.."
mkdir -p probe_bin /tmp/uniter_probe_build
for f in capi gemm attention layernorm embed adamw encoder comm ot pool lmhead head; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DNDEBUG -DUNITER_GEMM_PROBE -DUNITER_ATTN_PROBE -c uniter_amd/csrc/$f.hip -o /tmp/uniter_probe_build/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o probe_bin/libuniter_hip.so /tmp/uniter_probe_build/*.o -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -DUNITER_GEMM_PROBE -x hip tests/native/test_kernels.cpp -o probe_bin/test_probe -Lprobe_bin -luniter_hip -Wl,-rpath,'$ORIGIN'
echo built probe_bin/test_probe
