#!/bin/bash
# Build the C++ demo against the in-tree library (python -m sttm_amd.build first).  Cross-compiles without a GPU.
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude examples/c_abi_demo.cpp -Lsttm_amd/lib -lsttm_hip -Wl,-rpath,'$ORIGIN/../sttm_amd/lib' -o examples/c_abi_demo
echo built examples/c_abi_demo
