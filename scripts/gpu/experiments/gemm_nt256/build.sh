#!/bin/bash
# builds libgemm_nt256.so next to this file (linked against libppf_hip.so for the library's error channel)
set -e
HERE=$(cd $(dirname $0) && pwd); ROOT=$(cd $HERE/../../../.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I $ROOT/include -I $ROOT/protopformer_amd/csrc \
    -o $HERE/libgemm_nt256.so $HERE/gemm_nt256.hip -L $ROOT/protopformer_amd/lib -lppf_hip -Wl,-rpath,$ROOT/protopformer_amd/lib
echo built $HERE/libgemm_nt256.so
