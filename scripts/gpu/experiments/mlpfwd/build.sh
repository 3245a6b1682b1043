#!/bin/bash
# builds libmlpfwd.so next to this file (the kernel needs the library's error channel: linked against libppf_hip.so)
set -e
HERE=$(cd $(dirname $0) && pwd); ROOT=$(cd $HERE/../../../.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I $ROOT/include -I $ROOT/protopformer_amd/csrc -I $HERE \
    -o $HERE/libmlpfwd.so $HERE/mlpfwd.hip -L $ROOT/protopformer_amd/lib -lppf_hip -Wl,-rpath,$ROOT/protopformer_amd/lib
echo built $HERE/libmlpfwd.so
