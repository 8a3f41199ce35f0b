#!/bin/bash
# Dev container only: compile tools/cpp_archive/make_cpp_archive.cpp against the libtorch inside the pip torch wheel.
# Output: tools/cpp_archive/bin/make_cpp_archive (git-ignored).
set -e
cd "$(dirname "$0")"
TORCH=$(python -c 'import torch, os; print(os.path.dirname(torch.__file__))')
ABI=$(python -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')
mkdir -p bin
g++ -O1 -std=c++17 -D_GLIBCXX_USE_CXX11_ABI=$ABI -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" \
    make_cpp_archive.cpp -o bin/make_cpp_archive -L"$TORCH/lib" -Wl,-rpath,"$TORCH/lib" -ltorch -ltorch_cpu -lc10
