#!/bin/bash
# Builds the diagnostic variant of the engine (s_memtime stamps in the tree kernels for ONE game, -DFPC_TREE_STAMPS=<block>)
# next to the product library; run it on the GPU box with tools/tree_stamps.py:
#   bash tools/tree_stamps.sh 7 && gpurun -- 'FPC_ENGINE_LIB=$PWD/tools/lib_tree_stamps.so python3 tools/tree_stamps.py 400 14'
# (the .so is git-ignored; delete it afterwards: everything under the repo travels to the GPU box)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DFPC_TREE_STAMPS=${1:-7} -x hip \
  alphazero-4-player-chess_amd/csrc/fpc_engine.cpp -o tools/lib_tree_stamps.so
ls -la tools/lib_tree_stamps.so
