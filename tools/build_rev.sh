#!/bin/bash
# Build libperitext_hip.so of a git revision beside the product, as peritext_amd/lib/exp_<name>.so (for same-box A/Bs with tools/lib_ab.py).
# Usage: tools/build_rev.sh <rev> <name> [-DX=1 ...]
set -e
REV=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" peritext_amd/csrc include | tar -x -C "$TMP"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -o "$ROOT/peritext_amd/lib/exp_$NAME.so" "$TMP/peritext_amd/csrc/peritext_hip.hip"
rm -rf "$TMP"
echo "$ROOT/peritext_amd/lib/exp_$NAME.so"
