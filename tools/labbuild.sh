#!/bin/bash
# tools/labbuild.sh SRC.hip NAME [-Dflags...]  ->  tools/lab/NAME.so (an experimental single-kernel build for tools/convlab.py)
cd "$(dirname "$0")/../ipercore_amd/csrc" || exit 1
mkdir -p ../../tools/lab
src=$1; name=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I. -I../../include "$@" "$src" -o "../../tools/lab/$name.so" 2>&1 | grep -iE "error|warning: v|spill" | head -5
