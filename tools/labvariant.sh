#!/bin/bash
# tools/labvariant.sh NAME SRC.hip [-Dflags...]  ->  tools/lab/liblwg_NAME.so: the whole library with ONE source rebuilt under extra flags
# (the other objects are the tree's: run `make -C ipercore_amd/csrc` first)
cd "$(dirname "$0")/../ipercore_amd/csrc" || exit 1
mkdir -p ../../tools/lab
name=$1; src=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$src" -o "/tmp/lab_$name.o" || exit 1
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../../tools/lab/liblwg_$name.so" $objs "/tmp/lab_$name.o"
