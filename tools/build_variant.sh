#!/bin/bash
# Build a variant of libmrag_hip.so into build_ab/<name>.so with extra compile flags, leaving the in-tree product
# library untouched (select it at run time with MRAG_HIP_LIB):  tools/build_variant.sh zz0 "-DMRAG_ZZ=0"
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab
make -C a-modular-rag-framework_amd/csrc -j8 BUILD=build_$name OUT=../../build_ab/$name.so EXTRA="$*" > build_ab/$name.log 2>&1 || { tail -30 build_ab/$name.log; exit 1; }
rm -rf a-modular-rag-framework_amd/csrc/build_$name
ls -la build_ab/$name.so
