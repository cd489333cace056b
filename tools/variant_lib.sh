#!/bin/bash
# diagnostic: the library with ONE source rebuilt with extra -D flags -> _abl/lib_<src>_<tag>.so (load it with SFM_LIB_PATH)
# usage: tools/variant_lib.sh ffn_fused stamps -DSFM_FFN_STAMPS
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/_abl
mkdir -p $OUT
CS=$ROOT/sincformer_metacog_speech_enhancement_amd/csrc
src=$1; tag=$2; shift 2
extra=""
case "$src" in attention|attention_pipe) extra="-fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form";; esac
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I $CS $extra "$@" -c $CS/$src.hip -o $OUT/${src}_$tag.o
objs=$(ls $CS/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_${src}_$tag.so $objs $OUT/${src}_$tag.o
echo $OUT/lib_${src}_$tag.so
