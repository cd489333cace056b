#!/bin/bash
# diagnostic: library variants of conv16p.hip with extra -D flags into _abl/lib_convp_<tag>.so   usage: convp_variants.sh tag "-DX=1 ..."
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/_abl
mkdir -p $OUT
CS=$ROOT/sincformer_metacog_speech_enhancement_amd/csrc
tag=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I $CS "$@" -c $CS/conv16p.hip -o $OUT/conv16p_$tag.o
objs=$(ls $CS/_obj/*.o | grep -v "/conv16p.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_convp_$tag.so $objs $OUT/conv16p_$tag.o
