#!/bin/bash
# diagnostic: builds the library with one ablation of the ring attention kernel at a time (-DSFM_ABL=n) into gpurun_out/abl/
# and times it at the headline shape.  Ablated kernels compute WRONG results on purpose; only the timing is read.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/_abl
mkdir -p $OUT
CS=$ROOT/sincformer_metacog_speech_enhancement_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I $CS -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -DSFM_ABL=$n -c $CS/attention.hip -o $OUT/attention_$n.o
  objs=$(ls $CS/_obj/*.o | grep -v "/attention.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_abl_$n.so $objs $OUT/attention_$n.o
done
