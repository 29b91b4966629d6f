#!/bin/bash
# A variant build of the library for an A/B of two builds: scratch/build_variant.sh <name> "<extra flags for kernels_svgf.hip>" ["<extra flags for kernels_trace.hip>"]
# -> scratch/_variants/libvhr_<name>.so (git-ignored; picked with VHR_LIB_VARIANT or scratch/ab_atrous_libs.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/vulkanhybridrenderer_amd/csrc
make -C $C -s -j8 > /dev/null
O=$R/scratch/_variants/obj_$1; mkdir -p $O
COMMON="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -I$R/include -I$C -Wno-unused-value"
cp $C/build/*.o $O/
[ -n "$2" ] && /opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize $2 -c $C/kernels_svgf.hip -o $O/kernels_svgf.o
[ -n "$3" ] && /opt/rocm/bin/hipcc $COMMON -ffp-contract=off -fno-slp-vectorize $3 -c $C/kernels_trace.hip -o $O/kernels_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/_variants/libvhr_$1.so $O/*.o -ldl
rm -rf $O
echo scratch/_variants/libvhr_$1.so
