#!/bin/bash
# Test infrastructure: builds the REFERENCE's own GPU extensions for gfx950 -- deformable convolution / deformable PS-RoI
# pooling (/root/reference/assets/ops/dcn/src, reference setup.py: assets/ops/dcn/setup.py:4-15) and the 2D-CTC op
# (/root/reference/ops/ctc_2d/csrc, ops/ctc_2d/setup.py) -- from their sources where they lie, into oracle/_ref/ (git-ignored;
# travels to the GPU box with the snapshot).  The reference's build system (setuptools +
# CUDAExtension + nvcc) is not run: each file is compiled directly with hipcc; oracle/ref_compat/compat.h (force-included) maps
# the few removed PyTorch / CUDA names the 2019 sources use.  The resulting modules (`deform_conv_cuda`, `deform_pool_cuda`:
# the very functions functions/deform_conv.py:5,31-187 and functions/deform_pool.py call) are what tests/test_dcn_reference_gpu.py
# and oracle/gen_golden_dcn.py run on the MI355X to pin oracle/dcn.py, oracle/deform_pool.py and the HIP kernels; `ctc_2d_csrc`
# (ops/ctc_2d/ctc_loss_2d.py:4,15,29) is the checker of tests/test_ctc2d_reference_gpu.py / oracle/gen_golden_ctc2d_ext.py.
# Usage: bash oracle/build_ref_ext.sh   (no-op with a message when /root/reference is absent)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${MEGREADER_REFERENCE:-/root/reference}
SRC=$REF/assets/ops/dcn/src
if [ ! -d "$SRC" ]; then echo "build_ref_ext: $SRC not present, nothing built"; exit 0; fi
OUT="$HERE/_ref"; mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
T=$(python -c "import torch,os;print(os.path.dirname(torch.__file__))")
PYINC=$(python -c "import sysconfig;print(sysconfig.get_paths()['include'])")
ABI=$(python -c "import torch;print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
SUFFIX=$(python -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
TMP=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DTORCH_API_INCLUDE_EXTENSION_H -D_GLIBCXX_USE_CXX11_ABI=$ABI \
  -I$HERE/ref_compat -include $HERE/ref_compat/compat.h -I$T/include -I$T/include/torch/csrc/api/include -I$PYINC \
  -Wno-deprecated-declarations -Wno-unused-result"
for mod in deform_conv_cuda deform_pool_cuda; do
  stamp="$OUT/$mod$SUFFIX"
  if [ -f "$stamp" ] && [ "$stamp" -nt "$SRC/${mod}.cpp" ] && [ "$stamp" -nt "$SRC/${mod}_kernel.cu" ] && [ "$stamp" -nt "$HERE/ref_compat/compat.h" ]; then
    continue
  fi
  $HIPCC $FLAGS -DTORCH_EXTENSION_NAME=$mod -x hip -c "$SRC/${mod}_kernel.cu" -o "$TMP/${mod}_kernel.o"
  $HIPCC $FLAGS -DTORCH_EXTENSION_NAME=$mod -x hip -c "$SRC/${mod}.cpp" -o "$TMP/${mod}.o"
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$stamp" "$TMP/${mod}.o" "$TMP/${mod}_kernel.o" \
    -L$T/lib -Wl,-rpath,$T/lib -lc10 -ltorch -ltorch_cpu -ltorch_python -lc10_hip -ltorch_hip
  echo "built $stamp"
done
# 2D-CTC: csrc/ctc2d.cpp (PYBIND11 module, includes csrc/ctc2d.h -> cuda/ctc2d.h under WITH_CUDA) + the two .cu files
C2=$REF/ops/ctc_2d/csrc
stamp="$OUT/ctc_2d_csrc$SUFFIX"
if [ -d "$C2" ] && ! { [ -f "$stamp" ] && [ "$stamp" -nt "$C2/cuda/ctc2d_cuda_kernel.cu" ] && [ "$stamp" -nt "$HERE/ref_compat/compat.h" ]; }; then
  for f in ctc2d.cpp cuda/ctc2d_cuda.cu cuda/ctc2d_cuda_kernel.cu; do
    $HIPCC $FLAGS -DWITH_CUDA -I$C2 -DTORCH_EXTENSION_NAME=ctc_2d_csrc -x hip -c "$C2/$f" -o "$TMP/ctc2d_$(basename $f).o"
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$stamp" "$TMP"/ctc2d_*.o \
    -L$T/lib -Wl,-rpath,$T/lib -lc10 -ltorch -ltorch_cpu -ltorch_python -lc10_hip -ltorch_hip
  echo "built $stamp"
fi
rm -rf "$TMP"
