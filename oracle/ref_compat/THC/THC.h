// stand-in for the umbrella header PyTorch removed (see ../compat.h); ops/ctc_2d/csrc/cuda/ctc2d_cuda.cu:5 includes it and
// uses nothing from it
#pragma once
