// stand-in for a header PyTorch removed (see ../compat.h); ops/ctc_2d/csrc/cuda/ctc2d_cuda.cu:6 includes it and uses nothing from it
#pragma once
