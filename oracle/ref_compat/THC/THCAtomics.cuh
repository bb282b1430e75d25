// stand-in for the header PyTorch removed (see ../compat.h): the atomicAdd overloads the reference kernels rely on
#pragma once
#include <ATen/hip/Atomic.cuh>
