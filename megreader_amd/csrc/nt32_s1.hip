// ping-pong 32x32x16 NT kernel, tile shape 1 (see nt32_impl.h)
#define MR_NT32_SHAPE 1
#include "nt32_impl.h"
