// ping-pong 32x32x16 NT kernel, tile shape 2 (see nt32_impl.h)
#define MR_NT32_SHAPE 2
#include "nt32_impl.h"
