// Dispatcher of the ping-pong 32x32x16 NT kernel (igemm_nt32.h); the instantiations live in nt32_s<shape>.hip.
#include "nt32.h"

namespace mr {

int launch_nt32_s1(int, int, const NtArgs&, const ConvGeom&, const EpiStore<bf16_t>&, hipStream_t);
int launch_nt32_s2(int, int, const NtArgs&, const ConvGeom&, const EpiStore<bf16_t>&, hipStream_t);
int launch_nt32_s3(int, int, const NtArgs&, const ConvGeom&, const EpiStore<bf16_t>&, hipStream_t);
int launch_nt32_s4(int, int, const NtArgs&, const ConvGeom&, const EpiStore<bf16_t>&, hipStream_t);

int nt32_tile(int shape, int* bm, int* bn) {
  switch (shape) {
    case 1: *bm = 256; *bn = 256; return 1;
    case 2: *bm = 288; *bn = 256; return 1;
    case 3: *bm = 256; *bn = 128; return 1;
    case 4: *bm = 128; *bn = 256; return 1;
  }
  return 0;
}

int launch_nt32(int shape, int variant, int amode, const NtArgs& a, const ConvGeom& g, const EpiStore<bf16_t>& epi,
                hipStream_t stream) {
  switch (shape) {
    case 1: return launch_nt32_s1(variant, amode, a, g, epi, stream);
    case 2: return launch_nt32_s2(variant, amode, a, g, epi, stream);
    case 3: return launch_nt32_s3(variant, amode, a, g, epi, stream);
    case 4: return launch_nt32_s4(variant, amode, a, g, epi, stream);
  }
  set_error("launch_nt32: unknown shape %d", shape);
  return MR_ERR_ARG;
}

}  // namespace mr
