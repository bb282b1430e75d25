/* Timing-only ablation switches of the MFMA kernels -- NOT part of the product C ABI.
 *
 * These exist only in libmegreader_hip_abl.so (`make -C megreader_amd/csrc ablation`, built with -DMR_ABLATION), which the
 * measurement scripts under tools/ load through MEGREADER_HIP_LIB.  Every non-zero mask deletes an ingredient of a kernel
 * (LDS-DMA staging, fragment reads, the atomic epilogue, ...) to time what it costs: the RESULTS ARE WRONG BY CONSTRUCTION.
 * The product library (libmegreader_hip.so) neither exports these symbols nor contains the ablated kernel instantiations. */
#ifndef MEGREADER_HIP_ABLATION_H
#define MEGREADER_HIP_ABLATION_H
#ifdef __cplusplus
extern "C" {
#endif
/* ablation mask of the 128x128 TN kernel: 1 no LDS-DMA, 2 no fragment reads, 4 no atomic epilogue, 8 no column sums */
int mr_set_tn_abl(int mask);
/* ablation mask of the all-taps wgrad kernel (csrc/tn_taps.hip: ABL template parameter) */
int mr_set_tn_taps_abl(int mask);
/* mr_tuning.nt_p8(2..4): no LDS-DMA / no fragment reads / MFMA + barriers only variants of the phased 256x256 NT kernel */
#ifdef __cplusplus
}
#endif
#endif
