/* megreader_hip.h -- C ABI of libmegreader_hip.so (MI355X / gfx950 kernels for MegReader's training hot path).
 *
 * Every entry point takes raw device pointers, plain sizes and a hipStream_t; no torch types cross this
 * boundary.  The library owns no tensors: the caller allocates every output and scratch buffer (its only state
 * is one 4 KiB zero page per device, the source of padded vectors for direct-to-LDS loads).  All functions return 0 on success and a non-zero MR_ERR_* code otherwise; mr_last_error() returns
 * a human-readable message for the calling thread.  Kernels are enqueued on `stream` and never synchronise.
 *
 * dtype codes: 0 = float32, 1 = bfloat16 (storage type of activations / operand images; accumulation is
 * always float32, CTC recursions float64).
 *
 * Layout conventions: activations NHWC (channel contiguous), convolution weights KRSC, matrices row-major.
 * "16-byte vector" = 4 float32 or 8 bfloat16; channel counts and leading dimensions of MFMA operands must
 * be multiples of one vector.
 *
 * Each group cites the reference interface (file:line in Megvii-CSG/MegReader) it replaces.
 */
#ifndef MEGREADER_HIP_H
#define MEGREADER_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MR_ABI_VERSION 3   /* 3 (round 6): mr_tuning grew (nt_m32, nt_m32_opt, reserved[6]).  2 (round 5): mr_ctc_fwd gained log_probs_f64; mr_tuning.tn_defer; mr_tn_defer / mr_tn_flush;
                              the 26 mr_set_* setters of version 1 are gone (mr_tuning) */
#define MR_DTYPE_F32 0
#define MR_DTYPE_BF16 1

const char* mr_last_error(void);
int mr_abi_version(void);
/* Measurement only (bench.py's roofline pass): while mr_phase_timer(1) is in force, entry points made of several launches (the
 * DCNv2 forward / backward) bracket their parts with HIP events on the launch stream; mr_phase_read(id, &ms, &work) waits for them
 * and returns the number of records of phase id with their total milliseconds and total algorithmic work (bytes for the
 * bandwidth-bound passes, flops for the GEMMs).  ids: 0 dcn forward, 1 gcol GEMM, 2 coordinate pass, 3 CSR build (4 launches),
 * 4 input-gradient gather, 5 im2col, 6 weight-gradient GEMM.  Not capturable.  Host only. */
int mr_phase_timer(int on);
int mr_phase_read(int id, double* total_ms, double* total_work);
/* creates the per-device zero page eagerly and applies MEGREADER_TUNING (below); call before hipGraph capture */
int mr_init(void);

/* ---- Tuning / A-B state: ONE struct -----------------------------------------------------------------------------------------
 * Everything in the library that is process-wide and mutable, apart from the per-device zero page and the registered split-
 * reduction workspace, is this struct.  Rounds 1-3 exported 26 separate `mr_set_*` functions over 26 unsynchronised globals;
 * they are gone.  Every field selects between kernels that compute the SAME result (test / A-B / tuning hooks; the defaults are
 * what bench.py measures); the timing-only ablations that compute wrong results exist only in the -DMR_ABLATION tools build
 * (include/megreader_hip_ablation.h).
 *   mr_tuning_get / mr_tuning_set  read / replace the whole struct (writers serialised by a mutex, values range-checked);
 *   MEGREADER_TUNING="field=value,field=value"  is applied once by mr_init().
 * Threading: the compute entry points may be called from any thread on any stream; each of them reads the fields it needs with
 * single atomic loads, so a call that races with mr_tuning_set sees, per field, either the old or the new value -- never garbage.
 * Flipping fields while a training step is in flight is still only meaningful for A/B measurements. */
typedef struct mr_tuning {
  int nt_variant;    /* 2 (default) = direct-to-LDS NT kernels, 1 = register-staged NT kernel */
  int nt_deep;       /* pipeline depth of the 4-wave direct-to-LDS NT kernel: 1 (default) = launches with at most ~1.5 workgroups
                        per CU and >= 8 k-steps (bf16) run with 4 LDS stage buffers and 3 k-steps of LDS-DMA in flight across raw
                        barriers; 0 = always the 2-buffer loop; 2 = always the 4-buffer loop.  Same results bit for bit */
  int nt_big;        /* 8-wave big-tile NT kernels (256x256 / 272x256 / 288x128): 0 automatic, -1 never, 1..13 forced variants */
  int nt_p8;         /* 1 = phased-schedule 256x256 NT kernel (igemm_p8.h) for the big-tile launches, 0 (default) = plain */
  int nt_force_bm;   /* force one 4-wave NT tile shape: bm in {128, 96, 64} with */
  int nt_force_bn;   /* bn in {128, 64}; bm = 0 (default) = the cost model */
  int gemm_skinny;   /* 1 (default): M <= 32 GEMMs (attention decode loop) take the latency-optimised kernel of gemm_skinny.hip */
  int tn_big;        /* experimental wide-tile TN (weight-gradient) kernels: 1 = 256x256, 2 = 128x256, 0 / -1 = never (default) */
  int tn_buf;        /* TN operand staging: 1 (default) = raw buffer resources (out-of-range -> zeros), 0 = flat pointers */
  int tn_taps;       /* 1 (default): all-taps weight-gradient kernel for 3x3 / stride 1 / padding == dilation layers (tn_taps.hip)
                        through mr_conv2d_wgrad_tab; the row table has another format for it: rebuild tables after changing this */
  int tn_taps_group; /* split reduction of the all-taps kernel: 0 automatic, 1 atomics only, > 1 forced group size */
  int tn_group;      /* same for the 128x128 TN GEMM kernel (uses the same workspace) */
  int tn_fin;        /* 128x128 TN GEMM kernel: 2 = split partials -> slabs + finalize launch (measured slower), 0 (default) */
  int tn_taps_fin;   /* all-taps kernel: 0 (default) leaders' atomics; 1 group sums added by a finalize launch; 2 no in-launch
                        reduction, the finalize launch sums every split (1 and 2 measured within +-1 % of 0) */
  int tn_taps_w8;    /* 1: 8-wave workgroup variant of the all-taps kernel (one per CU, half the partial tiles) */
  int tn_model;      /* 1 (default): conv wgrad launches use the measured dense-GEMM split model */
  int tn_splits;     /* P-split override of the TN kernels, 0 (default) = the makespan model */
  int bn_fused;      /* 1 (default): BatchNorm finalisation folded into the apply passes (C % 64 == 0), 0 = separate launches */
  int lstm_persist;  /* 1 (default): persistent recurrence kernels (bf16, H = 256); 0 = one launch per step; 2 = persistent
                        without the XCD-colocating block map */
  int lstm_fwd_bn;   /* step kernels: hidden columns per workgroup forward (0 automatic, 32, 64) */
  int lstm_bwd_bn;   /* ... and backward (0 automatic, 16, 32 (default), 64) */
  int dcn_fused;     /* 1 (default): fused DCNv2 kernels (dcn_fused.hip) where the shape allows; 0 = the general kernels */
  int dcn_v1_bwd;    /* general DCN path: 1 (default) = round-1 backward kernels, 0 = the round-2 experiments */
  int bn_onepass;    /* 1 (default): BatchNorm backward of tensors that fit the registers of one resident grid runs as ONE launch
                        (reductions, a barrier among the workgroups of a 64-channel slab, dx from registers); 0 = always the
                        reduction launch + the apply launch */
  int skinny_depth;  /* k-chunks in flight per wave of the M <= 32 GEMM: 0 (default) = 4, or 8 when K needs more than one round
                        trip at 4 (K > 512 in bf16); 4 / 8 forced */
  int nt_big_min_k;  /* smallest K for which the automatic choice considers the 8-wave big-tile NT kernels (default 512) */
  int tn_taps_min_p; /* layers with fewer output pixels (N * H * W) than this (default 10000) keep the 128x128 TN GEMM kernel
                        instead of the all-taps kernel (round-4 in-step A/B: FPN-attention 9.68 -> 9.56 ms, CRNN at 32 crops per
                        GPU 1.47 -> 1.41, the P >= 16384 layers of Res50-PPM keep the all-taps kernel); rebuild row tables after
                        changing it (as for tn_taps) */
  int tn_defer;      /* 1 (default): mr_tn_defer(1) records the weight-gradient launches of the 128x128 TN GEMM kernel and
                        mr_tn_flush launches them several problems per launch (round 5); 0: mr_tn_defer is ignored, every
                        launch is immediate (A/B) */
  int pool_fixed;    /* 1 (default): max-pool forward with the window geometry as template constants (one packed store for the
                        arg-max codes) and the pooled-element-organised backward of the 2x2 / stride 2 pool (round 5); 0: the
                        round-4 kernels (bit-identical results; A/B) */
  int ctc_linear;    /* 1 (default): 1-D CTC recursions in the scaled linear domain (float64 products with exact power-of-two
                        rescaling, emission table in LDS) when the table fits; 0: the log-domain kernels (float64 log-sum-exp
                        per state and step).  alpha / beta buffers change meaning with it: the same value must be in force for
                        mr_ctc_fwd and the mr_ctc_bwd that consumes its buffers */
  int nt_wide8;      /* 8-wave workgroups (two per CU) instead of the 4-wave ones on the NT tile shapes named by this bit mask:
                        1 = 128x128, 2 = 128x64, 4 = 96x128, 8 = 64x128, 16 = 96x64, 32 = 64x64; 0 = never (round 4) */
  int nt_ksplit;     /* split reduction of the 4-wave NT kernels for launches of a few tiles with a long k-loop (round 5:
                        at most a quarter of the CUs busy, >= 16 k-steps): 1 (default) = as many splits as fill the chip once
                        with >= 4 k-steps each (<= 8), n > 1 = at most n, 0 = never.  Partial tiles meet in f32 slabs of the
                        split-reduction workspace (mr_set_tn_taps_workspace) and are added in split order: the same bits every
                        run; without a registered workspace the launch is unsplit.  Like the weight-gradient launches that use
                        that workspace, split launches must be stream-ordered with every other user of it: a process that
                        drives the library from two streams at once (training beside inference) sets this field to 0 */
  int nt_m32;        /* round 6: ping-pong NT kernel on v_mfma_f32_32x32x16_bf16 (nt32.hip: 8 waves as two groups one phase apart,
                        one always in its MFMA-only compute phase; output tile through LDS).  0 (default) = never: measured EQUAL
                        to the round-5 kernels on every CRNN layer (profiles/r06_nt32_*: the steady-state k-tile is 15 % shorter,
                        the fixed cost per launch 6 us longer; both kernels sit on the same launch + epilogue cost and the
                        chip's power-limited MFMA rate); 1 = wherever the automatic choice takes a 256-column big tile
                        (256x256 -> 256x256, 272x256 -> 288x256 as 1x8 waves); 2..5 = shape 256x256 / 288x256 / 256x128 / 128x256
                        for every eligible bf16 launch (sweeps) */
  int nt_m32_opt;    /* schedule variant of that kernel, 10 * PH + OPT (tools build -DMR_NT32_SWEEP only); 0 = the default */
  int dcn_gcol;      /* round 6: 1 (default) = bf16 DCNv2 backward through ONE materialised gcol = dy * W (tuned NT GEMM, bf16 output)
                        read by a coordinate-gradient pass and a CSR gather pass for the input gradient (dcn_fused.hip, C in
                        {64, 128, 256, 512}); 0 = the round-3 fused kernels (gcol in accumulators, gather-GEMM).  Changes the size
                        mr_dcn2_ws_bytes reports: workspaces must be sized under the value in force at the call */
  int dcn_col_fwd;   /* round 6: 1 (default) = on the dcn_gcol shapes the bf16 forward also goes through the column matrix (one sampling
                        pass into the caller's col_ws + the tuned NT GEMM) and leaves it there for the backward's weight gradient
                        (mr_dcn2_col_saved / mr_dcn2_bwd3); 0 = the fused forward kernel, the backward samples again */
  int decode_persist; /* round 6: 1 (default) = the attention-GRU decode loop runs its forward and its backward as ONE persistent launch
                        each (decode_persist.hip: bf16, H = 512, T <= 64, Ep <= 576; mr_decode_persist_ok); 0 = three launches per
                        step; 2 = persistent without the block map that puts a batch group's 32 workgroups on one XCD (A/B knob) */
  int reserved[3];   /* zero */
} mr_tuning;
int mr_tuning_get(mr_tuning* out);
int mr_tuning_defaults(mr_tuning* out);
int mr_tuning_set(const mr_tuning* in);   /* MR_ERR_ARG (mr_last_error names the field) when a value is out of range */

/* ---- GEMM family (replaces cuBLAS/cuDNN behind nn.Linear / nn.LSTM input projection:
 *      decoders/crnn.py:13-24; decoders/attention_decoder.py:187-231) ------------------------------------- */
/* C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]);  A,B,C of `dtype`, bias f32 (nullable), relu 0/1 */
int mr_gemm_nt(int dtype, const void* A, long long lda, const void* B, int ldb, void* C, long long ldc,
               const float* bias, int relu, int M, int N, int K, hipStream_t stream);
/* M <= 32 problems (the decode-loop GEMMs of the attention decoder) take a latency-optimised kernel (csrc/gemm_skinny.hip:
 * 16 output columns per workgroup, K split over the four waves, operands fetched as MFMA fragments straight from global
 * memory) unless mr_tuning.gemm_skinny = 0. */
/* workspace of the all-taps kernel's in-launch split reduction: device memory zeroed once by the caller (16 KB of
 * tickets + 147456 B per workgroup of the largest launch = 2 * CUs slabs); NULL / 0 withdraws it (f32 atomics only).
 * Launches that use it must be stream-ordered with each other. */
int mr_set_tn_taps_workspace(void* ws, long long bytes);
/* host only: 1 when mr_conv2d_wgrad_tab (bf16, non-NULL row table) would run the all-taps kernel for this geometry
 * under the current mr_tuning.tn_taps setting */
int mr_tn_taps_would_run(int N, int H, int W, int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw,
                         int ph, int pw, int dh, int dw, int Ho, int Wo);
/* tile (BM*1000+BN) the NT kernels pick for an M x N problem; host-only query used for profiling labels */
int mr_nt_tile_code(int M, int N);
/* same, including the big-tile policy (returns 256256 for the 8-wave 256x256 kernel); cg = channels of the gathered
 * conv operand, 0 for a dense GEMM */
int mr_nt_kernel_code(int dtype, int M, int N, int K, int cg);
/* C[NA,NB] (f32) += A[P,NA]^T * B[P,NB];  row_perm_h>0: gate-interleaved rows are written back in
 * PyTorch gate-major order (see lstm section).  colsum (nullable, f32[NA]) += sum_p A[p,:] (bias gradient,
 * fused into the same pass over A). */
int mr_gemm_tn(int dtype, const void* A, long long lda, const void* B, long long ldb, float* C, int ldc, int P,
               int NA, int NB, int row_perm_h, float* colsum, hipStream_t stream);
/* the same with a second destination for the column sums (colsum2, nullable; ignored when colsum is null): b_ih and b_hh of an
 * LSTM direction (decoders/crnn.py:13 nn.LSTM keeps both) receive the same gradient from one launch */
int mr_gemm_tn2(int dtype, const void* A, long long lda, const void* B, long long ldb, float* C, int ldc, int P,
                int NA, int NB, int row_perm_h, float* colsum, float* colsum2, hipStream_t stream);

/* Deferred, grouped weight-gradient launches (round 5).  The weight gradients of small layers -- the 1x1 / strided layers of a
 * ResNet at batch 32 (backbones/resnet.py:113-181), the LSTM / Linear layers of the CRNN head (decoders/crnn.py:8-24) -- are GEMMs
 * of 16..64 output tiles: alone, each has to cut its pixel loop into 8..32 splits to fill the chip.  Nothing later in the backward
 * pass reads them, so they can wait and run side by side.  mr_tn_defer(1): on the calling host thread, bf16 launches that would
 * take the 128x128 TN GEMM kernel (mr_gemm_tn; mr_conv2d_wgrad_tab off the all-taps kernel) are RECORDED (operand pointers and
 * geometry) instead of launched; mr_tn_defer(0) stops recording.  mr_tn_flush launches everything recorded, up to 12 problems per
 * launch, each with 1 / n-th of the splits.  The caller keeps every operand alive and unmodified until the flush, flushes on the
 * stream the operands were produced on, and must not read an output (or zero it) before the flush.  mr_tn_defer returns the
 * previous setting, mr_tn_pending the number of recorded problems (both host only).  mr_tuning.tn_defer = 0 makes mr_tn_defer a
 * no-op (A/B). */
int mr_tn_defer(int on);
int mr_tn_pending(void);
/* The queue is per DEVICE and process-wide (records pushed by one host thread are flushed by whichever thread calls mr_tn_flush
 * with that device current); mr_tn_discard drops the current device's records without launching them and switches recording off
 * for the calling thread -- for a caller whose backward pass raised and is about to release the recorded operands (round 6). */
int mr_tn_discard(void);
int mr_tn_flush(hipStream_t stream);
/* mr_tn_flush on a stream that runs CONCURRENTLY with the stream of the other weight-gradient launches (the reference leaves this
 * to cuDNN's wgrad behind nn.Conv2d / nn.LSTM backward on PyTorch's single stream, backbones/crnn.py:44-55, decoders/crnn.py:13):
 * nothing it launches touches the per-device split-reduction workspace (mr_set_tn_taps_workspace).  The caller orders the
 * stream after the producers of the recorded operands and joins it before anything reads the gradients. */
int mr_tn_flush_beside(hipStream_t stream);

/* ---- Convolution (replaces cuDNN conv fwd/dgrad/wgrad behind nn.Conv2d: backbones/crnn.py:44-55,
 *      backbones/resnet.py:39-256, backbones/ppm.py:11-44, decoders/ctc_decoder2d.py:16-27) --------------- */
int mr_conv2d_fwd(int dtype, const void* x, const void* w_krsc, const float* bias, void* y, int relu, int Nimg,
                  int H, int W, int Cin, int ldx, int Cout, int ldy, int R, int S, int sh, int sw, int ph, int pw,
                  int dh, int dw, int Ho, int Wo, hipStream_t stream);
/* conv + bias (+ ReLU) + max-pool as ONE launch (round 6): y_pool [Nimg, PHo, PWo, Cout] and the arg-max codes of mr_maxpool_fwd,
 * bit-identical to mr_conv2d_fwd followed by mr_maxpool_fwd; the full-resolution activation never reaches HBM.  bf16, stride-1
 * convolutions whose 8-wave tiles can be cut on window-row / image boundaries -- mr_conv2d_fwd_pool_ok (host only) says which;
 * replaces cuDNN conv + ReLU + MaxPool2d of backbones/crnn.py:14-33 (conv1: 2x2 / 2; conv3, conv5: 2x2, stride (2,1), pad (0,1)). */
int mr_conv2d_fwd_pool_ok(int dtype, int Nimg, int H, int W, int Cin, int ldx, int Cout, int R, int S, int sh, int sw, int ph, int pw,
                          int dh, int dw, int Ho, int Wo, int pkh, int pkw, int psh, int psw, int pph, int ppw);
int mr_conv2d_fwd_pool(int dtype, const void* x, const void* w_krsc, const float* bias, void* y_pool, unsigned char* idx, int relu,
                       int Nimg, int H, int W, int Cin, int ldx, int Cout, int R, int S, int sh, int sw, int ph, int pw, int dh,
                       int dw, int Ho, int Wo, int pkh, int pkw, int psh, int psw, int pph, int ppw, int PHo, int PWo,
                       hipStream_t stream);
/* mr_conv2d_fwd (no ReLU, ldy == Cout) that also leaves the BatchNorm batch statistics of y in bn_sums (layout and zeroing as
 * for mr_bn_stats) -- accumulated in the GEMM epilogue, so the BatchNorm that follows (reference nn.Sequential(conv, bn) at
 * backbones/resnet.py:39-56,113-181, crnn.py:48-52) skips its reduction pass over y: mr_bn_fwd_train(flags bit 3). */
int mr_conv2d_fwd_stats(int dtype, const void* x, const void* w_krsc, const float* bias, void* y, double* bn_sums, int Nimg,
                        int H, int W, int Cin, int ldx, int Cout, int R, int S, int sh, int sw, int ph, int pw, int dh,
                        int dw, int Ho, int Wo, hipStream_t stream);
/* w_crsk = weights transposed to [Cin][R][S][Cout] (mr_prep_conv_weight) */
int mr_conv2d_dgrad(int dtype, const void* dy, const void* w_crsk, void* dx, int Nimg, int H, int W, int Cin,
                    int lddx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw,
                    int Ho, int Wo, hipStream_t stream);
/* dx = dgrad(dy, w) + addend: the gradient of the residual branch of a ResNet block (backbones/resnet.py:152-181, `out +=
 * residual`) is added in the epilogue of the dgrad of the block's first convolution.  addend: NHWC with dx's channel count and
 * leading dimension, may alias dx, null = mr_conv2d_dgrad. */
int mr_conv2d_dgrad_add(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, int Nimg, int H, int W,
                        int Cin, int lddx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw,
                        int Ho, int Wo, hipStream_t stream);
/* mr_conv2d_dgrad_add whose epilogue also accumulates the BatchNorm-BACKWARD sums of the gradient it produces: dx is the gradient
 * of y = act(bn(x) [+ residual]) of a training-mode BatchNorm (the conv -> bn -> relu chains of backbones/resnet.py:113-181);
 * bn_sums (f64 [copies][2][Cin], the scratch of mr_bn_bwd, zeroed by the caller) receives sum g' and sum g' xhat per channel
 * (g' = dx where bn_y > 0; bn_y null = no fused ReLU), and mr_bn_bwd with flags bit 3 then skips its reduction pass.  *produced
 * (host int) = 1 when the sums were written, 0 when this geometry runs on a kernel without that epilogue (dx is complete either
 * way; the caller then calls mr_bn_bwd without bit 3). */
int mr_conv2d_dgrad_bnb(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const void* bn_x,
                        const void* bn_y, const float* bn_mean, const float* bn_rstd, double* bn_sums, int* produced,
                        int Nimg, int H, int W, int Cin, int lddx, int Cout, int lddy, int R, int S, int sh, int sw, int ph,
                        int pw, int dh, int dw, int Ho, int Wo, hipStream_t stream);
/* dw_krsc (f32 [Cout][R][S][Cin]) and dbias (nullable, f32[Cout]) are accumulated atomically: zero them first.  Cout may be
 * smaller than lddy and need not be a multiple of the vector width when the channels Cout..lddy-1 of dy are zero padding
 * (27-channel DCN offset convolutions stored with 32): exactly Cout rows of dw / entries of dbias are written. */
int mr_conv2d_wgrad(int dtype, const void* dy, const void* x, float* dw_krsc, float* dbias, int Nimg, int H, int W,
                    int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh,
                    int dw, int Ho, int Wo, hipStream_t stream);
/* mr_conv2d_wgrad with a caller-owned row table of the im2col gather: rowtab = N*Ho*Wo entries of 8 bytes
 * ({element offset of the pixel's window in x, validity bit per tap}); build != 0 fills it first, build == 0 trusts
 * it (a layer's geometry is constant: build once, reuse every step).  bf16 and R*S <= 32; otherwise, or with
 * rowtab == NULL, identical to mr_conv2d_wgrad.  build bit 1: the launch may run concurrently with other weight-gradient
 * launches (another stream): the shared split-reduction workspace (mr_set_tn_taps_workspace) is not used, f32 atomics only. */
int mr_conv2d_wgrad_tab(int dtype, const void* dy, const void* x, float* dw_krsc, float* dbias, int N, int H, int W,
                        int Cin, int ldx, int Cout, int lddy, int R, int S, int sh, int sw, int ph, int pw, int dh,
                        int dw, int Ho, int Wo, void* rowtab, int build, hipStream_t stream);

/* ---- DB detector loss (replaces the ~100 torch launches of decoders/seg_detector_loss.py:157-185 L1BalanceCELoss =
 *      balance_cross_entropy_loss.py:29-56 + l1_loss.py:5-11 + dice_loss.py:28-42) ---------------------------------------------
 * binary / thresh / thresh_binary / gt: f32 [N, H*W] (the [N,1,H,W] maps); mask / thresh_map / thresh_mask: f32 [N, H*W].  The
 * reference multiplies gt [N,1,H,W] with mask [N,H,W]: broadcasting makes positive / negative [N,N,H,W] tensors (gt of sample a,
 * mask and loss of sample b); that is restated as is.  negloss: f32 scratch [N*N*H*W]; ws: mr_db_loss_ws_bytes() bytes ZEROED by
 * the caller; out: f32 [16] = {loss, bce, l1, dice, state read by mr_db_loss_bwd ...}.  The sum of the nc largest negative losses
 * is an exact radix selection; elements tied at the threshold share the remaining count equally in the gradient. */
long long mr_db_loss_ws_bytes(void);
/* tail of the two DB heads (decoders/seg_detector.py:77-79,142-147): binary = sigmoid(xb), thresh = sigmoid(xt) -- float32 outputs
 * whatever `dtype` the logits have -- and thresh_binary = 1 / (1 + exp(-k (binary - thresh))); n elements each */
int mr_db_head_tail_fwd(int dtype, const void* xb, const void* xt, float* binary, float* thresh, float* tbinary, long long n,
                        float k, hipStream_t stream);
int mr_db_head_tail_bwd(int dtype, const float* binary, const float* thresh, const float* tbinary, const float* gb,
                        const float* gt, const float* gtb, void* dxb, void* dxt, long long n, float k, hipStream_t stream);
int mr_db_loss_fwd(const float* binary, const float* thresh, const float* tbinary, const float* gt, const float* mask,
                   const float* tmap, const float* tmask, float* negloss, void* ws, float* out, int N, long long HW,
                   float negative_ratio, float eps, float l1_scale, float bce_scale, hipStream_t stream);
int mr_db_loss_bwd(const float* binary, const float* thresh, const float* tbinary, const float* gt, const float* mask,
                   const float* tmap, const float* tmask, const float* out, const float* gloss, float* g_binary,
                   float* g_thresh, float* g_tbinary, int N, long long HW, float l1_scale, float bce_scale,
                   hipStream_t stream);

/* ---- layout / elementwise helpers ---------------------------------------------------------------------- */
int mr_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cpad,
                    hipStream_t stream);
int mr_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, int H, int W, int ld, hipStream_t stream);
int mr_cast(int src_dtype, const void* src, int dst_dtype, void* dst, long long n, hipStream_t stream);
int mr_relu_bwd(int dtype, const void* dy, const void* y, void* dx, long long n, hipStream_t stream);
int mr_add(int dtype, const void* a, const void* b, void* out, long long n, int relu, hipStream_t stream);
/* out[perm(c)] (f32) += sum_p x[p,c] */
int mr_colsum(int dtype, const void* x, float* out, int P, int C, long long ld, int perm_h, hipStream_t stream);
/* [A,B,C] -> [B,A,C] */
int mr_permute_021(int dtype, const void* src, void* dst, int A, int B, int C, hipStream_t stream);
/* fp32 master weights (logical [K][C][R][S], arbitrary strides) -> operand images of `dtype`:
 * dst_krsc [K][R][S][Cpad] (zero padded channels), dst_crsk [C][R][S][ldk] (transposed, for dgrad); either may be null */
int mr_prep_conv_weight(int dtype, const float* src, long long sk, long long sc, long long sr, long long ss,
                        void* dst_krsc, void* dst_crsk, int K, int C, int R, int S, int Cpad, int ldk,
                        hipStream_t stream);
/* src f32 [R][C] with row stride lds (elements) -> dst_n [R][ldn] and/or dst_t [C][ldt] of `dtype` (rows permuted to
 * the gate-interleaved order when perm_h > 0) */
int mr_prep_matrix(int dtype, const float* src, int lds, void* dst_n, int ldn, void* dst_t, int ldt, int R, int C,
                   int perm_h, hipStream_t stream);
int mr_prep_bias(const float* a, const float* b, float* dst, int R, int perm_h, hipStream_t stream);

/* All of the above in ONE launch.  The host keeps a table of prep jobs in device memory (it only changes when the
 * set of parameters changes) and re-runs it after every optimizer update -- the fp32 master -> compute-dtype operand
 * images of every layer are regenerated together instead of by ~3 small kernels per layer per step.
 *   MR_PREP_CONV:   mr_prep_conv_weight(src, s0..s3 = sk,sc,sr,ss, dst_a = krsc, dst_b = crsk, d0..d3 = K,C,R,S,
 *                   pad = Cpad, ld_b = ldk)
 *   MR_PREP_MATRIX: mr_prep_matrix(src, s0 = lds, dst_a = dst_n, pad = ldn, dst_b = dst_t, ld_b = ldt, d0,d1 = R,C,
 *                   perm_h)
 *   MR_PREP_BIAS:   mr_prep_bias(src, src2, dst_a (f32), d0 = R, perm_h)
 *   MR_PREP_STEM:   mr_stem_pack(src, s0..s3 = sk,sc,sr,ss, dst_a = wpack, d1 = Cin)   (one block)
 * Work unit = one 64x64 tile of the job's logical matrix (conv: K rows x R*S*Cpad columns; matrix: R x C; bias:
 * 4096 elements).  The jobs are laid end to end on the grid: job.block_start = index of its first tile (jobs
 * sorted by block_start, job 0 starts at 0); total_blocks = sum over jobs of
 *   conv: ceil(K/64)*ceil(R*S*Cpad/64), matrix: ceil(R/64)*ceil(C/64), bias: ceil(R/4096), stem: 1.
 * Transposed images need dst_b 16-byte aligned and ld_b a multiple of one 16-byte vector for the fast path. */
#define MR_PREP_CONV 0
#define MR_PREP_MATRIX 1
#define MR_PREP_BIAS 2
#define MR_PREP_STEM 3
typedef struct mr_prep_job {
  const float* src;
  const float* src2;
  void* dst_a;
  void* dst_b;
  long long s0, s1, s2, s3;
  int kind;
  int d0, d1, d2, d3;
  int pad, ld_b, perm_h;
  int block_start, reserved;
} mr_prep_job;
int mr_prep_batch(int dtype, const mr_prep_job* jobs_device, int njobs, long long total_blocks, float* tick,
                  hipStream_t stream);   /* tick (nullable): an optimizer's hyper block whose step counter this launch advances */
/* sizeof(mr_prep_job) as compiled into the library (host only): bindings verify their struct mirror against it */
int mr_sizeof_prep_job(void);

/* dst[i][0..n[i]) += src[i][0..n[i]) for count <= MR_MAX_SEGMENTS f32 segments in one launch (the pointer / length
 * arrays are HOST arrays, copied into the kernel arguments).  Used to fold several small gradient pieces into the
 * optimizer's flat gradient buffer (replaces one `grad += piece` kernel per parameter). */
#define MR_MAX_SEGMENTS 8
int mr_accumulate_multi(int count, float* const* dst, const float* const* src, const long long* n,
                        hipStream_t stream);
/* zero fill of count <= MR_MAX_SEGMENTS buffers (16-byte aligned, sizes multiples of 16 bytes) in one launch: zero_grad() of the
 * fused optimizers clears the flat gradient buffers and the pre-zeroed scratch arena together */
int mr_zero_multi(int count, void* const* dst, const long long* bytes, hipStream_t stream);

/* ---- optimizers (replaces torch.optim.Adam / SGD at training/optimizer_scheduler.py:17-22) -------------- */
/* hyper: device f32[8] = {lr, beta1 (SGD: momentum), beta2, eps, weight_decay, completed steps, gradient scale (0 = 1), -}.
 * The update kernels run step hyper[5] + 1 and only READ the counter; the launch behind them advances it: mr_prep_batch(tick =
 * hyper) -- the regeneration of the weight images that follows every update anyway -- or mr_opt_tick.  hipGraph-replay safe. */
int mr_opt_tick(float* hyper, hipStream_t stream);
int mr_adam_step(float* p, const float* g, float* m, float* v, long long n, float* hyper, hipStream_t stream);
int mr_sgd_step(float* p, const float* g, float* buf, long long n, float* hyper, hipStream_t stream);

/* ---- BatchNorm2d / MaxPool2d (backbones/crnn.py:17-31,49-52; backbones/resnet.py:26-30,199) ------------- */
/* doubles of reduction scratch mr_bn_fwd_train / mr_bn_bwd want for C channels: 8 accumulator copies of [2][C] (fewer
 * same-address atomics) followed by C doubles the backward uses -- the unfused path for 2*C f32 per-channel means, the one-pass
 * path (mr_tuning.bn_onepass) for one arrival counter per 64-channel slab, 512 bytes apart.  "Zeroed" below means the WHOLE
 * scratch. */
long long mr_bn_scratch_doubles(int C);
/* training-mode BN folds its finalize kernels into the apply passes (C % 64 == 0) unless mr_tuning.bn_fused = 0 */
int mr_bn_fwd_train(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* save_mean, float* save_rstd, double* sums, const void* residual,
                    int relu, long long P, int C, float eps, float momentum, long long* num_batches_tracked,
                    hipStream_t stream); /* num_batches_tracked (nullable): int64 step counter, incremented by one;
                                          relu: bit0 fused ReLU, bit2 `sums` is already zero (skip the memset), bit3 `sums`
                                          already HOLDS the statistics (mr_conv2d_fwd_stats / mr_bn_stats): no reduction pass */
/* the statistics pass on its own: sums (f64 [8][2][C] = the first 16*C doubles of the mr_bn_scratch_doubles(C) scratch, zeroed by
 * the caller) += per-channel sum / sum of squares of x [P][C] */
int mr_bn_stats(int dtype, const void* x, double* sums, long long P, int C, hipStream_t stream);
int mr_bn_fwd_eval(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                   const float* running_mean, const float* running_var, float* tmp_mean, float* tmp_rstd,
                   const void* residual, int relu, long long P, int C, float eps, hipStream_t stream);
int mr_bn_bwd(int dtype, const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
              const float* save_rstd, double* sums, void* dx, void* dres, float* dgamma, float* dbeta, int flags,
              long long P, int C, hipStream_t stream); /* flags: bit0 fused ReLU, bit1 accumulate into dgamma/dbeta, bit2 `sums`
                                          is already zero (all mr_bn_scratch_doubles(C) of it), bit3 `sums` already holds the two
                                          reductions (mr_conv2d_dgrad_bnb).  Tensors that fit one resident grid (C % 64 == 0,
                                          P * C * sizeof(T) < 2 GiB) take ONE launch with a barrier among the workgroups of a
                                          64-channel slab (bounded wait; a workgroup that times out poisons its outputs with NaN);
                                          the others the reduction launch + the apply launch. */
int mr_maxpool_fwd(int dtype, const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int kh,
                   int kw, int sh, int sw, int ph, int pw, int Ho, int Wo, hipStream_t stream);
/* relu_y (nullable): the pool's OUTPUT [N,Ho,Wo,C] when its input is the output of a ReLU -- fuses that ReLU's backward
 * mask (value at the arg-max position == pooled output, so the mask is read at pooled resolution) */
int mr_maxpool_bwd(int dtype, const void* dy, const unsigned char* idx, const void* relu_y, void* dx, int N, int H,
                   int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo,
                   hipStream_t stream);

/* ---- fused backbone stem: Conv2d(Cin -> 64, 3x3, s1, p1) + bias + ReLU + MaxPool2d(2,2) -----------------------
 * replaces cnn.conv0 / relu0 / pooling0 of reference backbones/crnn.py:17-19,48-55 in one pass each way.
 * x: fp32 NCHW contiguous [N,Cin,H,W] (Cin 1 or 3, even H and W); w: fp32 [64,Cin,3,3] with element strides
 * wsk,wsc,wsr,wss; y: pooled activation NHWC [N,H/2,W/2,64] of `dtype`; code: one byte per pooled element
 * (bits 0-1 = first-maximum position in the 2x2 window, bit 2 = maximum > 0).
 * mr_stem_bwd ACCUMULATES into dw (strides dsk..dss) and dbias (either may be null); workspace must hold
 * mr_stem_bwd_workspace(Cin) floats.  The layer's input gradient is not produced (the input is the image). */
long long mr_stem_bwd_workspace(int Cin);
/* wpack (bf16 [64][32], k = (dr*3+ds)*Cin + c, zero padded): the filter bank pre-packed for the bf16 MFMA kernel by
 * mr_stem_pack / an MR_PREP_STEM job.  When it is null (or dtype is f32, or W/2 is not a multiple of 16) mr_stem_fwd
 * runs the generic FMA kernel straight from w. */
int mr_stem_pack(const float* w, long long wsk, long long wsc, long long wsr, long long wss, void* wpack, int Cin,
                 hipStream_t stream);
int mr_stem_fwd(int dtype, const float* x, const float* w, long long wsk, long long wsc, long long wsr,
                long long wss, const void* wpack, const float* bias, void* y, unsigned char* code, int N, int Cin,
                int H, int W, hipStream_t stream);
int mr_stem_bwd(int dtype, const void* dy, const unsigned char* code, const float* x, float* workspace, float* dw,
                long long dsk, long long dsc, long long dsr, long long dss, float* dbias, int N, int Cin, int H,
                int W, hipStream_t stream);

/* ---- bidirectional LSTM recurrence (replaces cuDNN RNN behind nn.LSTM: decoders/crnn.py:13,21,91-93) ----- */
/* ws / ws_bytes: exchange workspace of the persistent one-launch recurrence (size from mr_lstm_ws_bytes; zeroed by
 * the call with a memset on `stream`; its last 256 bytes hold a status word: 0 = ok, 1 / 2 = a bounded spin of the
 * forward / backward kernel timed out).  Null / 0 selects one launch per time step.  dc is only used by the
 * per-step path (f32 [N, 2H] scratch). */
int mr_lstm_fwd(int dtype, const void* xproj, const void* whh, void* out, float* cbuf, void* gates, int T, int N,
                int H, void* ws, long long ws_bytes, hipStream_t stream);
int mr_lstm_bwd(int dtype, const void* dout, const void* whhT, const float* cbuf, void* gates, float* dc, int T,
                int N, int H, void* ws, long long ws_bytes, hipStream_t stream);
/* host only: workspace bytes the persistent recurrence wants (0 = not applicable: f32, H != 256, mr_tuning.lstm_persist = 0) */
long long mr_lstm_ws_bytes(int dtype, int T, int N, int H);
/* debug hook (host only): non-null -> the persistent backward also writes its reduced recurrent term [T,N,2H] f32 */
int mr_lstm_debug_buffer(float* p);
/* (column-tile width of the step kernels: mr_tuning.lstm_fwd_bn / lstm_bwd_bn; 0 = LDS-staged split-K body, else the
 * direct-fragment body) */

/* ---- 1-D CTC fused with log-softmax (replaces log_softmax + nn.CTCLoss: decoders/crnn.py:48,96-98) ------- */
int mr_ctc_fwd(int dtype, const void* logits, int ldl, const void* targets, int targets_i64,
               const void* input_lengths, const void* target_lengths, int lengths_i64, int T, int N, int C, int S,
               int blank, int zero_infinity, float* log_probs, double* alpha, double* beta, double* nll,
               double* loss, double* log_probs_f64, hipStream_t stream);
/* log_probs_f64 (nullable, f64 [T][N][C]): the same log-probabilities widened to float64 -- what the reference returns as
 * `pred` (decoders/crnn.py:96 `log_softmax(pred, dim=2).to(torch.float64)`), written by the same kernel.
 * mr_ctc_bwd writes the padding columns C .. ldg-1 of grad_logits as zeros.
 * alpha, beta: f64 [N][T][2S+1] -- scratch that travels from mr_ctc_fwd to mr_ctc_bwd: the log-domain forward / backward
 * variables (mr_tuning.ctc_linear = 0, or an emission table beyond 64 KB of LDS) or, by default, their scaled linear-domain
 * counterparts (alpha rescaled per step by exact powers of two; beta without the emission of its own step).  beta may be null in mr_ctc_fwd when no gradient is wanted (the beta recursion runs
 * concurrently with alpha on other wavefronts of the same workgroup, so storing it costs no extra latency and makes
 * the gradient kernel independent per (t, n) row). */
int mr_ctc_bwd(int dtype, const float* log_probs, const double* alpha, const double* beta, const double* nll,
               const void* targets, int targets_i64, const void* input_lengths, const void* target_lengths,
               int lengths_i64, const double* grad_out, int T, int N, int C, int S, int blank, int zero_infinity,
               void* grad_logits, int ldg, hipStream_t stream);

/* ---- 2D-CTC (replaces the CUDA extension ops/ctc_2d: csrc/ctc2d.h:7-43, cuda/ctc2d_cuda_kernel.cu) -------------
 * log_probs [T,H,N,C] contiguous (`dtype`), targets [N,S] i64, lengths [N] i64.  Reference pybind signatures:
 *   ctc2d_forward(log_probs, targets, input_lengths, target_lengths, BLANK, TINY) -> (nll[N], log_alpha[N,T,H,2S+1])
 *   ctc2d_backward(grad_out, log_probs, targets, input_lengths, target_lengths, nll, log_alpha, BLANK) -> grad
 * Here the caller allocates nll / alpha (f32), the beta scratch (f32, same shape as alpha) and grad (`dtype`). */
int mr_ctc2d_fwd(int dtype, const void* log_probs, const long long* targets, const long long* input_lengths,
                 const long long* target_lengths, int T, int H, int N, int C, int S, int blank, float* nll,
                 float* alpha, hipStream_t stream);
int mr_ctc2d_bwd(int dtype, const float* grad_out, const void* log_probs, const long long* targets,
                 const long long* input_lengths, const long long* target_lengths, const float* nll,
                 const float* alpha, float* beta, void* grad, int T, int H, int N, int C, int S, int blank,
                 hipStream_t stream);

/* ---- ResNet50-dilated / PPM / 2D-CTC head helpers (backbones/ppm.py:11-44, decoders/ctc_decoder2d.py:16-45) ----- */
int mr_adaptive_avgpool_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int OH, int OW,
                            hipStream_t stream);
int mr_adaptive_avgpool_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int OH, int OW,
                            hipStream_t stream);
/* Pyramid pooling (reference backbones/ppm.py:13-20,36-40): nscales (<= 8) adaptive average pools of ONE map per launch.  ys / dys:
 * host arrays of device pointers [N][oh[i]][ow[i]][C].  Forward outputs are bit-identical to mr_adaptive_avgpool_fwd's; backward
 * writes dx = sum_i adaptive_avgpool_bwd(dys[i]) (f32 accumulation, one rounding).  H * W * 128 bytes must fit 64 KB of LDS. */
int mr_adaptive_avgpool_multi_fwd(int dtype, const void* x, void* const* ys, const int* oh, const int* ow, int nscales, int N,
                                  int H, int W, int C, hipStream_t stream);
int mr_adaptive_avgpool_multi_bwd(int dtype, void* const* dys, const int* oh, const int* ow, int nscales, void* dx, int N, int H,
                                  int W, int C, hipStream_t stream);
/* bilinear resize, align_corners=False; writes channels [coff, coff+C) of rows with stride ldy; accumulate=1 adds */
int mr_bilinear_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int OH, int OW, int ldy, int coff,
                    int accumulate, hipStream_t stream);
int mr_bilinear_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int OH, int OW, int lddy,
                    int coff, hipStream_t stream);
/* nearest-neighbour upsampling by an integer factor s (decoders/seg_detector.py:22-43): y[n,oh,ow,coff+c] =
 * x[n,oh/s,ow/s,c] (+ add[n,oh,ow,c], nullable); backward sums the s x s block */
int mr_nearest_up_fwd(int dtype, const void* x, const void* add, void* y, int N, int H, int W, int C, int s, int ldy,
                      int coff, hipStream_t stream);
int mr_nearest_up_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int s, int lddy, int coff,
                      hipStream_t stream);
/* nn.ConvTranspose2d(cin, cout, 2, 2) of the DB heads (decoders/seg_detector.py:66-79) = a GEMM y2[p, co*4 + i*2 + j] = sum_ci
 * x[p, ci] W[ci, co, i, j] (mr_gemm_nt on the weight's own [Cin][Cout*4] matrix) + this depth-to-space pass with the bias:
 * y[n, 2h+i, 2w+j, co] = y2[p, (co, i, j)] + bias[co].  y2 [N*H*W][ld2 >= 4*C], y NHWC [N, 2H, 2W, C], bias f32 nullable. */
int mr_deconv2x2_d2s(int dtype, const void* y2, int ld2, const float* bias, void* y, int N, int H, int W, int C,
                     hipStream_t stream);
/* the inverse gather for backward: dy2[p, (co, i, j)] = dy[n, 2h+i, 2w+j, co], columns 4*C .. ld2-1 zero */
int mr_deconv2x2_s2d(int dtype, const void* dy, void* dy2, int ld2, int N, int H, int W, int C, hipStream_t stream);
int mr_copy_channels(int dtype, const void* src, int lds, int soff, void* dst, int ldd, int doff, long long P, int C,
                     hipStream_t stream);
/* dx [N, H, W, C] <- the gradient of the sub-sampling x[:, ::sh, ::sw, :]: dxs [N, Ho, Wo, C] at the sampled positions, zeros
 * elsewhere (one pass over dx).  With mr_conv2d_dgrad on the sub-sampled grid this is the data gradient of a strided 1x1
 * convolution without padding (the downsample branch of a ResNet stage, backbones/resnet.py:204-213). */
int mr_scatter_strided(int dtype, const void* dxs, void* dx, int N, int H, int W, int C, int sh, int sw, int Ho, int Wo,
                       hipStream_t stream);
int mr_scale_channels(int dtype, const void* x, const float* scale, void* y, int N, long long HW, int C,
                      hipStream_t stream);
/* lp[w,h,n,c] = log(max(softmax_h(mask)[n,h,w] * softmax_c(cls)[n,h,w,c], tiny)); also returns both softmaxes (f32) */
int mr_ctc2d_head_fwd(int dtype, const void* mask_logits, int lda, const void* cls_logits, int ldz, float* lp,
                      float* mask_prob, float* cls_prob, int N, int H, int W, int C, float tiny, hipStream_t stream);
int mr_ctc2d_head_bwd(int dtype, const float* grad_lp, const float* mask_prob, const float* cls_prob, void* dmask,
                      int ldda, void* dcls, int lddz, int N, int H, int W, int C, float tiny, hipStream_t stream);

/* ---- Modulated deformable conv v2 (replaces assets/ops/dcn: src/deform_conv_cuda.cpp:486-679 and
 *      src/deform_conv_cuda_kernel.cu:569-766; python API functions/deform_conv.py:108-177) --------------------------
 * Fused path (csrc/dcn_fused.hip; C % 64 == 0, Co % 64 == 0, kh*kw <= 9 -- every DCN layer of deformable_resnet50): the column
 *   matrix never exists.  forward = sample -> LDS -> MFMA; backward = offset / mask gradients from gcol tiles kept in the MFMA
 *   accumulators, input gradient as a CSR-inverted GATHER-GEMM (no f32 atomics), dW / dbias by a TN GEMM whose operand is
 *   sampled on the fly.
 * General path (any C % vector == 0): the three piecewise kernels below around mr_gemm_nt / mr_gemm_tn:
 *   forward  = mr_dcn2_im2col (whole batch) + mr_gemm_nt(col, w_krsc, bias)
 *   backward = mr_gemm_nt(dy, w^T) -> gcol; mr_dcn2_coord_grad; mr_dcn2_col2im; mr_dcn2_im2col + mr_gemm_tn (dW, dbias)
 * offset / mask are f32 and indexed per sample as FLAT [2*kh*kw][Ho][Wo] / [kh*kw][Ho][Wo] arrays from the sample's
 * base (per-sample strides off_bs / msk_bs), exactly like deform_conv_cuda_kernel.cu:599-612. */
int mr_dcn2_im2col(int dtype, const void* x, const float* offset, long long off_bs, const float* mask,
                   long long msk_bs, void* col, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                   int dil, int Ho, int Wo, hipStream_t stream);
int mr_dcn2_coord_grad(int dtype, const void* gcol, const void* x, const float* offset, long long off_bs,
                       const float* mask, long long msk_bs, float* doffset, float* dmask, int N, int H, int W,
                       int C, int kh, int kw, int stride, int pad, int dil, int Ho, int Wo, hipStream_t stream);
int mr_dcn2_col2im(int dtype, const void* gcol, const float* offset, long long off_bs, const float* mask,
                   long long msk_bs, float* dx, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                   int dil, int Ho, int Wo, hipStream_t stream);

/* single-call forms (replace modulated_deform_conv_cuda_forward / _backward,
 * assets/ops/dcn/src/deform_conv_cuda.cpp:486-679).  Caller owns all buffers incl. the workspace col_ws (the reference passes
 * `columns` the same way, functions/deform_conv.py:135) of mr_dcn2_ws_bytes(..., backward) bytes -- 0 for the fused forward
 * (col_ws may be NULL), the CSR of the scatter pattern for the fused backward, the column matrix [N*Ho*Wo, kh*kw*C] otherwise.
 * w_n [Co][kh*kw*C] / w_t [kh*kw*C][Co] are the mr_prep_matrix images of the KRSC weight; dx32 / doffset / dmask are ACCUMULATED
 * into and must arrive zeroed (the reference's Function allocates them with zeros_like, functions/deform_conv.py:150-154);
 * dw f32 [Co][kh*kw*C] and dbias f32 [Co] accumulated.  Any of the three output groups -- {doffset, dmask}, dx32, {dw, dbias} --
 * may be null: only the others are computed. */
long long mr_dcn2_ws_bytes(int dtype, int N, int H, int W, int C, int Co, int kh, int kw, int Ho, int Wo, int backward);
/* (mr_tuning.dcn_fused = 0: general kernels for every shape; mr_tuning.dcn_v1_bwd: backward variant of the general path) */
int mr_dcn2_fwd(int dtype, const void* x, const void* w_n, const float* bias, const float* offset, long long off_bs,
                const float* mask, long long msk_bs, void* y, void* col_ws, int N, int H, int W, int C, int Co, int kh,
                int kw, int stride, int pad, int dil, int Ho, int Wo, hipStream_t stream);
int mr_dcn2_bwd(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, long long off_bs,
                const float* mask, long long msk_bs, void* col_ws, float* dx32, float* doffset, float* dmask, float* dw,
                float* dbias, int N, int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int Ho,
                int Wo, hipStream_t stream);
/* host only: 1 when the fused kernels serve this shape (C and Co multiples of 64, at most 9 taps, mr_tuning.dcn_fused) */
int mr_dcn2_fused(int dtype, int H, int W, int C, int Co, int kh, int kw);
/* host only: 1 when mr_dcn2_bwd2 can write the input gradient in `dtype` directly (dx_t) for this shape */
int mr_dcn2_dx_direct(int dtype, int N, int H, int W, int C, int Co, int kh, int kw);
/* mr_dcn2_bwd with: dx_t (nullable, INSTEAD of dx32) = the input gradient in `dtype`, overwritten (no zero fill, no conversion
 * pass; only where mr_dcn2_dx_direct says 1); flags bit 0 = col_ws was zeroed once and has been used by this function only
 * since (its counters clean themselves): the memset node in front of the CSR build is dropped. */
int mr_dcn2_bwd2(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, long long off_bs,
                 const float* mask, long long msk_bs, void* col_ws, float* dx32, void* dx_t, int flags, float* doffset,
                 float* dmask, float* dw, float* dbias, int N, int H, int W, int C, int Co, int kh, int kw, int stride, int pad,
                 int dil, int Ho, int Wo, hipStream_t stream);
/* round 6: mr_dcn2_col_saved (host only) = 1 when mr_dcn2_fwd leaves the sampled column matrix [N*Ho*Wo, kh*kw*C] in its col_ws
 * (bf16, mr_tuning.dcn_col_fwd); mr_dcn2_bwd3 = mr_dcn2_bwd2 + col_saved (nullable): that buffer, kept alive by the caller --
 * the weight gradient reads it instead of sampling x again. */
int mr_dcn2_col_saved(int dtype, int H, int W, int C, int Co, int kh, int kw);
int mr_dcn2_bwd3(int dtype, const void* dy, const void* x, const void* w_t, const float* offset, long long off_bs,
                 const float* mask, long long msk_bs, void* col_ws, float* dx32, void* dx_t, int flags, float* doffset,
                 float* dmask, float* dw, float* dbias, const void* col_saved, int N, int H, int W, int C, int Co, int kh, int kw,
                 int stride, int pad, int dil, int Ho, int Wo, hipStream_t stream);
/* Packed offset/mask operand of the deformable ResNet blocks (reference backbones/resnet.py:125-142: offset_mask =
 * conv2_offset(x); conv2(x, offset_mask[:, :18], offset_mask[:, -9:].sigmoid())).  raw = the offset conv's output, NHWC
 * [N][HW][ld] in `dtype` (channels n_offset + n_mask <= ld).  mr_dcn_unpack writes the flat f32 NCHW offset [N][n_offset][HW]
 * and mask = sigmoid(logits) [N][n_mask][HW] that mr_dcn2_fwd / _bwd read; mr_dcn_pack_grad folds their gradients back into
 * d raw (dmask * m * (1 - m) for the mask logits, zeros in the padding channels).  One launch each: they replace the slice,
 * cast, contiguous, sigmoid, sigmoid_backward, slice_backward and add launches of the unfused autograd graph. */
int mr_dcn_unpack(int dtype, const void* raw, int ld, float* offset, float* mask, int N, int HW, int n_offset, int n_mask,
                  hipStream_t stream);
int mr_dcn_pack_grad(int dtype, const float* doffset, const float* dmask, const float* mask, void* graw, int ld, int N,
                     int HW, int n_offset, int n_mask, hipStream_t stream);

/* ---- Deformable PS-RoI pooling (assets/ops/dcn/src/deform_pool_cuda.cpp:30-77 deform_psroi_pooling_cuda_forward /
 * _backward; kernels deform_pool_cuda_kernel.cu:52-143, 146-268).  data [B][C][H][W] f32, rois [R][5] (batch index, x1,
 * y1, x2, y2), trans [R][channels_trans][part][part] (ignored when no_trans), out / top_count / out_grad
 * [R][output_dim][P][P]; all caller-allocated.  bwd ACCUMULATES into data_diff / trans_diff (caller zero-fills, as
 * functions/deform_pool.py:57-59 does). */
int mr_deform_psroi_fwd(const float* data, const float* rois, const float* trans, float* out, float* top_count, int B,
                        int C, int H, int W, int R, int channels_trans, int no_trans, float spatial_scale,
                        int output_dim, int group_size, int pooled_size, int part_size, int sample_per_part,
                        float trans_std, hipStream_t stream);
int mr_deform_psroi_bwd(const float* out_grad, const float* data, const float* rois, const float* trans,
                        const float* top_count, float* data_diff, float* trans_diff, int B, int C, int H, int W, int R,
                        int channels_trans, int no_trans, float spatial_scale, int output_dim, int group_size,
                        int pooled_size, int part_size, int sample_per_part, float trans_std, hipStream_t stream);

/* ---- DB detector post-processing (structure/representers/seg_detector_representer.py:63-168): the per-pixel stages.
 * mr_db_components: prob f32 [N][H][W] thresholded (> thresh) -> labels i32 [N][H][W] (root pixel index of the
 * 8-connected component, -1 background) and the run end points of every component, points int4 [cap] = (n, root, x, y),
 * *count = their number (zero on entry; may exceed cap).  mr_db_box_scores: boxes f32 [B][9] = image index + 4 vertices
 * in order; out f32 [B][2] = (sum of prob, pixel count) over the pixels inside or on the border of each box. */
int mr_db_components(const float* prob, float thresh, int* labels, void* points, int* count, int cap, int N, int H, int W,
                     hipStream_t stream);
int mr_db_box_scores(const float* prob, const float* boxes, float* out, int B, int N, int H, int W, hipStream_t stream);

/* ---- Attention-GRU decoder step kernels (decoders/attention_decoder.py:146-231; the GEMMs use mr_gemm_nt/tn) ----- */
int mr_attn_step_fwd(int dtype, const void* hproj, const void* eproj, const float* v, const void* enc, float* weights,
                     void* context, int N, int T, int Hd, int Ep, hipStream_t stream);
/* deproj / denc (f32, shared by all steps of a sequence) are accumulated (+=); dv (f32) is accumulated atomically */
int mr_attn_step_bwd(int dtype, const void* dcontext, const float* dweights, const void* hproj, const void* eproj,
                     const float* v, const void* enc, const float* weights, void* dhproj, float* deproj, float* dv,
                     float* denc, int N, int T, int Hd, int Ep, hipStream_t stream);
int mr_gru_gates_fwd(int dtype, const void* gi_a, const void* gi_b, const void* gh, const void* h, void* hnew,
                     float* save, int N, int H, hipStream_t stream);
int mr_gru_gates_bwd(int dtype, const void* dhnew, const float* save, const void* gh, const void* h, void* dgi,
                     void* dgh, void* dh, int N, int H, hipStream_t stream);
int mr_nll_step_fwd(int dtype, const void* logits, int ldl, const long long* target, long long tstride,
                    const float* mask, float* lp, float* loss, long long* argmax, int N, int C, int accumulate,
                    int softmax_out, hipStream_t stream);
/* training decode loop: mr_nll_step_fwd (log-probs) that also writes the word index fed to the NEXT step,
 * feed_idx[n] = *feed_flag ? target[n] : arg-max[n].  feed_flag points to ONE int in device memory: the teacher-forcing coin of
 * this step (decoders/attention_decoder.py:107-110) -- a device value, so a replayed hipGraph follows the coins of the current
 * step, not those of the step that was captured; null = arg-max feedback. */
int mr_nll_step_feed_fwd(int dtype, const void* logits, int ldl, const long long* target, long long tstride,
                         const float* mask, float* lp, float* loss, long long* argmax, const int* feed_flag,
                         long long* feed_idx, int N, int C, int accumulate, hipStream_t stream);
int mr_nll_step_bwd(int dtype, const float* gloss, const float* lp, const long long* target, long long tstride,
                    const float* mask, void* dlogits, int ldd, int N, int C, hipStream_t stream);
/* word embedding of the attention decoder (decoders/attention_decoder.py:187-193): out[n, :D] = table[idx[n]] cast to
 * `dtype`, columns D..ldo-1 zeroed; backward scatter-adds into the f32 table gradient */
int mr_embed_rows_fwd(int dtype, const long long* idx, const float* table, void* out, int N, int V, int D, int ldo,
                      hipStream_t stream);
int mr_embed_rows_bwd(int dtype, const long long* idx, const void* dout, float* dtable, int N, int V, int D, int ldo,
                      hipStream_t stream);
/* round-3 decode loop (one autograd Function for the 32 steps, megreader_amd/decoders/attention_decoder.py): the same math as
 * the step kernels above with strided operands -- hproj / gh are column slices (leading dimensions ldh / ldgh) of ONE stacked
 * GEMM output per step, gi_a rows are gathered from a [classes, 3H] table through idx (nullable), the three consumers of h'
 * are summed inside mr_gru_bwd2 (dh_a / dh_b / dh_c nullable), mr_attn_bwd2 runs on a (N, Hd/64) grid and leaves the
 * encoder-side gradient to ONE mr_attn_denc launch after the loop (denc[n,t,c] = sum_s weights[s,n,t] * dcontext[s,n,c]),
 * mr_rows_scatter_add is the gradient of the table gather (f32 atomics into dtable [V][D]). */
int mr_attn_fwd2(int dtype, const void* hproj, long long ldh, const void* eproj, const float* v, const void* enc,
                 float* weights, void* context, int N, int Tn, int Hd, int Ep, hipStream_t stream);
int mr_attn_bwd2(int dtype, const void* dcontext, const float* dweights, long long ldw, const void* hproj, long long ldh,
                 const void* eproj, const float* v, const void* enc, const float* weights, void* dhproj, long long lddh,
                 float* deproj, float* dv, int N, int Tn, int Hd, int Ep, hipStream_t stream);
int mr_attn_denc(int dtype, const float* weights, const void* dcontext, void* denc, int S, int N, int Tn, int Ep,
                 hipStream_t stream);
int mr_gru_fwd2(int dtype, const void* gi_a, long long lda, const long long* idx, const void* gi_b, const void* gh,
                long long ldgh, const void* h, void* hnew, float* save, int N, int H, hipStream_t stream);
int mr_gru_bwd2(int dtype, const void* dh_a, const void* dh_b, const void* dh_c, const float* save, const void* gh,
                long long ldgh, const void* h, void* dgi, void* dgh, long long lddgh, void* dh_prev, int N, int H,
                hipStream_t stream);
int mr_rows_scatter_add(int dtype, const long long* idx, const void* rows, long long ldr, float* dtable, int R, int V,
                        int D, hipStream_t stream);

/* Persistent forward of the attention-GRU decode loop (reference decoders/attention_decoder.py:84-118 around
 * AttentionRNNCell :146-231): all S steps in ONE launch of ceil(N/16) x 32 co-resident workgroups; the stacked hidden projection
 * [4H][H] (attention hidden half + W_hh), the context columns of W_ih and the encoder rows stay in registers / LDS for the whole
 * sequence and the three per-step hand-offs (partial energies -> owners, contexts, h') go through tagged 8-byte granules
 * (decode_persist.hip).  bf16 only, H = 512.  cat_w [4H][H], cat_b [4H] f32 (nullable), ic_w [3H][ldic], G = word table
 * [classes][ldG >= 3H] gathered by idx [S][N] (the word fed to each step), eproj [N][T][H], enc [N][T][Ep], v [H] f32.
 * Writes what the per-step path (mr_gemm_nt + mr_attn_fwd2 + mr_gemm_gru_fwd) saves for the backward: H_all [S+1][N][H]
 * (H_all[0] = the initial state, read), HC_all [S][N][4H], W_att [S][N][T] f32, CTX_all [S][N][Ep], SAVE_all [S][N][3H] f32.
 * flags null: every step is fed the word idx[s] (teacher forcing).  flags [S] int32 on the device (the per-step coins of
 * attention_decoder.py:107-110): where flags[s] == 0, step s + 1 is fed the arg-max of step s's output layer out_w [C][H] / out_b
 * (f32, nullable), C <= 256, first index on ties as mr_out_nll_fwd -- and the kernel WRITES that word to idx[s + 1].
 * ws: exchange workspace of mr_decode_persist_ws_bytes(N) bytes, zeroed by the call (ws_bytes > 0) or by the caller (pass the
 * size NEGATIVE).  A hand-off that times out records a code in the status word behind the workspace and poisons h' with NaN. */
int mr_decode_persist_ok(int dtype, int N, int T, int H, int Ep);   /* host only; 0 also when mr_tuning.decode_persist = 0 or the
                                                                        current device has fewer CUs than the launch has workgroups */
long long mr_decode_persist_ws_bytes(int N);                        /* host only */
int mr_decode_persist_fwd(const void* cat_w, const float* cat_b, const void* ic_w, long long ldic, const void* G, long long ldG,
                          long long* idx, const int* flags, const void* out_w, const float* out_b, int C, const void* eproj,
                          const void* enc, const float* v, void* H_all, void* HC_all, float* W_att, void* CTX_all,
                          float* SAVE_all, void* ws, long long ws_bytes, int S, int N, int T, int Ep, hipStream_t stream);
/* ... and its backward: all S steps in reverse in one launch -- what mr_gemm_gru_bwd / mr_gru_bwd2 + mr_gemm_nt + mr_attn_bwd2 compute
 * per step (N <= 32).  cat_wt [H][4H] and ic_wt [Ep][ldict >= 3H] are the TRANSPOSED weight images (row = output of the backward
 * GEMM, K = stacked column / gate unit); DHO_all [S][N][H] = gradient of every h' from the output layer; ga (nullable) = gradient
 * of the attention weights, element (n, s, t) at ga[n * ldga + s * T + t].  Writes DGI_all [S][N][3H], DHC_all [S][N][4H],
 * DCTX_all [S][N][Ep], deproj [N][T][H] (f32, plain stores: the per-step path accumulates into it), ADDS into dv [H] (f32) and,
 * when denc is not null, writes denc [N][T][Ep] = sum_s W_att[s] dctx[s] (what mr_attn_denc computes after the per-step loop). */
int mr_decode_persist_bwd_ok(int dtype, int N, int T, int H, int Ep);   /* host only */
long long mr_decode_persist_bwd_ws_bytes(int N);                        /* host only */
int mr_decode_persist_bwd(const void* cat_wt, const void* ic_wt, long long ldict, const void* eproj, const void* enc,
                          const float* v, const void* H_all, const void* HC_all, const float* W_att, const float* SAVE_all,
                          const void* DHO_all, const float* ga, long long ldga, void* DGI_all, void* DHC_all, void* DCTX_all,
                          float* deproj, float* dv, void* denc, void* ws, long long ws_bytes, int S, int N, int T, int Ep,
                          hipStream_t stream);

/* ---- Round-4 decode-step fusions (csrc/gemm_skinny.hip): the element-wise GRU kernels in the epilogue of the M <= 32 GEMM next
 *      to them, the output layer + log-softmax + NLL + arg-max feedback as one kernel.  Same reference lines as above
 *      (attention_decoder.py:92-115,195-231); forward chain per step 6 -> 4 launches, backward 4 -> 3. ---------------------- */
/* mr_gemm_nt(ctx, w_ic) + mr_gru_fwd2 in one launch: gi_c = ctx [M, K] * w_ic [3H, K]^T stays in f32 and feeds the GRU cell of
 * the workgroup's 16 hidden units.  G: word table rows gathered by idx (null: row m), leading dimension ldG >= 3H; gh: the
 * hidden projection with its bias, leading dimension ldgh; h / hnew [M, H]; save f32 [M, 3H] (r, z, n).  M <= 32. */
int mr_gemm_gru_fwd(int dtype, const void* ctx, long long ldc, const void* w_ic, long long ldw, const void* G, long long ldG,
                    const long long* idx, const void* gh, long long ldgh, const void* h, void* hnew, float* save, int M, int H,
                    int K, hipStream_t stream);
/* mr_gemm_nt(dhc, w_t) + mr_gru_bwd2 in one launch: dh_a = dhc [M, K] * w_t [H, K]^T is never stored; dh_b / dh_c nullable;
 * dh_prev may alias dh_b.  Outputs as mr_gru_bwd2. */
int mr_gemm_gru_bwd(int dtype, const void* dhc, long long lda, const void* w_t, long long ldw, const void* dh_b,
                    const void* dh_c, const float* save, const void* gh, long long ldgh, const void* h, void* dgi, void* dgh,
                    long long lddgh, void* dh_prev, int M, int H, int K, hipStream_t stream);
/* mr_gemm_nt(h, W, bias) + mr_nll_step_feed_fwd in one launch (one workgroup per sample; the logits stay in f32 and are not
 * stored): lp f32 [N, C] = log_softmax, loss[n] (+)= -lp[n, target] * mask[n], argmax, feed_idx[n] = *feed_flag ? target[n] :
 * argmax[n] (feed_flag / feed_idx / loss / argmax / mask / bias nullable).  C <= 256. */
int mr_out_nll_fwd(int dtype, const void* h, long long ldh, const void* W, long long ldw, const float* bias,
                   const long long* target, long long tstride, const float* mask, float* lp, float* loss, long long* argmax,
                   const int* feed_flag, long long* feed_idx, int N, int C, int K, int accumulate, hipStream_t stream);

/* eval head: softmax over classes of logits [T,N,C] -> f32 [N,C,1,T] (decoders/crnn.py:101-104) */
int mr_softmax_nc1t(int dtype, const void* logits, int ldl, float* out, int T, int N, int C, hipStream_t stream);

/* ---- evaluation decode + metrics on the GPU ------------------------------------------------
 * mr_ctc_greedy_decode: structure/representers/ctc_representer.py:20-34.  pred[n,c,t] at pred + n*sn + c*sc + t*st
 *   (element strides; dtype 0 = f32, 1 = bf16, 2 = f64).  argmax over c (first index on ties); a symbol is skipped if
 *   it equals `previous` or is `unknown` (an unknown does not update `previous`); emitted if != blank.
 *   out: i32 [N, T] blank padded; out_len (nullable): i32 [N] number of emitted symbols.
 * mr_ctc2d_greedy_decode: structure/representers/ctc_representer2d.py:27-51.  classify[n,c,h,w], mask[n,0,h,w] f32
 *   with element strides; per column h* = argmax_h max_c(classify*mask), c* = argmax_c at h*; same collapse.
 * mr_seq_measure: structure/measurers/sequence_recognition_measurer.py:66-72,101-112 on id sequences (blank / unknown
 *   dropped as concern/charsets.py:60-62 does; fold: nullable id -> canonical id table for `.upper()`):
 *   acc[n] = sequences equal; ed[n] = Levenshtein distance (-1 if a sequence exceeds 63 symbols);
 *   score[n] (f64) = 0 if len(label) == 0 else 1 - min(len, ed) / len. */
int mr_ctc_greedy_decode(int dtype, const void* pred, long long sn, long long sc, long long st, int N, int C, int T,
                         int blank, int unknown, int* out, int* out_len, hipStream_t stream);
int mr_ctc2d_greedy_decode(const float* classify, long long cn, long long cc, long long ch, long long cw,
                           const float* mask, long long mn, long long mh, long long mw, int N, int C, int H, int W,
                           int blank, int unknown, int* out, int* out_len, hipStream_t stream);
int mr_seq_measure(const int* labels, int S, const int* preds, int S2, int N, int blank, int unknown, const int* fold,
                   int* acc, int* ed, int* label_len, double* score, hipStream_t stream);

/* ---- input pipeline on the GPU ----------------------------------------------------------------
 * mr_resize_normalize: data/processes/resize_image.py:29-38,48-53 (cv2.resize of the float32 image, INTER_LINEAR;
 *   modes "resize" and "pad") fused with data/processes/normalize_image.py:8-17 (-= RGB_MEAN in double, /= 255 in f32,
 *   HWC -> CHW).  src: packed uint8 HWC (3 channel) images; desc: device array of
 *   struct { long long offset; int h, w, pitch, dst_w; double scale_x, scale_y; } (mr_sizeof_img_desc() bytes
 *   each; scale = 1. / ((double)dst / src) as cv2 computes it); dst: f32 [N,3,H,W].
 * mr_encode_labels: concern/charsets.py:37-58 + data/processes/make_recognition_label.py:11-24.  codepoints: i32
 *   UTF-32 text of all strings back to back, offsets: i64 [N+1]; table_cp (sorted) / table_id: the charset's
 *   codepoint -> id map (case folding baked in by the host); label: i32 [N, max_size] zero padded,
 *   length: i32 [N] = min(len, max_size). */
int mr_sizeof_img_desc(void);
int mr_resize_normalize(const unsigned char* src, const void* desc, int N, int H, int W, double mean0, double mean1,
                        double mean2, float* dst, hipStream_t stream);
int mr_encode_labels(const int* codepoints, const long long* offsets, int N, int max_size, const int* table_cp,
                     const int* table_id, int ntab, int unknown, int* label, int* length, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MEGREADER_HIP_H */
