/*
 * mau_hip.h -- C ABI of libmau_hip.so: the MI355X (gfx950) hot path of the
 * Metadata-Augmented U-Net (forward + backward of src/model.py as driven by
 * src/train.py:243-256 in the reference repository).
 *
 * Boundary rules:
 *   - plain pointers and sizes only, no torch / C++ types;
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (the caller passes its current stream);
 *   - no hidden allocation, no retained pointers: every workspace is an argument;
 *   - re-entrant per stream; safe to capture into a hipGraph (no sync, no malloc);
 *   - return value: 0 = ok, non-zero = error (message via mau_last_error()).
 *
 * Internal activation layout ("NHWC-ld"): element (n, y, x, c) of a tensor lives at
 *   base + ((n*H + y)*W + x)*ld + c,   ld % 8 == 0,  ld >= C,
 * and channels [C, roundup(C,8)) are always zero.  `dtype` selects the arithmetic type of
 * activations and packed weights: MAU_F32 (parity mode, exact fp32 MFMA),
 * MAU_BF16 (throughput mode, bf16 MFMA with fp32 accumulation) or MAU_F16 (the same kernels on
 * fp16 operands: 3 more mantissa bits, no loss scaling -- meant for inference).  Parameters,
 * gradients of parameters, BatchNorm statistics and the model's external
 * inputs/outputs are always fp32 in the reference's own layouts (NCHW, OIHW).
 *
 * Each entry point names the reference code it replaces (file:line in the
 * reference repo).
 */
#ifndef MAU_HIP_H
#define MAU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAU_F32 0
#define MAU_BF16 1
#define MAU_F16 2     /* fp16 activations / packed weights, v_mfma_f32_32x32x16_f16, fp32 accumulation (BASELINE configs[4]) */

#define MAU_OK 0
#define MAU_ERR_ARG 1      /* bad argument (shape, alignment, dtype)      */
#define MAU_ERR_HIP 2      /* a HIP runtime call / kernel launch failed   */
#define MAU_ERR_DEVICE 3   /* no gfx950 device                             */

typedef void* mau_stream_t;

/* ---- library ---------------------------------------------------------- */
#define MAU_ABI_VERSION 5  /* 2: single-launch reductions (tickets), multi-tensor weight pack, fused BatchNorm passes; 3: first-layer kernels;
                            * 4: mau_set_cu_budget (an experiment, measured a loss); 5: mau_set_cu_budget removed, mau_conv3x3_variant reports K groups,
                            *    mau_conv3x3_fwd_pool */
int mau_abi_version(void);
const char* mau_last_error(void);
/* 0 when the current HIP device is a gfx950 (MI355X); MAU_ERR_DEVICE otherwise. */
int mau_device_check(void);

/* ---- layout at the module boundary ------------------------------------ */
/* maps (B,C,H,W) fp32 as handed over by collate_fn (src/dataset.py:99-106) -> NHWC-ld. */
int mau_nchw_to_nhwc(const float* src, void* dst, int dtype, int N, int C, int H, int W, int ld,
                     mau_stream_t stream);
/* inverse (tests, debugging, gradient w.r.t. maps). */
int mau_nhwc_to_nchw(const void* src, float* dst, int dtype, int N, int C, int H, int W, int ld,
                     mau_stream_t stream);

/* Input pipeline (reference src/dataset.py:53-72 hands over 23 fp32 planes per tile, 18 of them the one-hot
 * expansion of two class maps, src/data/processing_10m/process.py:176-181 + normalization.py:96-100): the class maps
 * travel as uint8 (N,H,W) and the one-hot channels are generated on the device.
 * out (N,H,W,ldo) = [onehot(cls_a) num_classes | cont (N,ncont,H,W) fp32 | onehot(cls_b) num_classes | zero pad];
 * flip (N) uint8 or NULL: nonzero mirrors that tile along W (RandomFlip, src/dataset.py:134-141). ncont <= 8. */
int mau_pack_tile_onehot(const unsigned char* cls_a, const unsigned char* cls_b, const float* cont,
                         const unsigned char* flip, void* out, int ldo, int dtype, int N, int H, int W,
                         int num_classes, int ncont, mau_stream_t stream);
/* dst[n,c,y,x] = src[n,c,y, flip[n] ? W-1-x : x]: the target half of RandomFlip (src/dataset.py:139). */
int mau_flip_rows(const float* src, float* dst, const unsigned char* flip, int N, int C, int H, int W,
                  mau_stream_t stream);

/* ---- 3x3 convolution, stride 1, pad 1 (nn.Conv2d(.,.,3,padding=1), src/model.py:12,14) ---- */
/* Channel-chunk size of the packed weights for `dtype` (K-chunk of the implicit GEMM). */
int mau_conv3x3_kc(int dtype);
/* Elements (of `dtype`) of a packed weight buffer with `nout` output and `nin` input channels. */
size_t mau_conv3x3_packed_elems(int dtype, int nout, int nin);
/* OIHW fp32 weights (Cout,Cin,3,3) -> forward pack `wf` [Cin/KC][9][Cout64][KC] and/or data-gradient
 * pack `wd` [Cout/KC][9][Cin64][KC] (taps rotated by 180 degrees); either may be NULL, not both.  The packs are opaque
 * inputs of mau_conv3x3_fwd for the same dtype (MAU_BF16 permutes the rows inside every 64-row block). */
int mau_conv3x3_pack_weights(const float* w_oihw, void* wf, void* wd, int dtype, int Cout, int Cin,
                             mau_stream_t stream);
/* Every convolution of a network in ONE launch (an optimizer step changes all of them: 18 launches -> 1 in the U-Net).
 * The caller keeps a table of mau_conv3x3_pack_desc_bytes()-sized rows: it fills row `index` on the HOST with
 * mau_conv3x3_pack_desc_fill (arguments as mau_conv3x3_pack_weights; tile0 = *next_tile_host of the previous row, 0 for
 * row 0; device pointers are only recorded), copies the table to the device once, and calls
 * mau_conv3x3_pack_weights_multi(device table, rows, total tiles = the last *next_tile_host, dtype, stream). */
size_t mau_conv3x3_pack_desc_bytes(void);
int mau_conv3x3_pack_desc_fill(void* descs_host, int index, const float* w_oihw, void* wf, void* wd, int dtype, int Cout,
                               int Cin, int tile0, int* next_tile_host);
int mau_conv3x3_pack_weights_multi(const void* descs, int n, int total_tiles, int dtype, mau_stream_t stream);
/* Rows of the BatchNorm partial-statistics slab the forward launch writes for an N x H x W batch with Cout output
 * channels: one per 8x16-pixel tile for MAU_F32; one per (workgroup tile of 16x16 or 32x16 pixels, wave row) for
 * MAU_BF16 -- the tile height is chosen per layer from how well its work items fill the 256 CUs. */
int mau_conv3x3_num_pixel_tiles(int dtype, int N, int H, int W, int Cout);
/* Inference form of an ENCODER block's second convolution (reference src/model.py:18-21 in eval mode, then `self.pool`, :218,268-271):
 * y = relu(post_scale * (conv3x3(x) + bias) + post_shift) as mau_conv3x3_fwd writes it AND pooled = MaxPool2d(2,2)(y) (floor mode),
 * (N, H/2, W/2, ldpool), in one launch: the 16-bit epilogues take the window maxima from the registers they store (activations are
 * >= 0, whose 16-bit patterns order like integers); MAU_F32 runs mau_maxpool2x2_fwd behind the convolution -- the same bits as the two
 * calls either way.  One tensor source (no x1, no broadcast embedding). */
int mau_conv3x3_fwd_pool(const void* x, int ldx, int C0, const void* wpk, const float* bias, const float* post_scale,
                         const float* post_shift, void* y, int ldy, int Cout, void* pooled, int ldpool, int dtype, int N, int H,
                         int W, mau_stream_t stream);
/* Which tile variant of the convolution kernel runs such a layer (diagnostics and tests: "did the big-tile variant run?"):
 * pixel rows of a workgroup tile (16 pixels wide), waves per workgroup, output channels per workgroup -- written to HOST ints.
 * 16-bit: (64,8,64) = <64,4,8>, (32,4,64) = <64,4,4> (two workgroups per CU, level 0), (32,8,128) = <128,4,8>, (16,..) / (8,..) =
 * the small-image forms; MAU_F32: the fp32 kernel's one tiling. */
int mau_conv3x3_variant(int dtype, int N, int H, int W, int Cin, int Cout, int* tile_rows_host, int* waves_host, int* cout_block_host,
                        int* k_groups_host);
/* The network's FIRST convolution (reference src/model.py:222 / :67 conv0_0.conv1; nn.Conv2d(spatial_channels, 64, 3, padding=1),
 * src/model.py:12) for inputs of at most mau_conv3x3_first_max_channels() = 8 channels, 16-bit modes: reads the input AS THE
 * REFERENCE'S collate_fn DELIVERS IT -- x (N,Cin,H,W) fp32 contiguous (src/dataset.py:99-106) -- and the fp32 master weights
 * w (Cout,Cin,3,3) directly (no layout kernel, no weight pack); writes y (N,H,W,ldy) in `dtype` (MAU_BF16 | MAU_F16; ldy % 8 == 0,
 * channels [Cout, ldy) zero) = conv + bias, and either
 *   slab != NULL : BatchNorm partial sums from the fp32 accumulators, [rows][2 * roundup(Cout,64)] with
 *                  rows = mau_conv3x3_first_rows(N,H,W) (one row per persistent workgroup; feed it to
 *                  mau_bn_stats_finalize_train / mau_bn_stats_sums_f64 exactly like the slab of mau_conv3x3_fwd), or
 *   post_scale/post_shift != NULL : y = relu(post_scale * (conv + bias) + post_shift) (eval-mode BatchNorm + ReLU folded in).
 * x8 (optional): the input converted to `dtype` as an NHWC tensor of 8 channels (N,H,W,8), channels >= Cin zero -- what the
 * weight gradient of this layer reads (mau_conv3x3_wgrad2 with ldx = 8); written from the same pass over x. */
int mau_conv3x3_first_max_channels(void);
int mau_conv3x3_first_rows(int N, int H, int W);
/* The same layer's weight gradient (autograd's conv weight gradient of conv0_0.conv1, src/train.py:252 with src/model.py:12,222):
 * dw (Cout,Cin,3,3) fp32 = sum over pixels of dz (x) x, complete (no second call), bitwise reproducible.  x8 = the NHWC-8 copy of the
 * input written by mau_conv3x3_first_fwd; dz (N,H,W,lddz) the gradient w.r.t. the convolution's output in `dtype`; ws = workspace of
 * mau_conv3x3_first_wgrad_ws_elems(N,H,W,Cout) floats (one compact slab per persistent workgroup).  Cin <= 8, 16-bit types. */
size_t mau_conv3x3_first_wgrad_ws_elems(int N, int H, int W, int Cout);
int mau_conv3x3_first_wgrad(const void* x8, const void* dz, int lddz, float* dw_oihw, float* ws, int Cin, int Cout, int dtype, int N,
                            int H, int W, mau_stream_t stream);
int mau_conv3x3_first_fwd(const float* x_nchw, int Cin, const float* w_oihw, const float* bias, const float* post_scale,
                          const float* post_shift, void* y, int ldy, int Cout, float* slab, void* x8, int dtype, int N, int H,
                          int W, mau_stream_t stream);
/* y = conv3x3(cat([x, broadcast(emb)], C)) + bias.
 *   x     NHWC-ld with C0 channels;
 *   emb   optional fp32 (N,E): E extra input channels, constant over (y,x) inside the image and
 *         zero in the padding halo -- fuse_embeddings (src/model.py:248-259) without ever
 *         materialising the tiled map; NULL/E=0 for ordinary convolutions;
 *   emb_ws  workspace of N*E elements of `dtype` (the embedding in the activation dtype; needed for
 *         MAU_BF16 when E > 0, ignored otherwise);
 *   wpk   forward pack for Cin = C0+E;  bias fp32 (Cout) or NULL;
 *   post_scale/post_shift  optional fp32 (Cout) pair: y = relu(post_scale*(conv+bias) + post_shift) -- eval-mode
 *         nn.BatchNorm2d + nn.ReLU (src/model.py:13-16) folded into the epilogue for the inference path
 *         (app/model_utils.py:102-109, test/evaluate.py:181-186); NULL, NULL = plain convolution;
 *   slab  optional fp32 [num_pixel_tiles][2][Cout64]: per-tile sum / sum of squares of y over the
 *         tile's valid pixels (first half of train-mode nn.BatchNorm2d, src/model.py:13,15).
 * The same entry point computes the data gradient when given dy and the `wd` pack. */
int mau_conv3x3_fwd(const void* x, int ldx, int C0, const float* emb, void* emb_ws, int E, const void* wpk,
                    const float* bias, const float* post_scale, const float* post_shift, void* y, int ldy,
                    int Cout, float* slab, int dtype, int N, int H, int W, mau_stream_t stream);
/* Same with the input channels taken from TWO tensors and the broadcast vector -- torch.cat([x, x1, broadcast(emb)], 1)
 * without materialising the concatenation: the decoder's cat([skip, up(low)], 1) (src/model.py:279-282) and U-Net++'s
 * node inputs (src/model.py:136-177).  x1 (N,H,W) NHWC-ld with C1 channels, ldx1 % 8 == 0; 16-bit dtypes only,
 * C0 % 16 == 0 (a 16-channel stage reads one tensor); C1 = 0 / x1 = NULL reduces to mau_conv3x3_fwd. */
int mau_conv3x3_fwd2(const void* x, int ldx, int C0, const void* x1, int ldx1, int C1, const float* emb, void* emb_ws,
                     int E, const void* wpk, const float* bias, const float* post_scale, const float* post_shift,
                     void* y, int ldy, int Cout, float* slab, int dtype, int N, int H, int W, mau_stream_t stream);
/* Weight gradient  dW = x (*) dy  reduced over all pixels, in two steps:
 *   mau_conv3x3_wgrad        -> acc: fp32 split-K partial slabs [nsplit][9][Cout64][Cin64], plain stores, every dtype
 *                               (nsplit = mau_conv3x3_wgrad_splits(...))
 *   mau_conv3x3_unpack_wgrad -> sums the splits in fixed order and writes OIHW fp32 (Cout,Cin,3,3).
 * x is the convolution's input (with the optional broadcast `emb` / `emb_ws` as in mau_conv3x3_fwd). */
int mau_conv3x3_wgrad_splits(int dtype, int N, int H, int W, int Cout, int Cin);
size_t mau_conv3x3_wgrad_acc_elems(int dtype, int N, int H, int W, int Cout, int Cin);
int mau_conv3x3_wgrad(const void* x, int ldx, int C0, const float* emb, void* emb_ws, int E, const void* dy,
                      int lddy, int Cout, float* acc, int dtype, int N, int H, int W, mau_stream_t stream);
/* weight gradient w.r.t. the two-tensor input of mau_conv3x3_fwd2 (Cin = C0 + C1 + E; C0 % 8 == 0). */
int mau_conv3x3_wgrad2(const void* x, int ldx, int C0, const void* x1, int ldx1, int C1, const float* emb, void* emb_ws,
                       int E, const void* dy, int lddy, int Cout, float* acc, int dtype, int N, int H, int W,
                       mau_stream_t stream);
int mau_conv3x3_unpack_wgrad(const float* acc, int nsplit, float* dw_oihw, int Cout, int Cin,
                             mau_stream_t stream);

/* ---- BatchNorm2d (+ReLU) (src/model.py:13,15,16) ------------------------ */
/* slab [rows][M] fp32 -> sums[M] fp64: column sums in two deterministic levels; `ws` is an fp64
 * workspace of mau_reduce_rows_ws_elems(rows, M) elements (NULL = single level, slow for many rows).
 * `tickets`: NULL = the two levels are two launches; else a ZEROED uint32 buffer of mau_reduce_tickets_elems() entries
 * (zero again when the call's work has finished; not to be shared by calls running concurrently on two streams): the two
 * levels are ONE launch -- the workgroup that draws a column block's last ticket adds that block's partials, in the same
 * fixed order as the two-launch form (bit-identical results). */
size_t mau_reduce_rows_ws_elems(int rows, int M);
int mau_reduce_tickets_elems(void);
int mau_reduce_rows_f64(const float* slab, int rows, int M, int ldrow, double* sums, double* ws, unsigned* tickets,
                        mau_stream_t stream);
/* same, and additionally the sums rounded to fp32 in `sums32` (the BatchNorm weight/bias gradients that autograd
 * returns, next to the fp64 sums the backward formula uses).  append > 0 (single-launch form only): sums[M] = append --
 * the local pixel count that is all-reduced together with the sums under data parallelism. */
int mau_reduce_rows_f64_f32(const float* slab, int rows, int M, int ldrow, double* sums, float* sums32,
                            double* ws, unsigned* tickets, double append, mau_stream_t stream);
/* same, result rounded to fp32. */
int mau_reduce_rows_f32(const float* slab, int rows, int M, int ldrow, float* out, double* ws, unsigned* tickets,
                        mau_stream_t stream);
/* Data-parallel forward: the statistics slab of mau_conv3x3_fwd [rows][2*Cout64] -> sums = [sum(y) (C) | sum(y^2) (C)]
 * (+ sums[2*C] = append when append > 0), one launch; ws: mau_reduce_rows_ws_elems(rows, 2*C) fp64 elements. */
int mau_bn_stats_sums_f64(const float* slab, int rows, int C, double* sums, double* ws, unsigned* tickets, double append,
                          mau_stream_t stream);
/* Train mode: sums = [sum(y) | sum(y^2)] (2*C fp64, already all-reduced over ranks when data
 * parallel), count = number of pixels summed.  Writes scale = gamma*invstd, shift = beta - mean*scale,
 * mean, invstd and updates running_mean / running_var (unbiased) / num_batches_tracked exactly as
 * nn.BatchNorm2d(momentum, eps).forward does in training.
 * count == 0: `sums` has 2*C + 1 elements and sums[2*C] is the pixel count (all-reduced together with the sums, so that
 * ranks with different local batch sizes still agree on the global statistics). */
int mau_bn_finalize_train(const double* sums, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          float momentum, float eps, float* scale, float* shift, float* mean,
                          float* invstd, int C, mau_stream_t stream);
/* Single-GPU training shortcut: conv slab [rows][2*Cout64] -> (fp64 partials in `ws`,
 * mau_bn_stats_ws_elems(rows, C) elements) -> the outputs of mau_bn_finalize_train; one launch with `tickets` (see
 * mau_reduce_rows_f64), two launches with tickets == NULL. */
size_t mau_bn_stats_ws_elems(int rows, int C);
int mau_bn_stats_finalize_train(const float* slab, int rows, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                                double* ws, unsigned* tickets, int C, mau_stream_t stream);
/* Eval mode: scale/shift from the running statistics (mean/invstd outputs optional, may be NULL). */
int mau_bn_coeffs_eval(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, float* mean,
                       float* invstd, int C, mau_stream_t stream);
/* a = relu(scale*y + shift) on NHWC-ld (pad channels written as 0). */
int mau_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift, void* a, int lda,
                      int dtype, int64_t npix, int C, mau_stream_t stream);
/* a = relu(scale*y + shift) AND pooled (N,H/2,W/2) = nn.MaxPool2d(2,2)(a) in ONE pass over y: an encoder block's
 * output feeds the next level through the pool and the decoder through the skip (src/model.py:268-271). */
/* argidx (optional, NULL to skip) [N][H/2][W/2][roundup(C,8)/8] uint16: per pooled window and 8-channel vector, 2 bits per
 * channel = position 2*dy + dx of the window's first maximum (ATen's scan order, strict '>'): the pool's backward routing,
 * consumed by mau_pool_bn_bwd_reduce / _apply. */
int mau_bn_relu_apply_pool(const void* y, int ldy, const float* scale, const float* shift, void* a, int lda,
                           void* pooled, int ldp, unsigned short* argidx, int dtype, int N, int H, int W, int C,
                           mau_stream_t stream);
/* Backward, pass 1: dz = da * [scale*y+shift > 0]; per-block partial sums of dz and dz*xhat
 * (xhat = (y-mean)*invstd) into slab [rows][2][ldslab]; returns rows via *rows_out (host). */
int mau_bn_relu_bwd_reduce(const void* da, int ldda, const void* y, int ldy, const float* scale,
                           const float* shift, const float* mean, const float* invstd, float* slab,
                           int ldslab, int dtype, int64_t npix, int C, mau_stream_t stream);
/* Backward, pass 2: dy = scale*(dz - s1/count - xhat*s2/count); sums = [s1 | s2] fp64 (2*C);
 * count == 0: the count is sums[2*C] (see mau_bn_finalize_train). */
int mau_bn_relu_bwd_apply(const void* da, int ldda, const void* y, int ldy, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const double* sums,
                          double count, void* dy, int lddy, int dtype, int64_t npix, int C,
                          mau_stream_t stream);
int mau_bn_bwd_rows(int64_t npix);

/* ---- BatchNorm2d + ReLU fused with the streaming operator behind it (csrc/bn_fused.hip) ----
 * An encoder block's output feeds nn.MaxPool2d(2,2) AND a skip connection (src/model.py:268-271 / :279-282).  Backward of its
 * second BatchNorm without ever writing the incoming gradient da = dskip + maxpool_backward(dpl): both passes recompute it
 * (the arg-max from `argidx`, written by mau_bn_relu_apply_pool in the forward pass).  dpl (N,H/2,W/2) [with argidx] or
 * dskip (N,H,W) may be NULL, not both.  slab / sums / count as mau_bn_relu_bwd_reduce / _apply; bit-identical to
 * mau_maxpool2x2_bwd_add + mau_bn_relu_bwd_reduce + mau_bn_relu_bwd_apply. */
int mau_pool_bn_bwd_reduce(const void* y, int ldy, const void* dpl, int lddpl, const unsigned short* argidx,
                           const void* dskip, int lddskip,
                           const float* scale, const float* shift, const float* mean, const float* invstd, float* slab,
                           int ldslab, int dtype, int N, int H, int W, int C, mau_stream_t stream);
int mau_pool_bn_bwd_apply(const void* y, int ldy, const void* dpl, int lddpl, const unsigned short* argidx,
                          const void* dskip, int lddskip,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          const double* sums, double count, void* dy, int lddy, int dtype, int N, int H, int W, int C,
                          mau_stream_t stream);
/* The final 1x1 conv + tanh (src/model.py:241,284-292) directly behind the last block's second BatchNorm (C <=
 * mau_head_bn_max_channels()): out = head(relu(scale*y + shift)) from the RAW conv output y -- the activation is never
 * written; backward in two passes over y: (1) BatchNorm partial sums (slab as mau_bn_relu_bwd_reduce) + the head's dW / db
 * partials (head_slab as mau_head_bwd), (2) dy of the conv; da = W^T dz is recomputed from dout both times.  Bit-identical
 * to mau_bn_relu_apply + mau_head_fwd / mau_head_bwd + mau_bn_relu_bwd_reduce + mau_bn_relu_bwd_apply.
 * Images of at least 32 pixels (HW >= 32) and N * HW < 2^31: a thread walks pixels in steps of 32 with one conditional wrap into
 * the next image (smaller images: the separate kernels). */
int mau_head_bn_max_channels(void);
int mau_head_bn_fwd(const void* y, int ldy, const float* scale, const float* shift, const float* w, const float* b, float* out,
                    int tanh0, int dtype, int N, int HW, int C, int Co, mau_stream_t stream);
int mau_head_bn_bwd_reduce(const void* y, int ldy, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* w, const float* out, const float* dout, float* bn_slab,
                           int ldslab, float* head_slab, int tanh0, int dtype, int N, int HW, int C, int Co,
                           mau_stream_t stream);
int mau_head_bn_bwd_apply(const void* y, int ldy, const float* scale, const float* shift, const float* mean,
                          const float* invstd, const double* sums, double count, const float* w, const float* out,
                          const float* dout, void* dy, int lddy, int tanh0, int dtype, int N, int HW, int C, int Co,
                          mau_stream_t stream);

/* dst = srcs[0] + srcs[1] + ... + srcs[k-1] (k <= mau_sum_tensors_max() NHWC-ld tensors of C channels, pixel pitches lds[i]; HOST
 * arrays of k device pointers / pitches): the sum autograd forms when an activation has several readers (the row slots of the U-Net++,
 * src/model.py:136-177), in one pass -- fp32 sums in the order given, one rounding. */
int mau_sum_tensors_max(void);
int mau_sum_tensors(const void* const* srcs, const int* lds, int k, void* dst, int lddst, int dtype, int64_t npix, int C,
                    mau_stream_t stream);

/* ---- the broadcast embedding of a convolution as a rank-one term (csrc/embfold.hip) ----
 * fuse_embeddings / the U-Net++ nodes' emb_map (src/model.py:248-259, :111-121) put E spatially constant channels behind the tensors a
 * 3x3 conv reads.  W_eff = [ W[:, :Ct] | T ], T[co][i][tap] = sum_e W[co][Ct+e][tap] * emb[i][e]  (Cout x (Ct+Ep) x 3 x 3, Ep >= N,
 * columns i >= N zero): the same convolution over Ct tensor channels + the N x Ep IDENTITY as "embedding" gives the same output with
 * Ct + Ep instead of Ct + E input channels.  bwd: dW (Cout x (Ct+E) x 3 x 3) and demb (N x E) from dW_eff (either may be NULL).
 * fp32, fixed summation order. */
int mau_emb_fold_fwd(const float* w, const float* emb, float* weff, int Cout, int Ct, int E, int N, int Ep, mau_stream_t stream);
size_t mau_emb_fold_ws_elems(int Cout, int N, int E);   /* floats of ws (needed when demb is asked for) */
int mau_emb_fold_bwd(const float* w, const float* emb, const float* dweff, float* dw, float* demb, float* ws, int Cout, int Ct, int E,
                     int N, int Ep, mau_stream_t stream);

/* ---- MaxPool2d(2,2) (src/model.py:57,218) -------------------------------- */
int mau_maxpool2x2_fwd(const void* x, int ldx, void* y, int ldy, int dtype, int N, int H, int W, int C,
                       mau_stream_t stream);
/* dx (N,H,W) = scatter of dy (N,H/2,W/2) to the first maximum of each window; rest zero. */
int mau_maxpool2x2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int dtype,
                       int N, int H, int W, int C, mau_stream_t stream);
/* dx = dskip + maxpool2x2 backward: x feeds the pool AND a skip connection (src/model.py:268-271 / :279-282); the two
 * gradients autograd would add in a separate pass are combined while dx is written. */
int mau_maxpool2x2_bwd_add(const void* x, int ldx, const void* dy, int lddy, const void* dskip, int lddskip, void* dx,
                           int lddx, int dtype, int N, int H, int W, int C, mau_stream_t stream);

/* ---- bilinear resize, align_corners=True (src/model.py:111-121,219,243-246) ---- */
/* dst[..., choff:choff+C] = resize(src (N,h,w,C)) to (H,W); other channels of dst untouched. */
int mau_resize_bilinear_fwd(const void* src, int ldsrc, int h, int w, void* dst, int lddst, int choff,
                            int dtype, int N, int H, int W, int C, mau_stream_t stream);
/* The same upsampling (h <= H, w <= W) reading the RAW conv output y of a block whose BatchNorm + ReLU has this resize as its
 * only consumer (the U-Net's bottleneck and decoder blocks, src/model.py:279-282): relu(scale*y + shift), rounded to `dtype`,
 * is formed on the four corners in registers; bit-identical to mau_bn_relu_apply + mau_resize_bilinear_fwd. */
int mau_resize_bilinear_bn_fwd(const void* y, int ldy, int h, int w, const float* scale, const float* shift, void* dst,
                               int lddst, int choff, int dtype, int N, int H, int W, int C, mau_stream_t stream);
/* dsrc (N,h,w,C) = adjoint of the above applied to ddst[..., choff:choff+C] (gather form, no atomics). */
int mau_resize_bilinear_bwd(const void* ddst, int ldddst, int choff, int H, int W, void* dsrc, int lddsrc,
                            int dtype, int N, int h, int w, int C, mau_stream_t stream);
/* dst[..., choff:choff+C] = src[..., :C]  (channel concat building block, src/model.py:279-282);
 * also zero-fills dst channels [choff+C, zero_to) when zero_to > choff+C. */
int mau_copy_channels(const void* src, int ldsrc, void* dst, int lddst, int choff, int zero_to, int dtype,
                      int64_t npix, int C, mau_stream_t stream);
/* dst[n, :, choff:choff+E] = emb[n] broadcast over the HW pixels (materialised form of
 * fuse_embeddings, src/model.py:248-259; the fused form is the `emb` source of mau_conv3x3_fwd). */
int mau_bcast_fill(const float* emb, void* dst, int lddst, int choff, int zero_to, int dtype, int N, int HW,
                   int E, mau_stream_t stream);
/* demb (N,E) fp32 = sum over pixels of dx[..., choff:choff+E]  (adjoint of the embedding broadcast). */
/* ws: fp32 workspace of mau_bcast_bwd_ws_elems(N, HW, E) elements (per-chunk partial sums, summed in
 * fixed order); NULL selects the slow element-granular path. */
size_t mau_bcast_bwd_ws_elems(int N, int HW, int E);
int mau_bcast_bwd(const void* dx, int lddx, int choff, float* demb, float* ws, int dtype, int N, int HW, int E,
                  mau_stream_t stream);

/* ---- head: 1x1 conv + tanh on channel 0 (src/model.py:241,284-292) -------- */
/* out (N,Co,H,W) fp32 NCHW; tanh on channel 0 iff tanh0 != 0 (the reference does so iff Co == 2). */
int mau_head_fwd(const void* a, int lda, const float* w, const float* b, float* out, int tanh0, int dtype,
                 int N, int HW, int C, int Co, mau_stream_t stream);
/* da NHWC-ld = W^T dz with dz = dout * (1 - out^2 on the tanh channel); slab
 * [mau_head_bwd_rows][mau_head_bwd_rowlen] holds per-block partial sums: for output channel o,
 * row[o*(C8+8) + c] = partial dW[o][c] and row[o*(C8+8) + C8] = partial db[o]  (C8 = roundup(C,8)). */
int mau_head_bwd(const void* a, int lda, const float* w, const float* out, const float* dout, void* da,
                 int ldda, float* slab, int tanh0, int dtype, int N, int HW, int C, int Co,
                 mau_stream_t stream);
int mau_head_bwd_rows(int N, int HW);
int mau_head_bwd_rowlen(int C, int Co);

/* ---- MetadataEncoder: Linear(F,32) -> ReLU -> Linear(32,D) (src/model.py:38-48) ---- */
int mau_meta_mlp_fwd(const float* md, const float* w0, const float* b0, const float* w2, const float* b2,
                     float* hidden, float* emb, int N, int F, int Hd, int D, mau_stream_t stream);
/* dhidden_ws: fp32 workspace (N,Hd). */
int mau_meta_mlp_bwd(const float* md, const float* w0, const float* w2, const float* hidden,
                     const float* demb, float* dw0, float* db0, float* dw2, float* db2, float* dhidden_ws,
                     int N, int F, int Hd, int D, mau_stream_t stream);

/* ---- TemporalEncoder recurrence: nn.LSTM(input_size=1, hidden_size=H, batch_first=True), last hidden state
 *      (src/model.py:23-34; the sequence is 828 monthly temperatures in the reference's data, conf/config.yaml:20) ---- */
/* Largest hidden size the persistent kernels support (one gate row per thread). */
int mau_lstm_max_hidden(void);
/* x (B,T) fp32; w_ih (4H) [input_size 1], w_hh (4H,H), b_ih, b_hh (4H): torch's parameter layout, gate order i,f,g,o.
 * h_last (B,H) = h_T.  gates (B,T,4H) post-activation and cells (B,T,H), both or neither: saved for mau_lstm_bwd. */
int mau_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                 float* h_last, float* gates, float* cells, int B, int T, int H, mau_stream_t stream);
/* dh_last (B,H) -> dw_ih (4H), dw_hh (4H,H), db_ih = db_hh (4H); ws: fp32 workspace of mau_lstm_bwd_ws_elems(B,T,H)
 * elements (pre-activation gate gradients of every step + partial sums, added in a fixed order). */
size_t mau_lstm_bwd_ws_elems(int B, int T, int H);
int mau_lstm_bwd(const float* x, const float* w_hh, const float* gates, const float* cells, const float* dh_last,
                 float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* ws, int B, int T, int H,
                 mau_stream_t stream);

/* nn.Linear of TemporalEncoder.fc (src/model.py:27,34): out (N,D) = x (N,F) w^T + b, w (D,F) as torch stores it. */
int mau_linear_fwd(const float* x, const float* w, const float* b, float* out, int N, int F, int D, mau_stream_t stream);
/* dx (N,F) (optional, NULL to skip), dw (D,F), db (D) from dout (N,D). */
int mau_linear_bwd(const float* x, const float* w, const float* dout, float* dx, float* dw, float* db, int N, int F,
                   int D, mau_stream_t stream);

/* ---- optimizer: torch.optim.AdamW (src/train.py:213-214,255) on every 3x3 convolution weight of a network, fused with the
 *      re-pack of the updated weights (mau_conv3x3_pack_weights): ONE launch, each parameter streamed once ----
 * Table rows (mau_adamw_pack_desc_bytes() each, filled on the HOST by mau_adamw_pack_desc_fill, copied to the device by the
 * caller): w OIHW fp32 (Cout,Cin,3,3) updated in place, grad, exp_avg, exp_avg_sq of the same shape, wf / wd = the packs of
 * mau_conv3x3_pack_weights for `dtype` (both NULL: no pack); tile0 = *next_tile_host of the previous row (0 for row 0).
 * step: DEVICE float, the step count t of this update (already incremented).  p <- p(1 - lr*wd); m <- b1 m + (1-b1) g;
 * v <- b2 v + (1-b2) g^2; p <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps). */
size_t mau_adamw_pack_desc_bytes(void);
int mau_adamw_pack_desc_fill(void* descs_host, int index, float* w, const float* grad, float* exp_avg, float* exp_avg_sq,
                             void* wf, void* wd, int Cout, int Cin, int tile0, int* next_tile_host);
int mau_adamw_pack_step(const void* descs, int n, int total_tiles, int dtype, const float* step, float lr, float beta1,
                        float beta2, float eps, float weight_decay, mau_stream_t stream);

/* ---- loss: F.mse_loss (src/utils/losses.py:27-39) -------------------------- */
/* loss[0] = mean((out-tgt)^2) (fp64 accumulation, fixed order); dout (optional) = 2*(out-tgt)/n;
 * partial: fp64 workspace of mau_mse_blocks(n) elements. */
int mau_mse_blocks(int64_t n);
int mau_mse_fwd_bwd(const float* out, const float* tgt, double* partial, float* loss, float* dout, int64_t n,
                    mau_stream_t stream);

/* ---- L1 + gradient (finite-difference) losses (src/utils/losses.py:5-25 gradient_loss, :68 F.l1_loss) ---- */
/* out/tgt (B,C,H,W) fp32.  terms[0..2] = { mean|out-tgt|, mean| |dy out|-|dy tgt| |, mean| |dx out|-|dx tgt| | }
 * (gradient_loss = terms[1] + terms[2]);  dout (optional) = d/d out of  w_l1*terms[0] + w_grad*(terms[1]+terms[2]).
 * partial: fp64 workspace of 3*mau_l1_gradient_blocks(n) elements, n = B*C*H*W. */
int mau_l1_gradient_blocks(int64_t n);
int mau_l1_gradient_loss(const float* out, const float* tgt, double* partial, float* terms, float* dout, float w_l1,
                         float w_grad, int B, int C, int H, int W, mau_stream_t stream);

/* ---- SSIM term of compute_loss_l1_grad_ssim (src/utils/losses.py:70-97; piq.ssim defaults, PARITY UNPINNED: piq is
 *      not available to generate a fixture; value only -- the reference detaches it at :96) ---- */
/* out/tgt (B,C,H,W) fp32.  prep != 0 applies the reference's channel preparation while loading (:72-84): channel 0 ->
 * (v+1)/2, channel 1 -> clamp(v,0,1).  per_image (B) = SSIM per sample (mean over channels), loss[0] = 1 - mean(per_image).
 * ws: fp64 workspace of mau_ssim_ws_elems(B,C,H,W) elements (per-tile sums, added in a fixed order). */
size_t mau_ssim_ws_elems(int B, int C, int H, int W);
int mau_ssim_loss(const float* out, const float* tgt, double* ws, float* per_image, float* loss, int prep, int B, int C,
                  int H, int W, mau_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MAU_HIP_H */
