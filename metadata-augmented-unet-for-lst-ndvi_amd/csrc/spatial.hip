// spatial.hip -- layout change at the module boundary, MaxPool2d(2,2), bilinear resize
// (align_corners=True) fused with the channel concat, and the adjoint of the embedding broadcast.
// All HBM-streaming, one 8-channel vector (16 B bf16 / 32 B f32) per lane.
// Reference: src/model.py:57,218 (pool), :219,243-246,111-121 (upsample / _upsample_match),
// :279-282,136-177 (torch.cat on channels), :248-259,98-108 (embedding broadcast).
#include <stdlib.h>
#include "mau_common.h"

namespace mau {

// ---- NCHW fp32 <-> NHWC-ld T -------------------------------------------------------------
// tile of 64 pixels x 8 channels through LDS so that both sides are coalesced
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C,
                                                           int64_t HW, int ld) {
  // grid: (pixel blocks of 256, channel groups of 8, N)
  __shared__ float tile[8][256 + 1];
  const int n = blockIdx.z, c0 = blockIdx.y * 8;
  const int64_t p0 = (int64_t)blockIdx.x * 256;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    const int64_t p = p0 + threadIdx.x;
    tile[j][threadIdx.x] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  const int64_t p = p0 + threadIdx.x;
  if (p < HW && c0 < ld) {
    F8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = tile[j][threadIdx.x];
    store8<T>(dst + ((size_t)n * HW + p) * ld + c0, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C,
                                                           int64_t HW, int ld) {
  const int n = blockIdx.z, c0 = blockIdx.y * 8;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const F8 v = load8<T>(src + ((size_t)n * HW + p) * ld + c0);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (c0 + j < C) dst[((size_t)n * C + c0 + j) * HW + p] = v.v[j];
}

// ---- compact tile -> network input (input pipeline, reference src/dataset.py:53-72 + processing_10m/process.py:176-181) ----
// The reference stores and ships 23 fp32 planes per tile of which 18 are the one-hot expansion of two 9-class maps.
// Here the two class maps travel as uint8 and the one-hot channels are generated while the NHWC-ld tensor is written:
// out[n, y, x, :] = [onehot(cls_a) | cont[0..ncont) | onehot(cls_b) | 0 pad], x mirrored when flip[n] != 0 (RandomFlip).
// grid = (x chunks, rows, images); one thread per pixel, 16-byte channel-group stores.
template <typename T>
__global__ __launch_bounds__(256) void pack_tile_onehot_kernel(const uint8_t* __restrict__ cls_a, const uint8_t* __restrict__ cls_b,
                                                               const float* __restrict__ cont, const uint8_t* __restrict__ flip,
                                                               T* __restrict__ out, int ld, int H, int W, int nc, int ncont) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int y = blockIdx.y, n = blockIdx.z;
  const int xs = (flip != nullptr && flip[n]) ? W - 1 - x : x;
  const size_t hw = (size_t)H * W, ps = (size_t)y * W + xs;
  const int a = cls_a[n * hw + ps], b = cls_b[n * hw + ps];
  float cv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cv[j] = j < ncont ? cont[((size_t)n * ncont + j) * hw + ps] : 0.f;
  T* o = out + ((size_t)n * hw + (size_t)y * W + x) * ld;
  for (int g = 0; g < ld; g += 8) {
    F8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = g + j;
      float f = 0.f;
      if (c < nc) f = (a == c) ? 1.f : 0.f;
      else if (c < nc + ncont) {
        const int k = c - nc;
        f = k == 0 ? cv[0] : k == 1 ? cv[1] : k == 2 ? cv[2] : k == 3 ? cv[3] : k == 4 ? cv[4] : k == 5 ? cv[5] : k == 6 ? cv[6] : cv[7];
      } else if (c < 2 * nc + ncont) f = (b == c - nc - ncont) ? 1.f : 0.f;
      v.v[j] = f;
    }
    store8<T>(o + g, v);
  }
}

// dst[n,c,y,x] = src[n,c,y, flip[n] ? W-1-x : x]   (the target half of RandomFlip, src/dataset.py:139)
__global__ __launch_bounds__(256) void flip_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const uint8_t* __restrict__ flip,
                                                        int rows_per_image, int W) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const int n = blockIdx.z;
  const size_t row = ((size_t)n * rows_per_image + blockIdx.y) * W;
  dst[row + x] = src[row + (flip[n] ? W - 1 - x : x)];
}

// ---- MaxPool2d(2,2), floor mode -------------------------------------------------------------
// grid = (x-chunks of the output row, output rows, images): no 64-bit div/mod per element
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int H, int W,
                                                          int C8) {
  const int Ho = H / 2, Wo = W / 2, nv = C8 >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wo * nv) return;
  const int xo = idx / nv, c = (idx - xo * nv) * 8;
  const int yo = blockIdx.y, n = blockIdx.z;
  const T* b = x + (((size_t)n * H + 2 * yo) * W + 2 * xo) * ldx + c;
  const F8 v00 = load8<T>(b), v01 = load8<T>(b + ldx), v10 = load8<T>(b + (size_t)W * ldx), v11 = load8<T>(b + (size_t)W * ldx + ldx);
  F8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = fmaxf(fmaxf(v00.v[j], v01.v[j]), fmaxf(v10.v[j], v11.v[j]));
  store8<T>(y + (((size_t)n * Ho + yo) * Wo + xo) * ldy + c, o);
}

// one thread per INPUT pixel vector: grad goes to the first maximum of the window (scan order
// (0,0),(0,1),(1,0),(1,1) with strict '>' as ATen's max_pool2d), uncovered odd rows/cols get 0.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                          const T* __restrict__ dskip, int lddskip,
                                                          T* __restrict__ dx, int lddx, int H, int W, int C8) {
  const int Ho = H / 2, Wo = W / 2, nv = C8 >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * nv) return;
  const int xi = idx / nv, c = (idx - xi * nv) * 8;
  const int yi = blockIdx.y, n = blockIdx.z;
  F8 o = zero8();
  const int yo = yi >> 1, xo = xi >> 1;
  if (yo < Ho && xo < Wo) {
    const T* b = x + (((size_t)n * H + 2 * yo) * W + 2 * xo) * ldx + c;
    const F8 v00 = load8<T>(b), v01 = load8<T>(b + ldx), v10 = load8<T>(b + (size_t)W * ldx), v11 = load8<T>(b + (size_t)W * ldx + ldx);
    const F8 g = load8<T>(dy + (((size_t)n * Ho + yo) * Wo + xo) * lddy + c);
    const int me = (yi & 1) * 2 + (xi & 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int arg = 0;
      float m = v00.v[j];
      if (v01.v[j] > m) { m = v01.v[j]; arg = 1; }
      if (v10.v[j] > m) { m = v10.v[j]; arg = 2; }
      if (v11.v[j] > m) { m = v11.v[j]; arg = 3; }
      o.v[j] = (arg == me) ? g.v[j] : 0.f;
    }
  }
  if (dskip != nullptr) {        // the other consumer of x (a skip connection): its gradient is added in the same pass
    const F8 sk = load8<T>(dskip + (((size_t)n * H + yi) * W + xi) * lddskip + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] += sk.v[j];
  }
  store8<T>(dx + (((size_t)n * H + yi) * W + xi) * lddx + c, o);
}

// ---- bilinear, align_corners=True ------------------------------------------------------------
// ATen: scale = (in-1)/(out-1) (0 when out == 1); src = scale*dst; i0 = (int)src; i1 = i0 + (i0 < in-1);
// l1 = src - i0; l0 = 1 - l1.
__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
__device__ __forceinline__ void ac_src(float scale, int dst, int in, int& i0, int& i1, float& l0, float& l1) {
  // (no FMA contraction: ATen rounds the product before subtracting the index; contracted, l1 = fma(scale, dst, -i0) differs by
  //  up to an ulp of the SOURCE COORDINATE (1e-5 at 256 pixels), and differently in every kernel the compiler inlines this into)
#pragma clang fp contract(off)
  const float s = scale * (float)dst;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// one interpolated value, every operation rounded on its own (no FMA contraction): both forward kernels below must give
// the same bits for the same destination pixel -- left to the compiler, the loop-carried form in resize_fwd_cell_kernel
// contracted differently from resize_fwd_kernel (1 ulp; enough to move a cancellation-heavy full-size gradient by 3e-3)
__device__ __forceinline__ float lerp4(float ly0, float ly1, float lx0, float lx1, float v00, float v01, float v10, float v11) {
#pragma clang fp contract(off)
  const float top = lx0 * v00 + lx1 * v01;
  const float bot = lx0 * v10 + lx1 * v11;
  return ly0 * top + ly1 * bot;
}

// grid = (x-chunks of the destination row, destination rows, images); the row's (y0, y1, ly) are block-uniform
template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const T* __restrict__ src, int ldsrc, int h, int w, T* __restrict__ dst,
                                                         int lddst, int choff, int H, int W, int C8) {
  const int nv = C8 >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * nv) return;
  const int xo = idx / nv, c = (idx - xo * nv) * 8;
  const int yo = blockIdx.y, n = blockIdx.z;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  ac_src(sy, yo, h, y0, y1, ly0, ly1);
  ac_src(sx, xo, w, x0, x1, lx0, lx1);
  const T* b = src + (size_t)n * h * w * ldsrc + c;
  const F8 v00 = load8<T>(b + ((size_t)y0 * w + x0) * ldsrc), v01 = load8<T>(b + ((size_t)y0 * w + x1) * ldsrc);
  const F8 v10 = load8<T>(b + ((size_t)y1 * w + x0) * ldsrc), v11 = load8<T>(b + ((size_t)y1 * w + x1) * ldsrc);
  F8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    o.v[j] = lerp4(ly0, ly1, lx0, lx1, v00.v[j], v01.v[j], v10.v[j], v11.v[j]);
  store8<T>(dst + (((size_t)n * H + yo) * W + xo) * lddst + choff + c, o);
}

// Upsampling (scale <= 1 both ways: every decoder resize of the models) by SOURCE CELL: thread = (source pixel (ys, xs), 8
// channels) loads the cell's four corners once and writes every destination pixel whose (y0, x0) is that cell -- 2x2 of
// them at scale 1/2, never more than 3x3 -- with exactly resize_fwd_kernel's coordinates and arithmetic (bit-identical).
// 4 loads per ~4 stores instead of 4 per 1, a quarter of the workgroups: resize_fwd_kernel ran at 3.1-3.6 TB/s of its own
// traffic where a plain fill of the destination reaches 6.9 (scripts/write_bw.py) -- it was bound by its one-output
// threads, not by the write stream.
// BN = true: `src` is the RAW output y of a conv whose BatchNorm + ReLU has this upsample as its only consumer (the U-Net's
// decoder blocks and bottleneck, reference src/model.py:279-282): a = relu(scale*y + shift), rounded to T exactly as
// bn_relu_apply would have stored it, is formed on the four corners in registers -- the activation is never written
// (bit-identical to the two-pass form).
template <typename T, bool BN>
__global__ __launch_bounds__(256) void resize_fwd_cell_kernel(const T* __restrict__ src, int ldsrc, int h, int w, T* __restrict__ dst,
                                                              int lddst, int choff, int H, int W, int C8, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int C) {
  const int nv = C8 >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= w * nv) return;
  const int xs = idx / nv, c = (idx - xs * nv) * 8;
  const int ys = blockIdx.y, n = blockIdx.z;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  const int ys1 = ys + (ys < h - 1 ? 1 : 0), xs1 = xs + (xs < w - 1 ? 1 : 0);
  const T* b = src + (size_t)n * h * w * ldsrc + c;
  F8 v00 = load8<T>(b + ((size_t)ys * w + xs) * ldsrc), v01 = load8<T>(b + ((size_t)ys * w + xs1) * ldsrc);
  F8 v10 = load8<T>(b + ((size_t)ys1 * w + xs) * ldsrc), v11 = load8<T>(b + ((size_t)ys1 * w + xs1) * ldsrc);
  if constexpr (BN) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sc = coef(scale, c + e, C), sh = coef(shift, c + e, C);
      v00.v[e] = opaque((float)(T)opaque(fmaxf(fmaf(v00.v[e], sc, sh), 0.f)));
      v01.v[e] = opaque((float)(T)opaque(fmaxf(fmaf(v01.v[e], sc, sh), 0.f)));
      v10.v[e] = opaque((float)(T)opaque(fmaxf(fmaf(v10.v[e], sc, sh), 0.f)));
      v11.v[e] = opaque((float)(T)opaque(fmaxf(fmaf(v11.v[e], sc, sh), 0.f)));
    }
  }
  // destination rows / columns whose source index can be this cell (loose bounds, exact test below)
  int ja = 0, jb = H - 1, ka = 0, kb = W - 1;
  if (sy > 0.f) {
    ja = max(0, (int)floorf((float)ys / sy) - 1);
    jb = min(H - 1, (int)ceilf(((float)ys + 1.f) / sy) + 1);
  }
  if (sx > 0.f) {
    ka = max(0, (int)floorf((float)xs / sx) - 1);
    kb = min(W - 1, (int)ceilf(((float)xs + 1.f) / sx) + 1);
  }
  for (int j = ja; j <= jb; ++j) {
    int y0, y1;
    float ly0, ly1;
    ac_src(sy, j, h, y0, y1, ly0, ly1);
    if (y0 != ys) continue;
    for (int k = ka; k <= kb; ++k) {
      int x0, x1;
      float lx0, lx1;
      ac_src(sx, k, w, x0, x1, lx0, lx1);
      if (x0 != xs) continue;
      F8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o.v[e] = lerp4(ly0, ly1, lx0, lx1, v00.v[e], v01.v[e], v10.v[e], v11.v[e]);
      store8<T>(dst + (((size_t)n * H + j) * W + k) * lddst + choff + c, o);
    }
  }
}

// Upsamplings by at most ~2 (every decoder resize of the models): a COLUMN of RY source cells per thread.  The (at most 3)
// destination columns of the thread's source column are set up once, a cell's bottom corners become the next cell's top
// corners (two loads per cell instead of four), and the next row's loads are in flight while the current cell's destination
// pixels are written: 1.0-1.4x the one-cell-per-thread form.  BN: the source is the RAW conv output of a block whose
// BatchNorm + ReLU has this upsample as its only consumer -- relu(scale*y + shift), rounded to T as bn_relu_apply stores it, is
// formed on the corners in registers.  Coordinates and interpolation arithmetic are resize_fwd_kernel's (bit-identical).
template <typename T, bool BN>
__global__ __launch_bounds__(256) void resize_rows_kernel(const T* __restrict__ src, int ldsrc, int h, int w, T* __restrict__ dst,
                                                             int lddst, int choff, int H, int W, int C8, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int C, int RY) {
  const int nv = C8 >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= w * nv) return;
  const int xs = idx / nv, c = (idx - xs * nv) * 8;
  const int n = blockIdx.z, ys0 = blockIdx.y * RY, ys_end = min(h, ys0 + RY);
  float sc[8], sh[8];
  if constexpr (BN) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = coef(scale, c + e, C);
      sh[e] = coef(shift, c + e, C);
    }
  }
  auto bnr = [&](const F8& v) {
    if constexpr (!BN) return v;
    F8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = opaque((float)(T)opaque(fmaxf(fmaf(v.v[e], sc[e], sh[e]), 0.f)));
    return o;
  };
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  const int xs1 = xs + (xs < w - 1 ? 1 : 0);
  // destination columns whose left source column is xs (at most 3 at scales >= ~1/2: the launcher guarantees it)
  int kcol[3] = {0, 0, 0}, ncol = 0;
  float clx0[3] = {0.f, 0.f, 0.f}, clx1[3] = {0.f, 0.f, 0.f};
  {
    int ka = 0, kb = W - 1;
    if (sx > 0.f) {
      ka = max(0, (int)floorf((float)xs / sx) - 1);
      kb = min(W - 1, (int)ceilf(((float)xs + 1.f) / sx) + 1);
    }
    for (int k = ka; k <= kb; ++k) {
      int x0, x1;
      float lx0, lx1;
      ac_src(sx, k, w, x0, x1, lx0, lx1);
      if (x0 == xs && ncol < 3) {
        kcol[ncol] = k;
        clx0[ncol] = lx0;
        clx1[ncol] = lx1;
        ++ncol;
      }
    }
  }
  const T* b = src + (size_t)n * h * w * ldsrc + c;
  F8 t0 = bnr(load8<T>(b + ((size_t)ys0 * w + xs) * ldsrc)), t1 = bnr(load8<T>(b + ((size_t)ys0 * w + xs1) * ldsrc));
  int yn = min(ys0 + 1, h - 1);
  F8 nb0 = load8<T>(b + ((size_t)yn * w + xs) * ldsrc), nb1 = load8<T>(b + ((size_t)yn * w + xs1) * ldsrc);
  for (int ys = ys0; ys < ys_end; ++ys) {
    const F8 b0 = bnr(nb0), b1 = bnr(nb1);
    yn = min(ys + 2, h - 1);                              // the next cell's bottom row: in flight while this cell is written
    nb0 = load8<T>(b + ((size_t)yn * w + xs) * ldsrc);
    nb1 = load8<T>(b + ((size_t)yn * w + xs1) * ldsrc);
    int ja = 0, jb = H - 1;
    if (sy > 0.f) {
      ja = max(0, (int)floorf((float)ys / sy) - 1);
      jb = min(H - 1, (int)ceilf(((float)ys + 1.f) / sy) + 1);
    }
    for (int j = ja; j <= jb; ++j) {
      int y0, y1;
      float ly0, ly1;
      ac_src(sy, j, h, y0, y1, ly0, ly1);
      if (y0 != ys) continue;
      T* drow = dst + ((size_t)n * H + j) * W * lddst + choff + c;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (q < ncol) {
          F8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o.v[e] = lerp4(ly0, ly1, clx0[q], clx1[q], t0.v[e], t1.v[e], b0.v[e], b1.v[e]);
          store8<T>(drow + (size_t)kcol[q] * lddst, o);
        }
    }
    t0 = b0;
    t1 = b1;
  }
}

// adjoint in gather form: every source pixel sums the destination pixels that read it, with the
// same (i0, i1, l0, l1) arithmetic as the forward pass.  grid = (x-chunks of the source row, source rows, images)
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ ddst, int ldddst, int choff, int H, int W,
                                                         T* __restrict__ dsrc, int lddsrc, int h, int w, int C8) {
  const int nv = C8 >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= w * nv) return;
  const int xi = idx / nv, c = (idx - xi * nv) * 8;
  const int yi = blockIdx.y, n = blockIdx.z;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  // destination rows/cols whose source interval can touch (yi, xi): |s*j - i| < 1 (loose bounds, exact test inside)
  int ja = 0, jb = H - 1, ka = 0, kb = W - 1;
  if (sy > 0.f) {
    ja = (int)floorf(((float)yi - 1.f) / sy) - 1;
    jb = (int)ceilf(((float)yi + 1.f) / sy) + 1;
    if (ja < 0) ja = 0;
    if (jb > H - 1) jb = H - 1;
  }
  if (sx > 0.f) {
    ka = (int)floorf(((float)xi - 1.f) / sx) - 1;
    kb = (int)ceilf(((float)xi + 1.f) / sx) + 1;
    if (ka < 0) ka = 0;
    if (kb > W - 1) kb = W - 1;
  }
  F8 acc = zero8();
  for (int j = ja; j <= jb; ++j) {
    int y0, y1;
    float ly0, ly1;
    ac_src(sy, j, h, y0, y1, ly0, ly1);
    if (y0 != yi && y1 != yi) continue;
    float wy = 0.f;
    if (y0 == yi) wy += ly0;
    if (y1 == yi) wy += ly1;
    for (int k = ka; k <= kb; ++k) {
      int x0, x1;
      float lx0, lx1;
      ac_src(sx, k, w, x0, x1, lx0, lx1);
      if (x0 != xi && x1 != xi) continue;
      float wx = 0.f;
      if (x0 == xi) wx += lx0;
      if (x1 == xi) wx += lx1;
      const F8 g = load8<T>(ddst + (((size_t)n * H + j) * W + k) * ldddst + choff + c);
      const float wgt = wy * wx;
#pragma unroll
      for (int q = 0; q < 8; ++q) acc.v[q] = fmaf(wgt, g.v[q], acc.v[q]);
    }
  }
  store8<T>(dsrc + (((size_t)n * h + yi) * w + xi) * lddsrc + c, acc);
}

// ---- adjoint of an upsampling (scale <= 1 in both directions: every decoder resize of the models) ----------------------
// (A forward kernel with a 2x2 block of destination pixels per thread -- 9 loads for 4 outputs instead of 16 -- measured
//  7 % SLOWER than resize_fwd_kernel: the forward is bound by its 4x larger write stream, not by its loads.)
// (Measured on top of the row-wise loads below: also skipping the zero-weight ROWS of an accumulator +9 % time -- kept only the
//  column test.  Round 3, same-box: the column test as a select on the data instead of an exec-masked branch +17 % time; three
//  waves per SIMD instead of two (launch bounds: 168 registers) +17 %; both together 5.9x -- scripts/resize_bench.py.)
// (Also measured and dropped: an LDS-tiled adjoint -- 16x8 source pixels x 4 channel vectors per workgroup, the destination
//  region and the per-row / per-column (index, weight) tables staged in LDS once -- ran 1.6x SLOWER than this kernel at
//  C = 128..512 (64-byte pieces of every pixel row per workgroup), equal at C = 1024.)
// Backward: one thread = a 2x2 block of SOURCE pixels x 8 channels; every destination pixel of the union window is loaded
// once and feeds up to four accumulators (9 loads per output instead of 16 at scale 1/2); per source pixel the
// accumulation order (destination rows, then columns) and the weights are those of resize_bwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd2_kernel(const T* __restrict__ ddst, int ldddst, int choff, int H, int W,
                                                          T* __restrict__ dsrc, int lddsrc, int h, int w, int C8) {
  const int nv = C8 >> 3, wc = (w + 1) >> 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= wc * nv) return;
  const int xb = idx / nv, c = (idx - xb * nv) * 8;
  const int yi0 = 2 * blockIdx.y, xi0 = 2 * xb, n = blockIdx.z;
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  int ja = 0, jb = H - 1, ka = 0, kb = W - 1;
  if (sy > 0.f) {
    ja = max(0, (int)floorf(((float)yi0 - 1.f) / sy) - 1);
    jb = min(H - 1, (int)ceilf(((float)yi0 + 2.f) / sy) + 1);
  }
  if (sx > 0.f) {
    ka = max(0, (int)floorf(((float)xi0 - 1.f) / sx) - 1);
    kb = min(W - 1, (int)ceilf(((float)xi0 + 2.f) / sx) + 1);
  }
  F8 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) acc[a][bq] = zero8();
  // Window of at most KW destination columns (an upsampling by ~2): the column weights are computed once, and every row of
  // the window issues its KW loads TOGETHER and unconditionally (clamped column, weight 0 outside the hits) -- with the loads
  // behind per-pixel hit tests the kernel was a chain of dependent load -> use round trips (2.4-2.9 TB/s of its traffic).
  // (Zero-weight columns and rows are skipped: same sums, same order as resize_bwd_kernel.)
  constexpr int KW = 8;
  // tighten the loose window to the destination rows / columns whose y0 (x0) lies in [yi0 - 1, yi0 + 1] ([xi0 - 1, xi0 + 1]):
  // y0 is monotone in j, so these are one run; a pixel outside it cannot touch the 2x2 block
  {
    int i0, i1;
    float l0, l1;
    while (ja < jb && (ac_src(sy, ja, h, i0, i1, l0, l1), i0 < yi0 - 1)) ++ja;
    while (jb > ja && (ac_src(sy, jb, h, i0, i1, l0, l1), i0 > yi0 + 1)) --jb;
    while (ka < kb && (ac_src(sx, ka, w, i0, i1, l0, l1), i0 < xi0 - 1)) ++ka;
    while (kb > ka && (ac_src(sx, kb, w, i0, i1, l0, l1), i0 > xi0 + 1)) --kb;
  }
  if (kb - ka + 1 <= KW) {
    float wxs[KW][2];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      wxs[kk][0] = wxs[kk][1] = 0.f;
      if (ka + kk <= kb) {
        int x0, x1;
        float lx0, lx1;
        ac_src(sx, ka + kk, w, x0, x1, lx0, lx1);
#pragma unroll
        for (int bq = 0; bq < 2; ++bq) wxs[kk][bq] = (x0 == xi0 + bq ? lx0 : 0.f) + (x1 == xi0 + bq ? lx1 : 0.f);
      }
    }
    for (int j = ja; j <= jb; j += 2) {                  // two rows per iteration: 2 * KW loads in flight
      float wy[2][2];
      const T* rowp[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int jj = min(j + rr, jb);
        int y0, y1;
        float ly0, ly1;
        ac_src(sy, jj, h, y0, y1, ly0, ly1);
        const bool live = j + rr <= jb;
#pragma unroll
        for (int a = 0; a < 2; ++a) wy[rr][a] = live ? (y0 == yi0 + a ? ly0 : 0.f) + (y1 == yi0 + a ? ly1 : 0.f) : 0.f;
        rowp[rr] = ddst + ((size_t)n * H + jj) * W * ldddst + choff + c;
      }
      F8 g[2][KW];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) g[rr][kk] = load8<T>(rowp[rr] + (size_t)min(ka + kk, W - 1) * ldddst);
      // A destination row reaches one or two of the block's rows; its weight for the other is exactly 0, and 0 * Inf would put a
      // NaN into a source pixel that never read that destination pixel (fp16 training without loss scaling can overflow a
      // gradient; ATen's adjoint only propagates non-finite values to true contributors).  The row weights depend on
      // (blockIdx.y, j) only -- wave-uniform -- so the test is a scalar branch (readfirstlane), not an exec mask.
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (__builtin_amdgcn_readfirstlane(__float_as_int(wy[rr][a])) == 0) continue;
#pragma unroll
          for (int kk = 0; kk < KW; ++kk)
#pragma unroll
            for (int bq = 0; bq < 2; ++bq)
              if (wxs[kk][bq] != 0.f) {                    // a column reaches one or two of the block's columns, not all
                const float wgt = wy[rr][a] * wxs[kk][bq];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[a][bq].v[q] = fmaf(wgt, g[rr][kk].v[q], acc[a][bq].v[q]);
              }
        }
    }
  } else
  for (int j = ja; j <= jb; ++j) {
    int y0, y1;
    float ly0, ly1;
    ac_src(sy, j, h, y0, y1, ly0, ly1);
    float wy[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) wy[a] = (y0 == yi0 + a ? ly0 : 0.f) + (y1 == yi0 + a ? ly1 : 0.f);
    const bool hit_y0 = (y0 == yi0) | (y1 == yi0), hit_y1 = (y0 == yi0 + 1) | (y1 == yi0 + 1);
    if (!(hit_y0 | hit_y1)) continue;
    for (int k = ka; k <= kb; ++k) {
      int x0, x1;
      float lx0, lx1;
      ac_src(sx, k, w, x0, x1, lx0, lx1);
      float wx[2];
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) wx[bq] = (x0 == xi0 + bq ? lx0 : 0.f) + (x1 == xi0 + bq ? lx1 : 0.f);
      const bool hit_x0 = (x0 == xi0) | (x1 == xi0), hit_x1 = (x0 == xi0 + 1) | (x1 == xi0 + 1);
      if (!(hit_x0 | hit_x1)) continue;
      const F8 g = load8<T>(ddst + (((size_t)n * H + j) * W + k) * ldddst + choff + c);
      const bool hy[2] = {hit_y0, hit_y1}, hx[2] = {hit_x0, hit_x1};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bq = 0; bq < 2; ++bq)
          if (hy[a] & hx[bq]) {
            const float wgt = wy[a] * wx[bq];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[a][bq].v[q] = fmaf(wgt, g.v[q], acc[a][bq].v[q]);
          }
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq)
      if (yi0 + a < h && xi0 + bq < w) store8<T>(dsrc + (((size_t)n * h + yi0 + a) * w + xi0 + bq) * lddsrc + c, acc[a][bq]);
}

// dst[..., choff + c] = src[..., c] for c < C (element granularity: tolerates any C / choff), then
// zero-fill dst channels [choff + C, zero_to)
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const T* __restrict__ src, int ldsrc, T* __restrict__ dst, int lddst,
                                                            int choff, int zero_to, int64_t npix, int C) {
  const int span = (zero_to > choff + C ? zero_to : choff + C) - choff;
  const int64_t total = npix * span;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = idx / span;
    const int c = (int)(idx % span);
    dst[pix * lddst + choff + c] = c < C ? src[pix * ldsrc + c] : (T)0.f;
  }
}
// vector path: a thread owns one 8-channel vector column and walks pixels (no div/mod per element)
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_vec_kernel(const T* __restrict__ src, int ldsrc, T* __restrict__ dst,
                                                                int lddst, int choff, int64_t npix, int C8, int pixb) {
  const int nv = C8 >> 3;
  const int nvl = nv < 256 ? nv : 256, PS = 256 / nvl;
  const int v = threadIdx.x % nvl, ps = threadIdx.x / nvl;
  if (ps >= PS) return;
  const int64_t p0 = (int64_t)blockIdx.x * pixb, p1 = p0 + pixb < npix ? p0 + pixb : npix;
  for (int vv = v; vv < nv; vv += nvl) {
    const int c = vv * 8;
    int64_t pix = p0 + ps;
    for (; pix + 3 * PS < p1; pix += 4 * PS) {
      F8 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = load8<T>(src + (pix + u * PS) * ldsrc + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) store8<T>(dst + (pix + u * PS) * lddst + choff + c, t[u]);
    }
    for (; pix < p1; pix += PS) store8<T>(dst + pix * lddst + choff + c, load8<T>(src + pix * ldsrc + c));
  }
}

// dst[p][c] = src_0[p][c] + src_1[p][c] + ... (k <= SUM_MAX_SRC NHWC-ld tensors, each with its own pixel pitch): the gradients an
// activation with several readers receives (U-Net++ row slots, src/model.py:136-177).  One pass, fp32 sums in the order given, one
// rounding -- instead of k - 1 generic strided torch adds (3 tensors of traffic each, a rounding per add).
constexpr int SUM_MAX_SRC = 8;
struct SumSrcs {
  const void* p[SUM_MAX_SRC];
  int ld[SUM_MAX_SRC];
};
template <typename T>
__global__ __launch_bounds__(256) void sum_tensors_kernel(SumSrcs srcs, int k, T* __restrict__ dst, int lddst, int64_t npix, int C8, int pixb) {
  const int nv = C8 >> 3;
  const int nvl = nv < 256 ? nv : 256, PS = 256 / nvl;
  const int v = threadIdx.x % nvl, ps = threadIdx.x / nvl;
  if (ps >= PS) return;
  const int64_t p0 = (int64_t)blockIdx.x * pixb, p1 = p0 + pixb < npix ? p0 + pixb : npix;
  for (int vv = v; vv < nv; vv += nvl) {
    const int c = vv * 8;
    for (int64_t pix = p0 + ps; pix < p1; pix += PS) {
      F8 t[SUM_MAX_SRC];
#pragma unroll
      for (int i = 0; i < SUM_MAX_SRC; ++i)
        if (i < k) t[i] = load8<T>(reinterpret_cast<const T*>(srcs.p[i]) + pix * srcs.ld[i] + c);
      F8 acc = t[0];
#pragma unroll
      for (int i = 1; i < SUM_MAX_SRC; ++i)
        if (i < k) {
#pragma unroll
          for (int q = 0; q < 8; ++q) acc.v[q] += t[i].v[q];
        }
      store8<T>(dst + pix * lddst + c, acc);
    }
  }
}

// dst[n, p, choff + e] = emb[n][e] for every pixel p (materialised broadcast; only used when the channel
// counts do not fit the 8-channel granularity of the fused loader), then zero-fill up to zero_to
template <typename T>
__global__ __launch_bounds__(256) void bcast_fill_kernel(const float* __restrict__ emb, T* __restrict__ dst, int lddst, int choff,
                                                         int zero_to, int HW, int E, int64_t npix) {
  const int span = (zero_to > choff + E ? zero_to : choff + E) - choff;
  const int64_t total = npix * span;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = idx / span;
    const int c = (int)(idx % span);
    dst[pix * lddst + choff + c] = c < E ? (T)emb[(pix / HW) * E + c] : (T)0.f;
  }
}

// demb[n][e] = sum over the HW pixels of dx[n, :, choff + e]  (adjoint of the embedding broadcast).
// Level 1: workgroup (64-channel group, image n, pixel chunk) -- 8 channel vectors x 32 pixel slots,
// 16-byte loads, 2 pixels in flight per lane -- writes a partial row [n][chunk][E]; level 2 adds the
// chunks in fixed order (deterministic, no atomics).  U-Net++ broadcasts the embedding into every
// decoder node up to 256x256 (src/model.py:136-177), so H*W reaches 65,536 pixels per image.
constexpr int BCAST_PIX_PER_CHUNK = 2048;

template <typename T>
__global__ __launch_bounds__(256) void bcast_bwd_partial_kernel(const T* __restrict__ dx, int lddx, int choff,
                                                                float* __restrict__ part, int HW, int E, int nchunks) {
  __shared__ float red[32][64 + 1];
  const int cv = threadIdx.x & 7, ps = threadIdx.x >> 3;
  const int n = blockIdx.y, chunk = blockIdx.z;
  const int e0 = blockIdx.x * 64 + cv * 8;
  const int p0 = chunk * BCAST_PIX_PER_CHUNK, p1 = min(HW, p0 + BCAST_PIX_PER_CHUNK);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (e0 < E) {
    const T* base = dx + (size_t)n * HW * lddx + choff + e0;
    int p = p0 + ps;
    for (; p + 32 < p1; p += 64) {
      const F8 a = load8<T>(base + (size_t)p * lddx), b = load8<T>(base + (size_t)(p + 32) * lddx);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += a.v[j] + b.v[j];
    }
    for (; p < p1; p += 32) {
      const F8 a = load8<T>(base + (size_t)p * lddx);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += a.v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ps][cv * 8 + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    float t = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
    if (e < E) part[((size_t)n * nchunks + chunk) * E + e] = t;
  }
}

__global__ void bcast_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ demb, int NE, int E, int nchunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NE) return;
  const int n = i / E, e = i % E;
  float t = 0.f;
  for (int c = 0; c < nchunks; ++c) t += part[((size_t)n * nchunks + c) * E + e];
  demb[i] = t;
}

// element-granular fallback (channel offset / count not a multiple of 8: tiny test models only)
template <typename T>
__global__ __launch_bounds__(256) void bcast_bwd_kernel(const T* __restrict__ dx, int lddx, int choff, float* __restrict__ demb,
                                                        int HW, int E) {
  __shared__ float red[4][64];
  const int n = blockIdx.y, e = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float s = 0.f;
  if (e < E)
    for (int p = rl; p < HW; p += 4) s += (float)dx[((size_t)n * HW + p) * lddx + choff + e];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && e < E) demb[(size_t)n * E + e] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

}  // namespace mau

using namespace mau;

extern "C" {

int mau_nchw_to_nhwc(const float* src, void* dst, int dtype, int N, int C, int H, int W, int ld, mau_stream_t stream) {
  MAU_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld % 8 == 0 && ld >= C, "nchw_to_nhwc: bad arguments");
  const int64_t HW = (int64_t)H * W;
  dim3 grid(ceil_div(HW, 256), ceil_div(round_up(C, 8), 8), N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(nchw_to_nhwc_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, src, (T*)dst, C, HW, ld));
  return check_launch("nchw_to_nhwc_kernel");
}

int mau_nhwc_to_nchw(const void* src, float* dst, int dtype, int N, int C, int H, int W, int ld, mau_stream_t stream) {
  MAU_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld % 8 == 0 && ld >= C, "nhwc_to_nchw: bad arguments");
  const int64_t HW = (int64_t)H * W;
  dim3 grid(ceil_div(HW, 256), ceil_div(C, 8), N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(nhwc_to_nchw_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)src, dst, C, HW, ld));
  return check_launch("nhwc_to_nchw_kernel");
}

int mau_pack_tile_onehot(const unsigned char* cls_a, const unsigned char* cls_b, const float* cont, const unsigned char* flip,
                         void* out, int ldo, int dtype, int N, int H, int W, int num_classes, int ncont, mau_stream_t stream) {
  MAU_REQUIRE(cls_a && cls_b && out && N > 0 && H > 0 && W > 0, "pack_tile_onehot: bad arguments");
  MAU_REQUIRE(num_classes >= 1 && num_classes <= 255 && ncont >= 0 && ncont <= 8 && (ncont == 0 || cont), "pack_tile_onehot: bad channel counts");
  MAU_REQUIRE(ldo % 8 == 0 && ldo >= 2 * num_classes + ncont, "pack_tile_onehot: bad ld");
  MAU_REQUIRE(H <= 65535 && N <= 65535, "pack_tile_onehot: H and N must fit a grid dimension");
  dim3 grid(ceil_div(W, 256), H, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(pack_tile_onehot_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, cls_a, cls_b, cont, flip,
                                       (T*)out, ldo, H, W, num_classes, ncont));
  return check_launch("pack_tile_onehot_kernel");
}

int mau_flip_rows(const float* src, float* dst, const unsigned char* flip, int N, int C, int H, int W, mau_stream_t stream) {
  MAU_REQUIRE(src && dst && flip && src != dst && N > 0 && C > 0 && H > 0 && W > 0, "flip_rows: bad arguments");
  MAU_REQUIRE((int64_t)C * H <= 65535 && N <= 65535, "flip_rows: C*H and N must fit a grid dimension");
  dim3 grid(ceil_div(W, 256), C * H, N);
  MAU_LAUNCH(flip_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, flip, C * H, W);
  return check_launch("flip_rows_kernel");
}

int mau_maxpool2x2_fwd(const void* x, int ldx, void* y, int ldy, int dtype, int N, int H, int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(x && y && N > 0 && H >= 2 && W >= 2 && C > 0, "maxpool2x2_fwd: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C8 && ldy >= C8, "maxpool2x2_fwd: bad ld");
  MAU_REQUIRE(H / 2 <= 65535 && N <= 65535, "maxpool2x2_fwd: H/2 and N must fit a grid dimension");
  dim3 grid(ceil_div((W / 2) * (C8 / 8), 256), H / 2, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(maxpool_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, H, W, C8));
  return check_launch("maxpool_fwd_kernel");
}

int mau_maxpool2x2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int dtype, int N, int H, int W,
                       int C, mau_stream_t stream) {
  MAU_REQUIRE(x && dy && dx && N > 0 && H >= 2 && W >= 2 && C > 0, "maxpool2x2_bwd: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ldx >= C8 && lddy >= C8 && lddx >= C8, "maxpool2x2_bwd: bad ld");
  MAU_REQUIRE(H <= 65535 && N <= 65535, "maxpool2x2_bwd: H and N must fit a grid dimension");
  dim3 grid(ceil_div(W * (C8 / 8), 256), H, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(maxpool_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)dy, lddy,
                                       (const T*)nullptr, 0, (T*)dx, lddx, H, W, C8));
  return check_launch("maxpool_bwd_kernel");
}

int mau_maxpool2x2_bwd_add(const void* x, int ldx, const void* dy, int lddy, const void* dskip, int lddskip, void* dx, int lddx,
                           int dtype, int N, int H, int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(x && dy && dskip && dx && N > 0 && H >= 2 && W >= 2 && C > 0, "maxpool2x2_bwd_add: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && lddskip % 8 == 0 && ldx >= C8 && lddy >= C8 && lddx >= C8 && lddskip >= C8,
              "maxpool2x2_bwd_add: bad ld");
  MAU_REQUIRE(H <= 65535 && N <= 65535, "maxpool2x2_bwd_add: H and N must fit a grid dimension");
  dim3 grid(ceil_div(W * (C8 / 8), 256), H, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(maxpool_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)dy, lddy,
                                       (const T*)dskip, lddskip, (T*)dx, lddx, H, W, C8));
  return check_launch("maxpool_bwd_kernel");
}

// Largest number of destination columns that share one left source column (host twin of ac_scale / ac_src: the same IEEE
// float operations): resize_rows_kernel holds at most 3 per thread.
static int resize_max_cols_per_cell(int in, int out) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  int best = 0, run = 0, prev = -1;
  for (int k = 0; k < out; ++k) {
    const float sf = scale * (float)k;
    int i0 = (int)sf;
    if (i0 > in - 1) i0 = in - 1;
    run = i0 == prev ? run + 1 : 1;
    prev = i0;
    if (run > best) best = run;
  }
  return best;
}
static bool resize_rows_ok(int h, int w, int H, int W) {
  return h <= H && w <= W && h <= 65535 && resize_max_cols_per_cell(w, W) <= 3;
}

int mau_resize_bilinear_fwd(const void* src, int ldsrc, int h, int w, void* dst, int lddst, int choff, int dtype, int N,
                            int H, int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(src && dst && N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "resize_bilinear_fwd: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldsrc % 8 == 0 && lddst % 8 == 0 && choff % 8 == 0 && ldsrc >= C8 && lddst >= choff + C8, "resize_bilinear_fwd: bad ld/choff");
  MAU_REQUIRE(H <= 65535 && N <= 65535, "resize_bilinear_fwd: H and N must fit a grid dimension");
  if (resize_rows_ok(h, w, H, W)) {                    // upsampling by <= ~2: a column of source cells per thread
    const int xb = ceil_div(w * (C8 / 8), 256);
    int RY = 8;
    while (RY > 1 && (int64_t)xb * ceil_div(h, RY) * N < 1024) RY >>= 1;
    dim3 gridr(xb, ceil_div(h, RY), N);
    MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((resize_rows_kernel<T, false>), gridr, dim3(256), 0, (hipStream_t)stream, (const T*)src, ldsrc, h, w, (T*)dst, lddst, choff,
                                         H, W, C8, (const float*)nullptr, (const float*)nullptr, C, RY));
    return check_launch("resize_rows_kernel");
  }
  if (h <= H && w <= W && h <= 65535) {                // any other upsampling: one thread per source cell
    dim3 gridc(ceil_div(w * (C8 / 8), 256), h, N);
    MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((resize_fwd_cell_kernel<T, false>), gridc, dim3(256), 0, (hipStream_t)stream, (const T*)src, ldsrc, h, w, (T*)dst, lddst, choff, H, W, C8,
                                         (const float*)nullptr, (const float*)nullptr, C));
    return check_launch("resize_fwd_cell_kernel");
  }
  dim3 grid(ceil_div(W * (C8 / 8), 256), H, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(resize_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)src, ldsrc, h, w, (T*)dst, lddst, choff, H, W, C8));
  return check_launch("resize_fwd_kernel");
}

int mau_resize_bilinear_bn_fwd(const void* y, int ldy, int h, int w, const float* scale, const float* shift, void* dst, int lddst,
                               int choff, int dtype, int N, int H, int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(y && dst && scale && shift && N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "resize_bilinear_bn_fwd: bad arguments");
  MAU_REQUIRE(h <= H && w <= W && h <= 65535 && N <= 65535, "resize_bilinear_bn_fwd: an upsampling (h <= H, w <= W), h and N within a grid dimension");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && lddst % 8 == 0 && choff % 8 == 0 && ldy >= C8 && lddst >= choff + C8, "resize_bilinear_bn_fwd: bad ld/choff");
  if (resize_rows_ok(h, w, H, W)) {                    // upsampling by <= ~2: a column of source cells per thread
    const int xb = ceil_div(w * (C8 / 8), 256);
    int RY = 8;
    while (RY > 1 && (int64_t)xb * ceil_div(h, RY) * N < 1024) RY >>= 1;
    dim3 gridr(xb, ceil_div(h, RY), N);
    MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((resize_rows_kernel<T, true>), gridr, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, h, w, (T*)dst, lddst, choff, H,
                                         W, C8, scale, shift, C, RY));
    return check_launch("resize_rows_kernel<BN>");
  }
  dim3 gridc(ceil_div(w * (C8 / 8), 256), h, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((resize_fwd_cell_kernel<T, true>), gridc, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, h, w, (T*)dst, lddst,
                                       choff, H, W, C8, scale, shift, C));
  return check_launch("resize_fwd_cell_kernel<BN>");
}

int mau_resize_bilinear_bwd(const void* ddst, int ldddst, int choff, int H, int W, void* dsrc, int lddsrc, int dtype, int N,
                            int h, int w, int C, mau_stream_t stream) {
  MAU_REQUIRE(ddst && dsrc && N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "resize_bilinear_bwd: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldddst % 8 == 0 && lddsrc % 8 == 0 && choff % 8 == 0 && lddsrc >= C8 && ldddst >= choff + C8, "resize_bilinear_bwd: bad ld/choff");
  MAU_REQUIRE(h <= 65535 && N <= 65535, "resize_bilinear_bwd: h and N must fit a grid dimension");
  if (h <= H && w <= W && h >= 2 && w >= 2) {          // adjoint of an upsampling: 2x2 source pixels per thread
    dim3 grid2(ceil_div(((w + 1) / 2) * (C8 / 8), 256), (h + 1) / 2, N);
    MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(resize_bwd2_kernel<T>, grid2, dim3(256), 0, (hipStream_t)stream, (const T*)ddst, ldddst, choff, H, W, (T*)dsrc, lddsrc, h, w, C8));
    return check_launch("resize_bwd2_kernel");
  }
  dim3 grid(ceil_div(w * (C8 / 8), 256), h, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(resize_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)ddst, ldddst, choff, H, W, (T*)dsrc, lddsrc, h, w, C8));
  return check_launch("resize_bwd_kernel");
}

int mau_copy_channels(const void* src, int ldsrc, void* dst, int lddst, int choff, int zero_to, int dtype, int64_t npix,
                      int C, mau_stream_t stream) {
  MAU_REQUIRE(src && dst && npix > 0 && C > 0 && ldsrc >= C && lddst >= choff + C && zero_to <= lddst, "copy_channels: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (C % 8 == 0) && (choff % 8 == 0) && (ldsrc % 8 == 0) && (lddst % 8 == 0) && zero_to <= choff + C;
  if (vec) {
    const int nv = C / 8, nvl = nv < 256 ? nv : 256;
    const int pixb = (256 / nvl) * 8;
    MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(copy_channels_vec_kernel<T>, dim3(ceil_div(npix, pixb)), dim3(256), 0, st, (const T*)src, ldsrc, (T*)dst, lddst, choff, npix, C, pixb));
  } else {
    const int span = (zero_to > choff + C ? zero_to : choff + C) - choff;
    const int grid = stream_grid(npix * span, 256);
    MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(copy_channels_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)src, ldsrc, (T*)dst, lddst, choff, zero_to, npix, C));
  }
  return check_launch("copy_channels_kernel");
}

int mau_sum_tensors_max(void) { return SUM_MAX_SRC; }

int mau_sum_tensors(const void* const* srcs, const int* lds, int k, void* dst, int lddst, int dtype, int64_t npix, int C,
                    mau_stream_t stream) {
  MAU_REQUIRE(srcs && lds && dst && k >= 1 && k <= SUM_MAX_SRC && npix > 0 && C > 0, "sum_tensors: 1..%d sources", SUM_MAX_SRC);
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(lddst % 8 == 0 && lddst >= C8 && ((uintptr_t)dst % 16) == 0, "sum_tensors: bad destination ld / alignment");
  SumSrcs a;
  for (int i = 0; i < SUM_MAX_SRC; ++i) {
    a.p[i] = i < k ? srcs[i] : srcs[0];
    a.ld[i] = i < k ? lds[i] : lds[0];
    MAU_REQUIRE(i >= k || (srcs[i] && lds[i] % 8 == 0 && lds[i] >= C8 && ((uintptr_t)srcs[i] % 16) == 0), "sum_tensors: bad source %d", i);
  }
  const int nv = C8 / 8, nvl = nv < 256 ? nv : 256;
  const int pixb = (256 / nvl) * 8;
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(sum_tensors_kernel<T>, dim3(ceil_div(npix, pixb)), dim3(256), 0, (hipStream_t)stream, a, k, (T*)dst, lddst, npix, C8, pixb));
  return check_launch("sum_tensors_kernel");
}

int mau_bcast_fill(const float* emb, void* dst, int lddst, int choff, int zero_to, int dtype, int N, int HW, int E,
                   mau_stream_t stream) {
  MAU_REQUIRE(emb && dst && N > 0 && HW > 0 && E > 0 && lddst >= choff + E && zero_to <= lddst, "bcast_fill: bad arguments");
  const int span = (zero_to > choff + E ? zero_to : choff + E) - choff;
  const int64_t npix = (int64_t)N * HW;
  const int grid = stream_grid(npix * span, 256);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(bcast_fill_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, emb, (T*)dst, lddst, choff, zero_to, HW, E, npix));
  return check_launch("bcast_fill_kernel");
}

size_t mau_bcast_bwd_ws_elems(int N, int HW, int E) { return (size_t)N * ceil_div(HW, BCAST_PIX_PER_CHUNK) * E; }

int mau_bcast_bwd(const void* dx, int lddx, int choff, float* demb, float* ws, int dtype, int N, int HW, int E,
                  mau_stream_t stream) {
  MAU_REQUIRE(dx && demb && N > 0 && HW > 0 && E > 0 && lddx >= choff + E, "bcast_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = ws != nullptr && choff % 8 == 0 && E % 8 == 0 && lddx % 8 == 0 && ((uintptr_t)dx % 16) == 0;
  if (!vec) {
    dim3 grid(ceil_div(E, 64), N);
    MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(bcast_bwd_kernel<T>, grid, dim3(256), 0, st, (const T*)dx, lddx, choff, demb, HW, E));
    return check_launch("bcast_bwd_kernel");
  }
  const int nchunks = ceil_div(HW, BCAST_PIX_PER_CHUNK);
  dim3 grid(ceil_div(E, 64), N, nchunks);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(bcast_bwd_partial_kernel<T>, grid, dim3(256), 0, st, (const T*)dx, lddx, choff, ws, HW, E, nchunks));
  MAU_LAUNCH(bcast_bwd_final_kernel, dim3(ceil_div(N * E, 256)), dim3(256), 0, st, ws, demb, N * E, E, nchunks);
  return check_launch("bcast_bwd_partial_kernel");
}

}  // extern "C"
