// bn_fused.hip -- BatchNorm2d + ReLU fused with the streaming operator on its far side, so that a full-resolution tensor is
// never written only to be read once (reference src/model.py:13-16 with :218/:268-271 (pool), :241/:284-292 (head)).
//
// Backward of an encoder block's second BatchNorm (its output feeds nn.MaxPool2d AND a skip connection):
//     unfused:  maxpool2x2_bwd_add (read a, dpl, dskip; write da) -> bn_relu_bwd_reduce (read da, y) -> bn_relu_bwd_apply (read da, y; write dy)
//     fused:    pool_bn_bwd_reduce (read y, dpl, dskip)           ->                                   pool_bn_bwd_apply (read y, dpl, dskip; write dy)
//   da = dskip + scatter(dpl to the first maximum of its 2x2 window of a) is recomputed in registers both times; the
//   activation a = relu(scale*y + shift) it needs for the arg-max is recomputed from y (rounded to the activation type,
//   exactly what bn_relu_apply_pool stored).  8.25 S -> 5.5 S bytes for an activation of S bytes, one launch fewer.
// Head (final 1x1 conv + tanh) behind the last block's second BatchNorm:
//     unfused:  bn_relu_apply (y -> a), head_fwd (a -> out);  head_bwd (a, dout -> da, dW, db), bn_relu_bwd_reduce, bn_relu_bwd_apply
//     fused:    head_bn_fwd (y -> out);  head_bn_bwd_reduce (y, dout -> BatchNorm sums, dW, db), head_bn_bwd_apply (y, dout -> dy)
//   10 S -> 4 S bytes, two launches fewer.
// The reduce kernels keep the workgroup geometry and per-thread accumulation order of bn_relu_bwd_reduce_kernel (and of
// head_bwd_kernel): (8 channel vectors) x (32 pixel slots, stride 32) over BWD_PIX_PER_BLOCK consecutive pixels, and they round
// the recomputed da / a to the activation type as the unfused path's stores do -- their sums are BIT-IDENTICAL to the unfused
// path's (tests/test_gpu_ops.py::test_fused_bn_backward_matches_unfused).
#include "mau_common.h"

namespace mau {

template <typename T>
__device__ __forceinline__ float round_to(float v) {
  return (float)(T)v;
}

// coefficients of 8 channels starting at c0 (zeros beyond C)
struct Coef8 {
  float sc[8], sh[8], mu[8], is[8];
  __device__ __forceinline__ void load(const float* scale, const float* shift, const float* mean, const float* invstd, int c0, int C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = c0 + j < C;
      sc[j] = ok ? scale[c0 + j] : 0.f;
      sh[j] = ok ? shift[c0 + j] : 0.f;
      mu[j] = (ok && mean) ? mean[c0 + j] : 0.f;
      is[j] = (ok && invstd) ? invstd[c0 + j] : 0.f;
    }
  }
};

// gradient arriving at pixel (n, yi, xi) of an activation that feeds MaxPool2d(2,2) and a skip connection:
//   dskip[pixel] + (dpl[window] if this pixel is the window's first maximum), rounded to T as maxpool_bwd_kernel stores it.
// yv: this pixel's raw conv output (already loaded).  Window activations are recomputed from y and rounded to T
// (bn_relu_apply_pool_kernel's stored values), scan order and strict '>' as ATen's max_pool2d.
template <typename T>
__device__ __forceinline__ F8 pool_grad(const T* __restrict__ y, int ldy, const T* __restrict__ dpl, int lddpl, const T* __restrict__ dskip,
                                        int lddskip, const Coef8& k, int n, int yi, int xi, int H, int W, int c0) {
  const int Ho = H >> 1, Wo = W >> 1, yo = yi >> 1, xo = xi >> 1;
  F8 o = zero8();
  if (dpl != nullptr && yo < Ho && xo < Wo) {
    const T* b = y + (((size_t)n * H + 2 * yo) * W + 2 * xo) * ldy + c0;
    const F8 v00 = load8<T>(b), v01 = load8<T>(b + ldy), v10 = load8<T>(b + (size_t)W * ldy), v11 = load8<T>(b + (size_t)W * ldy + ldy);
    const F8 g = load8<T>(dpl + (((size_t)n * Ho + yo) * Wo + xo) * lddpl + c0);
    const int me = (yi & 1) * 2 + (xi & 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a00 = round_to<T>(fmaxf(fmaf(v00.v[j], k.sc[j], k.sh[j]), 0.f)), a01 = round_to<T>(fmaxf(fmaf(v01.v[j], k.sc[j], k.sh[j]), 0.f));
      const float a10 = round_to<T>(fmaxf(fmaf(v10.v[j], k.sc[j], k.sh[j]), 0.f)), a11 = round_to<T>(fmaxf(fmaf(v11.v[j], k.sc[j], k.sh[j]), 0.f));
      int arg = 0;
      float m = a00;
      if (a01 > m) { m = a01; arg = 1; }
      if (a10 > m) { m = a10; arg = 2; }
      if (a11 > m) { m = a11; arg = 3; }
      o.v[j] = (arg == me) ? g.v[j] : 0.f;
    }
  }
  if (dskip != nullptr) {
    const F8 sk = load8<T>(dskip + (((size_t)n * H + yi) * W + xi) * lddskip + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] += sk.v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = round_to<T>(o.v[j]);
  return o;
}

// the LDS join of bn_relu_bwd_reduce_kernel: 32 pixel slots -> one slab row [2][ldslab] per workgroup
__device__ __forceinline__ void bn_sums_to_slab(float (*red)[32][64 + 1], const float* s1, const float* s2, int cv, int ps, float* slab,
                                                int ldslab) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][ps][cv * 8 + j] = s1[j];
    red[1][ps][cv * 8 + j] = s2[j];
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) s += red[which][r][c];
    const int cc = blockIdx.y * 64 + c;
    if (cc < ldslab) slab[((size_t)blockIdx.x * 2 + which) * ldslab + cc] = s;
  }
}

// ---- pool: backward pass 1 (partial sums of dz and dz * xhat) ----
template <typename T>
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dpl, int lddpl,
                                                                 const T* __restrict__ dskip, int lddskip, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* __restrict__ slab, int ldslab,
                                                                 int H, int W, int64_t npix, int C) {
  __shared__ float red[2][32][64 + 1];
  const int cv = threadIdx.x & 7, ps = threadIdx.x >> 3;
  const int c0 = blockIdx.y * 64 + cv * 8;
  const int64_t p0 = (int64_t)blockIdx.x * BWD_PIX_PER_BLOCK;
  const int64_t p1 = p0 + BWD_PIX_PER_BLOCK < npix ? p0 + BWD_PIX_PER_BLOCK : npix;
  Coef8 k;
  k.load(scale, shift, mean, invstd, c0, C);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (c0 < C) {
    int64_t p = p0 + ps;
    const int HW = H * W;
    int n = (int)(p / HW), r = (int)(p - (int64_t)n * HW);
    int yi = r / W, xi = r - yi * W;
    for (; p < p1; p += 32) {
      const F8 v = load8<T>(y + p * ldy + c0);
      const F8 g = pool_grad<T>(y, ldy, dpl, lddpl, dskip, lddskip, k, n, yi, xi, H, W, c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float act = fmaf(v.v[j], k.sc[j], k.sh[j]);
        const float dz = act > 0.f ? g.v[j] : 0.f;
        s1[j] += dz;
        s2[j] = fmaf(dz, (v.v[j] - k.mu[j]) * k.is[j], s2[j]);
      }
      xi += 32;
      while (xi >= W) {
        xi -= W;
        if (++yi == H) {
          yi = 0;
          ++n;
        }
      }
    }
  }
  bn_sums_to_slab(red, s1, s2, cv, ps, slab, ldslab);
}

// k0/k1 of dy = sc*dz - k0 - k1*y (bn_relu_bwd_apply_kernel's form)
struct Apply8 {
  float sc[8], sh[8], k0[8], k1[8];
  __device__ __forceinline__ void load(const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums,
                                       double inv_count, int c0, int C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cc = c0 + j;
      const bool ok = cc < C;
      sc[j] = ok ? scale[cc] : 0.f;
      sh[j] = ok ? shift[cc] : 0.f;
      const float mu = ok ? mean[cc] : 0.f, is = ok ? invstd[cc] : 0.f;
      const float m1 = ok ? (float)(sums[cc] * inv_count) : 0.f, m2 = ok ? (float)(sums[C + cc] * inv_count) : 0.f;
      k1[j] = sc[j] * m2 * is;
      k0[j] = sc[j] * m1 - k1[j] * mu;
    }
  }
  __device__ __forceinline__ float dy(int j, float yv, float da) const {
    const float act = fmaf(yv, sc[j], sh[j]);
    const float dz = act > 0.f ? da : 0.f;
    return fmaf(sc[j], dz, -fmaf(k1[j], yv, k0[j]));
  }
};

// ---- pool: backward pass 2; one thread = one 2x2 window x 8 channels (the four y vectors serve arg-max and dy alike) ----
template <typename T>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dpl, int lddpl,
                                                                const T* __restrict__ dskip, int lddskip, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const double* __restrict__ sums,
                                                                double inv_count, T* __restrict__ dyo, int lddy, int H, int W, int C, int C8) {
  const int nv = C8 >> 3, Wc = (W + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wc * nv) return;
  const int xo = idx / nv, c = (idx - xo * nv) * 8;
  const int yo = blockIdx.y, n = blockIdx.z;
  if (inv_count <= 0.0) inv_count = 1.0 / sums[2 * C];
  Apply8 k;
  k.load(scale, shift, mean, invstd, sums, inv_count, c, C);
  const int y0 = 2 * yo, x0 = 2 * xo;
  const bool hasx = x0 + 1 < W, hasy = y0 + 1 < H;
  const size_t row0 = ((size_t)n * H + y0) * W, row1 = row0 + W;
  F8 v[4];
  v[0] = load8<T>(y + (row0 + x0) * ldy + c);
  v[1] = hasx ? load8<T>(y + (row0 + x0 + 1) * ldy + c) : zero8();
  v[2] = hasy ? load8<T>(y + (row1 + x0) * ldy + c) : zero8();
  v[3] = (hasx && hasy) ? load8<T>(y + (row1 + x0 + 1) * ldy + c) : zero8();
  F8 d[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) d[u] = zero8();
  if (dskip != nullptr) {
    d[0] = load8<T>(dskip + (row0 + x0) * lddskip + c);
    if (hasx) d[1] = load8<T>(dskip + (row0 + x0 + 1) * lddskip + c);
    if (hasy) d[2] = load8<T>(dskip + (row1 + x0) * lddskip + c);
    if (hasx && hasy) d[3] = load8<T>(dskip + (row1 + x0 + 1) * lddskip + c);
  }
  if (dpl != nullptr && yo < Ho && xo < Wo) {             // (inside the pooled range all four pixels exist)
    const F8 g = load8<T>(dpl + (((size_t)n * Ho + yo) * Wo + xo) * lddpl + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = round_to<T>(fmaxf(fmaf(v[u].v[j], k.sc[j], k.sh[j]), 0.f));
      int arg = 0;
      float m = a[0];
      if (a[1] > m) { m = a[1]; arg = 1; }
      if (a[2] > m) { m = a[2]; arg = 2; }
      if (a[3] > m) { m = a[3]; arg = 3; }
#pragma unroll
      for (int u = 0; u < 4; ++u) d[u].v[j] = (arg == u ? g.v[j] : 0.f) + d[u].v[j];
    }
  }
  F8 o[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) o[u].v[j] = k.dy(j, v[u].v[j], round_to<T>(d[u].v[j]));
  store8<T>(dyo + (row0 + x0) * lddy + c, o[0]);
  if (hasx) store8<T>(dyo + (row0 + x0 + 1) * lddy + c, o[1]);
  if (hasy) store8<T>(dyo + (row1 + x0) * lddy + c, o[2]);
  if (hasx && hasy) store8<T>(dyo + (row1 + x0 + 1) * lddy + c, o[3]);
}

// ---- head: forward with BatchNorm + ReLU applied while the activation is loaded (head_fwd_kernel's mapping and arithmetic) ----
template <typename T>
__global__ __launch_bounds__(256) void head_bn_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out, int tanh0, int HW, int C,
                                                          int Co, int64_t npix) {
  const int sub = threadIdx.x & 7;
  float wr[HEAD_MAX_CO][8];
#pragma unroll
  for (int o = 0; o < HEAD_MAX_CO; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[o][j] = (o < Co && sub * 8 + j < C) ? w[o * C + sub * 8 + j] : 0.f;
  Coef8 k;
  k.load(scale, shift, nullptr, nullptr, sub * 8, C);
  const int64_t stride = (int64_t)gridDim.x * 32;
  for (int64_t pix0 = (int64_t)blockIdx.x * 32; pix0 < npix; pix0 += stride) {      // block-uniform trip count
    const int64_t pix = pix0 + (threadIdx.x >> 3);
    const bool live = pix < npix;
    float acc[HEAD_MAX_CO];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o) acc[o] = 0.f;
    if (live && sub * 8 < C) {
      const F8 x = load8<T>(y + pix * ldy + sub * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = round_to<T>(fmaxf(fmaf(x.v[j], k.sc[j], k.sh[j]), 0.f));
#pragma unroll
        for (int o = 0; o < HEAD_MAX_CO; ++o) acc[o] = fmaf(a, wr[o][j], acc[o]);
      }
    }
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o) {
      acc[o] += __shfl_xor(acc[o], 1);
      acc[o] += __shfl_xor(acc[o], 2);
      acc[o] += __shfl_xor(acc[o], 4);
    }
    if (live && sub < Co) {
      float r = 0.f;
#pragma unroll
      for (int o = 0; o < HEAD_MAX_CO; ++o)
        if (o == sub) r = acc[o];
      r += b[sub];
      if (tanh0 && sub == 0) r = tanhf(r);
      const int64_t n = pix / HW, q = pix - n * HW;
      out[((size_t)n * Co + sub) * HW + q] = r;
    }
  }
}

// dz of the head at pixel (n, q): dout * (1 - out^2) on the tanh channel
__device__ __forceinline__ void head_dz(const float* __restrict__ out, const float* __restrict__ dout, int tanh0, int Co, int HW, int64_t n,
                                        int q, float* dz) {
#pragma unroll
  for (int o = 0; o < HEAD_MAX_CO; ++o) {
    dz[o] = 0.f;
    if (o < Co) {
      const size_t oi = ((size_t)n * Co + o) * HW + q;
      float g = dout[oi];
      if (tanh0 && o == 0) {
        const float t = out[oi];
        g *= (1.f - t * t);
      }
      dz[o] = g;
    }
  }
}

// ---- head: backward pass 1.  One workgroup = BWD_PIX_PER_BLOCK pixels x all C <= 64 channels: BatchNorm partial sums (slab row
// as bn_relu_bwd_reduce_kernel) and the head's dW / db partials (slab row as head_bwd_kernel) in one pass over y ----
template <typename T>
__global__ __launch_bounds__(256) void head_bn_bwd_reduce_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ w,
                                                                 const float* __restrict__ out, const float* __restrict__ dout,
                                                                 float* __restrict__ bn_slab, int ldslab, float* __restrict__ head_slab,
                                                                 int tanh0, int HW, int C, int C8, int Co, int64_t npix) {
  __shared__ float red[2][32][64 + 1];
  __shared__ float hred[32 * HEAD_MAX_CO * 65];
  const int cv = threadIdx.x & 7, ps = threadIdx.x >> 3;
  const int c0 = cv * 8;
  const int64_t p0 = (int64_t)blockIdx.x * BWD_PIX_PER_BLOCK;
  const int64_t p1 = p0 + BWD_PIX_PER_BLOCK < npix ? p0 + BWD_PIX_PER_BLOCK : npix;
  const int rowlen = Co * (C8 + 8);
  float* row = head_slab + (size_t)blockIdx.x * rowlen;
  Coef8 k;
  k.load(scale, shift, mean, invstd, c0, C);
  float s1[8], s2[8], dwp[HEAD_MAX_CO][8], dbp[HEAD_MAX_CO], wr[HEAD_MAX_CO][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int o = 0; o < HEAD_MAX_CO; ++o) {
    dbp[o] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dwp[o][j] = 0.f;
      wr[o][j] = (o < Co && c0 + j < C) ? w[o * C + c0 + j] : 0.f;
    }
  }
  if (c0 < C8) {
    int64_t n = (p0 + ps) / HW;
    int q = (int)((p0 + ps) - n * HW);
    for (int64_t p = p0 + ps; p < p1; p += 32, q += 32) {
      while (q >= HW) {
        q -= HW;
        ++n;
      }
      float dz[HEAD_MAX_CO];
      head_dz(out, dout, tanh0, Co, HW, n, q, dz);
#pragma unroll
      for (int o = 0; o < HEAD_MAX_CO; ++o) dbp[o] += dz[o];
      const F8 v = load8<T>(y + p * ldy + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float act = fmaf(v.v[j], k.sc[j], k.sh[j]);
        const float a = round_to<T>(fmaxf(act, 0.f));                 // the activation bn_relu_apply stored
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < HEAD_MAX_CO; ++o)
          if (o < Co) {
            s = fmaf(dz[o], wr[o][j], s);
            dwp[o][j] = fmaf(dz[o], a, dwp[o][j]);
          }
        const float da = round_to<T>(s);                              // the gradient head_bwd stored
        const float dzb = act > 0.f ? da : 0.f;
        s1[j] += dzb;
        s2[j] = fmaf(dzb, (v.v[j] - k.mu[j]) * k.is[j], s2[j]);
      }
    }
  }
  // head partials (head_bwd_kernel's LDS join; one 64-channel group)
#pragma unroll
  for (int o = 0; o < HEAD_MAX_CO; ++o)
    if (o < Co) {
#pragma unroll
      for (int j = 0; j < 8; ++j) hred[(ps * HEAD_MAX_CO + o) * 65 + cv * 8 + j] = dwp[o][j];
    }
  __syncthreads();
  for (int t = threadIdx.x; t < Co * 64; t += 256) {
    const int o = t >> 6, c = t & 63;
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += hred[(r * HEAD_MAX_CO + o) * 65 + c];
    if (c < C8) row[o * (C8 + 8) + c] = s;
  }
  __syncthreads();
  if (cv == 0) {
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o)
      if (o < Co) hred[ps * HEAD_MAX_CO + o] = dbp[o];
  }
  __syncthreads();
  if (threadIdx.x < Co) {
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += hred[r * HEAD_MAX_CO + threadIdx.x];
    row[threadIdx.x * (C8 + 8) + C8] = s;
  }
  // BatchNorm partials
  bn_sums_to_slab(red, s1, s2, cv, ps, bn_slab, ldslab);
}

// ---- head: backward pass 2: dy of the conv in front of the BatchNorm, da recomputed from dout ----
template <typename T>
__global__ __launch_bounds__(256) void head_bn_bwd_apply_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const double* __restrict__ sums,
                                                                double inv_count, const float* __restrict__ w, const float* __restrict__ out,
                                                                const float* __restrict__ dout, T* __restrict__ dyo, int lddy, int tanh0,
                                                                int HW, int C, int Co, int64_t npix, int pixb) {
  const int cv = threadIdx.x & 7, ps = threadIdx.x >> 3;
  const int c0 = cv * 8;
  if (c0 >= ((C + 7) & ~7)) return;                        // (heads narrower than 64 channels; no barrier below)
  if (inv_count <= 0.0) inv_count = 1.0 / sums[2 * C];
  Apply8 k;
  k.load(scale, shift, mean, invstd, sums, inv_count, c0, C);
  float wr[HEAD_MAX_CO][8];
#pragma unroll
  for (int o = 0; o < HEAD_MAX_CO; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[o][j] = (o < Co && c0 + j < C) ? w[o * C + c0 + j] : 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * pixb;
  const int64_t p1 = p0 + pixb < npix ? p0 + pixb : npix;
  int64_t n = (p0 + ps) / HW;
  int q = (int)((p0 + ps) - n * HW);
  for (int64_t p = p0 + ps; p < p1; p += 32, q += 32) {
    while (q >= HW) {
      q -= HW;
      ++n;
    }
    float dz[HEAD_MAX_CO];
    head_dz(out, dout, tanh0, Co, HW, n, q, dz);
    const F8 v = load8<T>(y + p * ldy + c0);
    F8 o8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < HEAD_MAX_CO; ++o)
        if (o < Co) s = fmaf(dz[o], wr[o][j], s);
      o8.v[j] = k.dy(j, v.v[j], round_to<T>(s));
    }
    store8<T>(dyo + p * lddy + c0, o8);
  }
}

}  // namespace mau

using namespace mau;

extern "C" {

int mau_pool_bn_bwd_reduce(const void* y, int ldy, const void* dpl, int lddpl, const void* dskip, int lddskip, const float* scale,
                           const float* shift, const float* mean, const float* invstd, float* slab, int ldslab, int dtype, int N, int H,
                           int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(y && (dpl || dskip) && scale && shift && mean && invstd && slab && N > 0 && H >= 2 && W >= 2 && C > 0, "pool_bn_bwd_reduce: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= C8 && (!dpl || (lddpl % 8 == 0 && lddpl >= C8)) && (!dskip || (lddskip % 8 == 0 && lddskip >= C8)) && ldslab >= C,
              "pool_bn_bwd_reduce: bad ld");
  const int64_t npix = (int64_t)N * H * W;
  dim3 grid(ceil_div(npix, BWD_PIX_PER_BLOCK), ceil_div(C, 64));
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(pool_bn_bwd_reduce_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, (const T*)dpl,
                                       lddpl, (const T*)dskip, lddskip, scale, shift, mean, invstd, slab, ldslab, H, W, npix, C));
  return check_launch("pool_bn_bwd_reduce_kernel");
}

int mau_pool_bn_bwd_apply(const void* y, int ldy, const void* dpl, int lddpl, const void* dskip, int lddskip, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const double* sums, double count, void* dy, int lddy,
                          int dtype, int N, int H, int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(y && (dpl || dskip) && dy && sums && scale && shift && mean && invstd && N > 0 && H >= 2 && W >= 2 && C > 0 && count >= 0,
              "pool_bn_bwd_apply: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= C8 && lddy % 8 == 0 && lddy >= C8 && (!dpl || (lddpl % 8 == 0 && lddpl >= C8)) &&
                  (!dskip || (lddskip % 8 == 0 && lddskip >= C8)), "pool_bn_bwd_apply: bad ld");
  MAU_REQUIRE((H + 1) / 2 <= 65535 && N <= 65535, "pool_bn_bwd_apply: H/2 and N must fit a grid dimension");
  dim3 grid(ceil_div(((W + 1) / 2) * (C8 / 8), 256), (H + 1) / 2, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(pool_bn_bwd_apply_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, (const T*)dpl, lddpl,
                                       (const T*)dskip, lddskip, scale, shift, mean, invstd, sums, count > 0 ? 1.0 / count : 0.0, (T*)dy, lddy,
                                       H, W, C, C8));
  return check_launch("pool_bn_bwd_apply_kernel");
}

int mau_head_bn_max_channels(void) { return 64; }

int mau_head_bn_fwd(const void* y, int ldy, const float* scale, const float* shift, const float* w, const float* b, float* out, int tanh0,
                    int dtype, int N, int HW, int C, int Co, mau_stream_t stream) {
  MAU_REQUIRE(y && scale && shift && w && b && out && N > 0 && HW > 0 && C > 0, "head_bn_fwd: bad arguments");
  MAU_REQUIRE(Co >= 1 && Co <= HEAD_MAX_CO && C <= mau_head_bn_max_channels(), "head_bn_fwd: out_channels in [1,%d], C <= %d", HEAD_MAX_CO,
              mau_head_bn_max_channels());
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= round_up(C, 8), "head_bn_fwd: bad ld");
  const int64_t npix = (int64_t)N * HW;
  const int grid = stream_grid(npix * 8, 256);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(head_bn_fwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, scale, shift, w, b,
                                       out, tanh0, HW, C, Co, npix));
  return check_launch("head_bn_fwd_kernel");
}

int mau_head_bn_bwd_reduce(const void* y, int ldy, const float* scale, const float* shift, const float* mean, const float* invstd,
                           const float* w, const float* out, const float* dout, float* bn_slab, int ldslab, float* head_slab, int tanh0,
                           int dtype, int N, int HW, int C, int Co, mau_stream_t stream) {
  MAU_REQUIRE(y && scale && shift && mean && invstd && w && out && dout && bn_slab && head_slab && N > 0 && HW > 0 && C > 0,
              "head_bn_bwd_reduce: bad arguments");
  MAU_REQUIRE(Co >= 1 && Co <= HEAD_MAX_CO && C <= mau_head_bn_max_channels(), "head_bn_bwd_reduce: out_channels in [1,%d], C <= %d", HEAD_MAX_CO,
              mau_head_bn_max_channels());
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= C8 && ldslab >= C, "head_bn_bwd_reduce: bad ld");
  const int64_t npix = (int64_t)N * HW;
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(head_bn_bwd_reduce_kernel<T>, dim3(ceil_div(npix, BWD_PIX_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)y, ldy, scale, shift, mean, invstd, w, out, dout, bn_slab, ldslab, head_slab, tanh0, HW, C, C8,
                                       Co, npix));
  return check_launch("head_bn_bwd_reduce_kernel");
}

int mau_head_bn_bwd_apply(const void* y, int ldy, const float* scale, const float* shift, const float* mean, const float* invstd,
                          const double* sums, double count, const float* w, const float* out, const float* dout, void* dy, int lddy,
                          int tanh0, int dtype, int N, int HW, int C, int Co, mau_stream_t stream) {
  MAU_REQUIRE(y && scale && shift && mean && invstd && sums && w && out && dout && dy && N > 0 && HW > 0 && C > 0 && count >= 0,
              "head_bn_bwd_apply: bad arguments");
  MAU_REQUIRE(Co >= 1 && Co <= HEAD_MAX_CO && C <= mau_head_bn_max_channels(), "head_bn_bwd_apply: out_channels in [1,%d], C <= %d", HEAD_MAX_CO,
              mau_head_bn_max_channels());
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= C8 && lddy % 8 == 0 && lddy >= C8, "head_bn_bwd_apply: bad ld");
  const int64_t npix = (int64_t)N * HW;
  const int pixb = 256;                                      // 8 pixels per thread; >= 2048 workgroups at 2M pixels
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(head_bn_bwd_apply_kernel<T>, dim3(ceil_div(npix, pixb)), dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy,
                                       scale, shift, mean, invstd, sums, count > 0 ? 1.0 / count : 0.0, w, out, dout, (T*)dy, lddy, tanh0, HW, C,
                                       Co, npix, pixb));
  return check_launch("head_bn_bwd_apply_kernel");
}

}  // extern "C"
