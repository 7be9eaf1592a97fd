// bn_fused.hip -- BatchNorm2d + ReLU fused with the streaming operator on its far side, so that a full-resolution tensor is
// never written only to be read once (reference src/model.py:13-16 with :218/:268-271 (pool), :241/:284-292 (head)).
//
// Backward of an encoder block's second BatchNorm (its output feeds nn.MaxPool2d AND a skip connection):
//     unfused:  maxpool2x2_bwd_add (read a, dpl, dskip; write da) -> bn_relu_bwd_reduce (read da, y) -> bn_relu_bwd_apply (read da, y; write dy)
//     fused:    pool_bn_bwd_reduce (read y, dpl, dskip)           ->                                   pool_bn_bwd_apply (read y, dpl, dskip; write dy)
//   da = dskip + scatter(dpl to the first maximum of its 2x2 window of a) is recomputed in registers both times; the
//   arg-max comes from the 2-bit-per-channel index tensor the forward's bn_relu_apply_pool wrote (S / 64 bytes).
//   8.25 S -> 5.5 S bytes for an activation of S bytes, one launch fewer.
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
  return opaque((float)(T)opaque(v));          // (a conversion and nothing else: mau_common.h, opaque)
}

// ---- packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950): two channels per vector instruction ----
// Every operation below is the IEEE operation its scalar spelling is (fused multiply-add, multiply, add; conversions per element), so a
// kernel written on pairs returns the bits of the scalar one -- what it saves is issue slots: the head's backward reduce ran ~215 vector
// instructions per 8-channel vector (75 of them v_mov shuffling values into and out of the pairs the SLP vectoriser formed), i.e. it was
// bound by VALU issue (~90 us of pure issue at B = 32 x 256 x 256), not by its 302 MB of traffic.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
template <typename T>
__device__ __forceinline__ f2 round_to2(f2 v) {          // round_to<T> of both elements (one v_cvt_pk_* where the type has one)
  if constexpr (sizeof(T) == 4) return v;
  else if constexpr (__is_same(T, bf16)) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    asm("" : "+v"(v));
    f2 r = __builtin_convertvector(__builtin_convertvector(v, b2), f2);
    asm("" : "+v"(r));
    return r;
  } else return f2{round_to<T>(v[0]), round_to<T>(v[1])};
}
// the eight values of a 16-byte vector as four pairs
template <typename T>
struct P4 {
  f2 p[4];
  __device__ __forceinline__ static P4 load(const T* ptr) {
    const F8 v = load8<T>(ptr);
    P4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.p[i] = f2{v.v[2 * i], v.v[2 * i + 1]};
    return r;
  }
};

// coefficients of 8 channels starting at c0 (zeros beyond C)
struct Coef8 {
  float sc[8], sh[8], mu[8], is[8];
  __device__ __forceinline__ void load(const float* scale, const float* shift, const float* mean, const float* invstd, int c0, int C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = coef(scale, c0 + j, C);
      sh[j] = coef(shift, c0 + j, C);
      mu[j] = mean ? coef(mean, c0 + j, C) : 0.f;
      is[j] = invstd ? coef(invstd, c0 + j, C) : 0.f;
    }
  }
};

// Everything below is STRAIGHT-LINE code: every load is unconditional (indices clamped into the tensor) and conditions only steer
// selects.  A conditional 16-byte load compiles to an exec-masked block that ends in s_waitcnt vmcnt(0): the first version of
// these kernels paid one serialized memory round trip per conditional load (2.0-2.7 TB/s of their traffic).
//
// Loads of the gradient arriving at pixel (n, yi, xi) of an activation that feeds MaxPool2d(2,2) (POOL) and a skip connection (SKIP).
// argidx: 2 bits per channel = position of the window's first maximum, written by bn_relu_apply_pool_kernel in the forward pass.
template <typename T>
struct PoolLoads {
  F8 g, sk;
  unsigned bits;
  bool inpool;
  int me;
};
template <typename T, bool POOL, bool SKIP>
__device__ __forceinline__ PoolLoads<T> pool_load(const T* __restrict__ dpl, int lddpl, const unsigned short* __restrict__ argidx, int nv,
                                                  const T* __restrict__ dskip, int lddskip, int n, int yi, int xi, int H, int W, int c0) {
  PoolLoads<T> L;
  const int Ho = H >> 1, Wo = W >> 1, yo = yi >> 1, xo = xi >> 1;
  L.inpool = POOL && yo < Ho && xo < Wo;
  L.me = (yi & 1) * 2 + (xi & 1);
  L.bits = 0;
  if constexpr (POOL) {
    const int yc = yo < Ho ? yo : Ho - 1, xc = xo < Wo ? xo : Wo - 1;       // (H, W >= 2: Ho, Wo >= 1)
    const size_t win = ((size_t)n * Ho + yc) * Wo + xc;
    L.g = load8<T>(dpl + win * lddpl + c0);
    L.bits = argidx[win * nv + (c0 >> 3)];
  }
  if constexpr (SKIP) L.sk = load8<T>(dskip + (((size_t)n * H + yi) * W + xi) * lddskip + c0);
  return L;
}
// dskip[pixel] + (dpl[window] if this pixel is the window's first maximum), rounded to T as maxpool_bwd_kernel stores it
template <typename T, bool POOL, bool SKIP>
__device__ __forceinline__ F8 pool_eval(const PoolLoads<T>& L) {
  F8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float r = 0.f;
    if constexpr (POOL) r = (L.inpool && (int)((L.bits >> (2 * j)) & 3u) == L.me) ? L.g.v[j] : 0.f;
    if constexpr (SKIP) r += L.sk.v[j];
    o.v[j] = round_to<T>(r);
  }
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
__device__ __forceinline__ void advance32(int& n, int& yi, int& xi, int H, int W) {
  xi += 32;
  while (xi >= W) {
    xi -= W;
    if (++yi == H) {
      yi = 0;
      ++n;
    }
  }
}
template <typename T, bool POOL, bool SKIP>
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dpl, int lddpl,
                                                                 const unsigned short* __restrict__ argidx,
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
  auto accumulate = [&](const F8& v, const F8& g) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float act = fmaf(v.v[j], k.sc[j], k.sh[j]);
      const float dz = act > 0.f ? g.v[j] : 0.f;
      s1[j] += dz;
      s2[j] = fmaf(dz, (v.v[j] - k.mu[j]) * k.is[j], s2[j]);
    }
  };
  if (c0 < C) {
    int64_t p = p0 + ps;
    const int HW = H * W, nv = (C + 7) >> 3;
    int n = (int)(p / HW), r = (int)(p - (int64_t)n * HW);
    int yi = r / W, xi = r - yi * W;
    for (; p + 32 < p1; p += 64) {                       // two pixels per iteration: their loads are issued together
      int n2 = n, yi2 = yi, xi2 = xi;
      advance32(n2, yi2, xi2, H, W);
      const F8 va = load8<T>(y + p * ldy + c0), vb = load8<T>(y + (p + 32) * ldy + c0);
      const PoolLoads<T> La = pool_load<T, POOL, SKIP>(dpl, lddpl, argidx, nv, dskip, lddskip, n, yi, xi, H, W, c0);
      const PoolLoads<T> Lb = pool_load<T, POOL, SKIP>(dpl, lddpl, argidx, nv, dskip, lddskip, n2, yi2, xi2, H, W, c0);
      accumulate(va, pool_eval<T, POOL, SKIP>(La));
      accumulate(vb, pool_eval<T, POOL, SKIP>(Lb));
      n = n2, yi = yi2, xi = xi2;
      advance32(n, yi, xi, H, W);
    }
    for (; p < p1; p += 32) {
      const F8 v = load8<T>(y + p * ldy + c0);
      const PoolLoads<T> L = pool_load<T, POOL, SKIP>(dpl, lddpl, argidx, nv, dskip, lddskip, n, yi, xi, H, W, c0);
      accumulate(v, pool_eval<T, POOL, SKIP>(L));
      advance32(n, yi, xi, H, W);
    }
  }
  bn_sums_to_slab(red, s1, s2, cv, ps, slab, ldslab);
}

// k0/k1 of dy = sc*dz - k0 - k1*y (bn_relu_bwd_apply_kernel's form)
struct Apply8 {
  float sc[8], sh[8], k0[8], k1[8];
  __device__ __forceinline__ void load(const float* scale, const float* shift, const float* mean, const float* invstd, const double* sums,
                                       double inv_count, int c0, int C) {
    double d1[8], d2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                        // all 48 loads first, no wait between them
      sc[j] = coef(scale, c0 + j, C);
      sh[j] = coef(shift, c0 + j, C);
      k0[j] = coef(mean, c0 + j, C);
      k1[j] = coef(invstd, c0 + j, C);
      d1[j] = coef(sums, c0 + j, C);
      d2[j] = coef(sums + C, c0 + j, C);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mu = k0[j], is = k1[j];
      const float m1 = (float)(d1[j] * inv_count), m2 = (float)(d2[j] * inv_count);
      // (every product rounded on its own, the difference one explicit FMA: the generic and the fused apply kernels must agree
      //  to the bit -- left to the compiler, "a*b - c*d" contracts differently from kernel to kernel, visible in fp16 outputs)
      k1[j] = __fmul_rn(__fmul_rn(sc[j], m2), is);
      k0[j] = fmaf(sc[j], m1, -__fmul_rn(k1[j], mu));
    }
  }
  __device__ __forceinline__ float dy(int j, float yv, float da) const {
    const float act = fmaf(yv, sc[j], sh[j]);
    const float dz = act > 0.f ? da : 0.f;
    return fmaf(sc[j], dz, -fmaf(k1[j], yv, k0[j]));
  }
};

// ---- pool: backward pass 2; one thread = a column of RY 2x2 windows x 8 channels (the four y vectors of a window serve arg-max
// and dy alike; the per-channel coefficients are set up once per thread) ----
template <typename T, bool POOL, bool SKIP>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dpl, int lddpl,
                                                                const unsigned short* __restrict__ argidx,
                                                                const T* __restrict__ dskip, int lddskip, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const double* __restrict__ sums,
                                                                double inv_count, T* __restrict__ dyo, int lddy, int H, int W, int C, int C8,
                                                                int RY) {
  const int nv = C8 >> 3, Wc = (W + 1) >> 1, Hc = (H + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wc * nv) return;
  const int xo = idx / nv, c = (idx - xo * nv) * 8;
  const int n = blockIdx.z;
  if (inv_count <= 0.0) inv_count = 1.0 / sums[2 * C];
  Apply8 k;
  k.load(scale, shift, mean, invstd, sums, inv_count, c, C);
  const int x0 = 2 * xo;
  const bool hasx = x0 + 1 < W;
  const int x1 = hasx ? x0 + 1 : x0;                     // (clamped: a duplicate load whose result is never stored)
  const int xpc = xo < Wo ? xo : Wo - 1;
  for (int yo = blockIdx.y * RY; yo < min(Hc, (int)(blockIdx.y + 1) * RY); ++yo) {
    const int y0 = 2 * yo;
    const bool hasy = y0 + 1 < H;
    const size_t row0 = ((size_t)n * H + y0) * W, row1 = ((size_t)n * H + (hasy ? y0 + 1 : y0)) * W;
    F8 v[4], d[4], g;
    v[0] = load8<T>(y + (row0 + x0) * ldy + c);
    v[1] = load8<T>(y + (row0 + x1) * ldy + c);
    v[2] = load8<T>(y + (row1 + x0) * ldy + c);
    v[3] = load8<T>(y + (row1 + x1) * ldy + c);
    if constexpr (SKIP) {
      d[0] = load8<T>(dskip + (row0 + x0) * lddskip + c);
      d[1] = load8<T>(dskip + (row0 + x1) * lddskip + c);
      d[2] = load8<T>(dskip + (row1 + x0) * lddskip + c);
      d[3] = load8<T>(dskip + (row1 + x1) * lddskip + c);
    }
    const bool inpool = POOL && yo < Ho && xo < Wo;      // (inside the pooled range all four pixels exist)
    unsigned bits = 0;
    if constexpr (POOL) {
      const size_t win = ((size_t)n * Ho + (yo < Ho ? yo : Ho - 1)) * Wo + xpc;
      g = load8<T>(dpl + win * lddpl + c);
      bits = argidx[win * nv + (c >> 3)];
    }
    F8 o[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int arg = inpool ? (int)((bits >> (2 * j)) & 3u) : -1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float r = 0.f;
        if constexpr (POOL) r = arg == u ? g.v[j] : 0.f;
        if constexpr (SKIP) r += d[u].v[j];
        o[u].v[j] = k.dy(j, v[u].v[j], round_to<T>(r));
      }
    }
    store8<T>(dyo + (row0 + x0) * lddy + c, o[0]);
    if (hasx) store8<T>(dyo + (row0 + x0 + 1) * lddy + c, o[1]);
    if (hasy) store8<T>(dyo + (row1 + x0) * lddy + c, o[2]);
    if (hasx && hasy) store8<T>(dyo + (row1 + x0 + 1) * lddy + c, o[3]);
  }
}

// Pixel index -> (image n, pixel q inside the image) for the NCHW fp32 tensors of the head (out, dout): ONE division per thread at
// its first pixel, then steps of 32 pixels with a conditional wrap (HW >= 32, checked by the entry points).  (A 64-bit division per
// pixel -- or a `while (q >= HW)` wrap, which the compiler turns into nested loop copies full of v_mov -- cost more than the
// arithmetic of the pixel.)
__device__ __forceinline__ void pos_step32(int& n, int& q, int HW) {
  q += 32;
  const bool w = q >= HW;
  q -= w ? HW : 0;
  n += w ? 1 : 0;
}

// ---- head: forward with BatchNorm + ReLU applied while the activation is loaded (head_fwd_kernel's mapping and arithmetic) ----
// EX: Co == MC (no per-term `o < Co` selects).  A workgroup owns `pixb` consecutive pixels, a thread one 8-channel vector of every
// 32nd pixel, four pixels per iteration (their loads are issued together).  After the 8-lane butterfly every lane of a pixel holds the
// pixel's Co sums; lane `sub` (0..3) then finishes pixel u = sub of the iteration -- bias, tanh, store -- so that tanhf and the
// address arithmetic run once per FOUR pixels instead of once per pixel (they are exec-masked scalar code: the wave pays in full).
// BN = false: the input IS the activation (eval mode: BatchNorm + ReLU were folded into the convolution's epilogue) -- the plain 1x1
// head, mau_head_fwd's fast path.
template <typename T, int MC, bool EX, bool BN = true>
__global__ __launch_bounds__(256) void head_bn_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out, int tanh0, int HW, int C,
                                                          int Co, int64_t npix, int pixb) {
  const int sub = threadIdx.x & 7, ps = threadIdx.x >> 3;
  float wr[MC][8];
#pragma unroll
  for (int o = 0; o < MC; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[o][j] = (EX || o < Co) ? coef(w + o * C, sub * 8 + j, C) : 0.f;
  Coef8 k;
  if (BN) k.load(scale, shift, nullptr, nullptr, sub * 8, C);
  float bo[MC];
#pragma unroll
  for (int o = 0; o < MC; ++o) bo[o] = (EX || o < Co) ? b[o] : 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * pixb;
  const int64_t p1 = p0 + pixb < npix ? p0 + pixb : npix;
  int n = (int)((p0 + ps) / HW);
  int q = (int)((p0 + ps) - (int64_t)n * HW);
  for (int64_t p = p0 + ps; p - ps < p1; p += 128) {       // (block-uniform trip count: the butterflies below need every lane)
    F8 x[4];
    int nn[4], qq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t pix = p + 32 * u;
      const int64_t pc = pix < npix ? pix : npix - 1;                      // (clamped: unconditional load)
      x[u] = load8<T>(y + pc * ldy + (sub * 8 < C ? sub * 8 : 0));
      nn[u] = n;
      qq[u] = q;
      pos_step32(n, q, HW);
    }
    float acc[4][MC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int o = 0; o < MC; ++o) acc[u][o] = 0.f;
      if (sub * 8 < C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = BN ? round_to<T>(fmaxf(fmaf(x[u].v[j], k.sc[j], k.sh[j]), 0.f)) : x[u].v[j];
#pragma unroll
          for (int o = 0; o < MC; ++o) acc[u][o] = fmaf(a, wr[o][j], acc[u][o]);
        }
      }
#pragma unroll
      for (int o = 0; o < MC; ++o) {
        acc[u][o] += __shfl_xor(acc[u][o], 1);
        acc[u][o] += __shfl_xor(acc[u][o], 2);
        acc[u][o] += __shfl_xor(acc[u][o], 4);
      }
    }
    // lane sub < 4 finishes pixel u = sub
    const int us = sub & 3;
    const int nu = us == 0 ? nn[0] : us == 1 ? nn[1] : us == 2 ? nn[2] : nn[3];
    const int qu = us == 0 ? qq[0] : us == 1 ? qq[1] : us == 2 ? qq[2] : qq[3];
    const bool live = sub < 4 && p + 32 * us < p1;
#pragma unroll
    for (int o = 0; o < MC; ++o) {
      if (EX || o < Co) {
        float r = us == 0 ? acc[0][o] : us == 1 ? acc[1][o] : us == 2 ? acc[2][o] : acc[3][o];
        r += bo[o];
        if (o == 0 && tanh0) r = tanhf(r);
        if (live) out[((size_t)nu * Co + o) * HW + qu] = r;
      }
    }
  }
}

// dz of the head at pixel (n, q): dout * (1 - out^2) on the tanh channel (unconditional loads, clamped channel index)
template <int MC, bool EX>
__device__ __forceinline__ void head_dz(const float* __restrict__ out, const float* __restrict__ dout, int tanh0, int Co, int HW, int n,
                                        int q, float* dz) {
  const size_t base = (size_t)n * Co * HW + q;
  const float t = out[base];
  const float f0 = tanh0 ? fmaf(-t, t, 1.f) : 1.f;
#pragma unroll
  for (int o = 0; o < MC; ++o) {
    const float g = dout[base + (size_t)((EX || o < Co) ? o : Co - 1) * HW];
    dz[o] = (EX || o < Co) ? (o == 0 ? g * f0 : g) : 0.f;
  }
}

// ---- head: backward pass 1.  One workgroup = BWD_PIX_PER_BLOCK pixels x all C <= 64 channels: BatchNorm partial sums (slab row
// as bn_relu_bwd_reduce_kernel) and the head's dW / db partials (slab row as head_bwd_kernel) in one pass over y ----
#ifndef MAU_HEAD_WAVES
#define MAU_HEAD_WAVES 3          // 168 registers instead of 170: three waves per SIMD instead of two
#endif
template <typename T, int MC, bool EX>
__global__ __launch_bounds__(256, MAU_HEAD_WAVES) void head_bn_bwd_reduce_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
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
  // (pairs of channels: see "packed fp32" at the top of the file -- same operations, same order, same bits as the scalar spelling)
  f2 sc2[4], sh2[4], mu2[4], is2[4], s1p[4], s2p[4], dwq[MC][4], wq[MC][4];
  float dbp[MC];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc2[i] = f2{k.sc[2 * i], k.sc[2 * i + 1]};
    sh2[i] = f2{k.sh[2 * i], k.sh[2 * i + 1]};
    mu2[i] = f2{k.mu[2 * i], k.mu[2 * i + 1]};
    is2[i] = f2{k.is[2 * i], k.is[2 * i + 1]};
    s1p[i] = s2p[i] = f2{0.f, 0.f};
  }
#pragma unroll
  for (int o = 0; o < MC; ++o) {
    dbp[o] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dwq[o][i] = f2{0.f, 0.f};
      wq[o][i] = f2{o < Co ? coef(w + o * C, c0 + 2 * i, C) : 0.f, o < Co ? coef(w + o * C, c0 + 2 * i + 1, C) : 0.f};
    }
  }
  auto one = [&](const P4<T>& v, const float* dz) {
#pragma unroll
    for (int o = 0; o < MC; ++o) dbp[o] += dz[o];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f2 act = pk_fma(v.p[i], sc2[i], sh2[i]);
      const f2 a = round_to2<T>(f2{fmaxf(act[0], 0.f), fmaxf(act[1], 0.f)});       // the activation bn_relu_apply stored
      f2 s = f2{0.f, 0.f};
#pragma unroll
      for (int o = 0; o < MC; ++o)
        if (EX || o < Co) {
          s = pk_fma(splat(dz[o]), wq[o][i], s);
          dwq[o][i] = pk_fma(splat(dz[o]), a, dwq[o][i]);
        }
      const f2 da = round_to2<T>(s);                                             // the gradient head_bwd stored
      const f2 dzb = f2{act[0] > 0.f ? da[0] : 0.f, act[1] > 0.f ? da[1] : 0.f};
      s1p[i] += dzb;
      s2p[i] = pk_fma(dzb, (v.p[i] - mu2[i]) * is2[i], s2p[i]);
    }
  };
  // The loop is a software pipeline of depth two: the loads of the NEXT pixel pair (16 bytes of y + the head's out / dout words, per
  // pixel) are issued before the current pair is evaluated.  With the loads at the top of each iteration a wave had 2 KB of y in
  // flight and three waves per SIMD (LDS, registers) made 24 KB per CU -- 6 MB on the chip against the ~11 MB that 5.5 TB/s at ~2 us
  // of latency need: the kernel ran at 2.3 TB/s of its traffic.  Only pairs that lie inside the block are prefetched (no clamps: the
  // addresses stay linear in the loop counter); what is left at the end of the last block runs unpipelined.
  struct Raw {
    P4<T> v;
    float t, g[MC];
  };
  auto fetch = [&](int64_t pa, int na, int qa, Raw& r) {
    r.v = P4<T>::load(y + pa * ldy + c0);
    const size_t base = (size_t)na * Co * HW + qa;
    r.t = out[base];
#pragma unroll
    for (int o = 0; o < MC; ++o) r.g[o] = dout[base + (size_t)((EX || o < Co) ? o : Co - 1) * HW];
  };
  auto eval = [&](const Raw& r) {                        // head_dz's arithmetic on the fetched words
    float dz[MC];
    const float f0 = tanh0 ? fmaf(-r.t, r.t, 1.f) : 1.f;
#pragma unroll
    for (int o = 0; o < MC; ++o) dz[o] = (EX || o < Co) ? (o == 0 ? r.g[o] * f0 : r.g[o]) : 0.f;
    one(r.v, dz);
  };
  if (c0 < C8) {
    int n = (int)((p0 + ps) / HW);
    int q = (int)((p0 + ps) - (int64_t)n * HW);
    int64_t p = p0 + ps;
    if (p + 32 < p1) {
      Raw a, b;
      int n2 = n, q2 = q;
      pos_step32(n2, q2, HW);
      fetch(p, n, q, a);
      fetch(p + 32, n2, q2, b);
      n = n2;
      q = q2;
      pos_step32(n, q, HW);                              // (n, q) = position of pixel p + 64
      for (; p + 96 < p1; p += 64) {                     // invariant: (a, b) = pixels (p, p + 32), fetched; the next pair lies inside the block
        Raw na, nb;
        n2 = n;
        q2 = q;
        pos_step32(n2, q2, HW);
        fetch(p + 64, n, q, na);
        fetch(p + 96, n2, q2, nb);
        eval(a);
        eval(b);
        a = na;
        b = nb;
        n = n2;
        q = q2;
        pos_step32(n, q, HW);
      }
      eval(a);
      eval(b);
      p += 64;
    }
    for (; p < p1; p += 32) {                            // (at most one pixel is left)
      Raw r;
      fetch(p, n, q, r);
      eval(r);
      pos_step32(n, q, HW);
    }
  }
  float s1[8], s2[8], dwp[MC][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s1[2 * i] = s1p[i][0];
    s1[2 * i + 1] = s1p[i][1];
    s2[2 * i] = s2p[i][0];
    s2[2 * i + 1] = s2p[i][1];
#pragma unroll
    for (int o = 0; o < MC; ++o) {
      dwp[o][2 * i] = dwq[o][i][0];
      dwp[o][2 * i + 1] = dwq[o][i][1];
    }
  }
  // head partials (head_bwd_kernel's LDS join; one 64-channel group)
#pragma unroll
  for (int o = 0; o < MC; ++o)
    if (EX || o < Co) {
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
    for (int o = 0; o < MC; ++o)
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
template <typename T, int MC, bool EX>
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
  float wr[MC][8];
#pragma unroll
  for (int o = 0; o < MC; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[o][j] = o < Co ? coef(w + o * C, c0 + j, C) : 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * pixb;
  const int64_t p1 = p0 + pixb < npix ? p0 + pixb : npix;
  int n = (int)((p0 + ps) / HW);
  int q = (int)((p0 + ps) - (int64_t)n * HW);
  auto one = [&](int64_t p, const F8& v, const float* dz) {
    F8 o8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < MC; ++o)
        if (EX || o < Co) s = fmaf(dz[o], wr[o][j], s);
      o8.v[j] = k.dy(j, v.v[j], round_to<T>(s));
    }
    store8<T>(dyo + p * lddy + c0, o8);
  };
  int64_t p = p0 + ps;
  for (; p + 96 < p1; p += 128) {                        // four pixels per iteration: their loads are issued together
    int nn[4], qq[4];
    F8 v[4];
    float dz[4][MC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      nn[u] = n;
      qq[u] = q;
      pos_step32(n, q, HW);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = load8<T>(y + (p + 32 * u) * ldy + c0);
      head_dz<MC, EX>(out, dout, tanh0, Co, HW, nn[u], qq[u], dz[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(p + 32 * u, v[u], dz[u]);
  }
  for (; p < p1; p += 32) {
    float dz[MC];
    const F8 v = load8<T>(y + p * ldy + c0);
    head_dz<MC, EX>(out, dout, tanh0, Co, HW, n, q, dz);
    one(p, v, dz);
    pos_step32(n, q, HW);
  }
}

// mau_head_fwd's fast path (head.hip): heads of at most 64 channels on images of at least 32 pixels
int head_fwd_fast(const void* a, int lda, const float* w, const float* b, float* out, int tanh0, int dtype, int64_t npix, int HW, int C, int Co,
                  hipStream_t stream) {
  const int pixb = npix >= ((int64_t)1 << 20) ? 1024 : 128;
  const dim3 grid(ceil_div(npix, pixb));
  const float* none = nullptr;
#define MAU_HEAD_PLAIN(MC_, EX_)                                                                                                            \
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((head_bn_fwd_kernel<T, MC_, EX_, false>), grid, dim3(256), 0, stream, (const T*)a, lda, none, none, w, b, out, \
                                       tanh0, HW, C, Co, npix, pixb))
  if (Co == 2) MAU_HEAD_PLAIN(2, true);
  else if (Co == 1) MAU_HEAD_PLAIN(2, false);
  else if (Co == HEAD_MAX_CO) MAU_HEAD_PLAIN(HEAD_MAX_CO, true);
  else MAU_HEAD_PLAIN(HEAD_MAX_CO, false);
#undef MAU_HEAD_PLAIN
  return check_launch("head_fwd_kernel");
}

}  // namespace mau

using namespace mau;

extern "C" {

int mau_pool_bn_bwd_reduce(const void* y, int ldy, const void* dpl, int lddpl, const unsigned short* argidx, const void* dskip, int lddskip, const float* scale,
                           const float* shift, const float* mean, const float* invstd, float* slab, int ldslab, int dtype, int N, int H,
                           int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(y && (dpl || dskip) && (!dpl || argidx) && scale && shift && mean && invstd && slab && N > 0 && H >= 2 && W >= 2 && C > 0, "pool_bn_bwd_reduce: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= C8 && (!dpl || (lddpl % 8 == 0 && lddpl >= C8)) && (!dskip || (lddskip % 8 == 0 && lddskip >= C8)) && ldslab >= C,
              "pool_bn_bwd_reduce: bad ld");
  const int64_t npix = (int64_t)N * H * W;
  dim3 grid(ceil_div(npix, BWD_PIX_PER_BLOCK), ceil_div(C, 64));
#define MAU_POOL_REDUCE(P, S)                                                                                                                  \
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((pool_bn_bwd_reduce_kernel<T, P, S>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, (const T*)dpl, \
                                       lddpl, argidx, (const T*)dskip, lddskip, scale, shift, mean, invstd, slab, ldslab, H, W, npix, C))
  if (dpl && dskip) MAU_POOL_REDUCE(true, true);
  else if (dpl) MAU_POOL_REDUCE(true, false);
  else MAU_POOL_REDUCE(false, true);
#undef MAU_POOL_REDUCE
  return check_launch("pool_bn_bwd_reduce_kernel");
}

int mau_pool_bn_bwd_apply(const void* y, int ldy, const void* dpl, int lddpl, const unsigned short* argidx, const void* dskip, int lddskip, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const double* sums, double count, void* dy, int lddy,
                          int dtype, int N, int H, int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(y && (dpl || dskip) && (!dpl || argidx) && dy && sums && scale && shift && mean && invstd && N > 0 && H >= 2 && W >= 2 && C > 0 && count >= 0,
              "pool_bn_bwd_apply: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= C8 && lddy % 8 == 0 && lddy >= C8 && (!dpl || (lddpl % 8 == 0 && lddpl >= C8)) &&
                  (!dskip || (lddskip % 8 == 0 && lddskip >= C8)), "pool_bn_bwd_apply: bad ld");
  MAU_REQUIRE((H + 1) / 2 <= 65535 && N <= 65535, "pool_bn_bwd_apply: H/2 and N must fit a grid dimension");
  // RY window rows per thread (the coefficient set-up -- 48 loads, fp64 means -- is paid once per thread), as long as >= ~2048 workgroups remain
  const int xb = ceil_div(((W + 1) / 2) * (C8 / 8), 256), Hc = (H + 1) / 2;
  int RY = 8;
  while (RY > 1 && (int64_t)xb * ceil_div(Hc, RY) * N < 1024) RY >>= 1;
  dim3 grid(xb, ceil_div(Hc, RY), N);
#define MAU_POOL_APPLY(P, S)                                                                                                                   \
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((pool_bn_bwd_apply_kernel<T, P, S>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, (const T*)dpl, lddpl, \
                                       argidx, (const T*)dskip, lddskip, scale, shift, mean, invstd, sums, count > 0 ? 1.0 / count : 0.0, (T*)dy, lddy, \
                                       H, W, C, C8, RY))
  if (dpl && dskip) MAU_POOL_APPLY(true, true);
  else if (dpl) MAU_POOL_APPLY(true, false);
  else MAU_POOL_APPLY(false, true);
#undef MAU_POOL_APPLY
  return check_launch("pool_bn_bwd_apply_kernel");
}

int mau_head_bn_max_channels(void) { return 64; }

int mau_head_bn_fwd(const void* y, int ldy, const float* scale, const float* shift, const float* w, const float* b, float* out, int tanh0,
                    int dtype, int N, int HW, int C, int Co, mau_stream_t stream) {
  MAU_REQUIRE(y && scale && shift && w && b && out && N > 0 && HW > 0 && C > 0, "head_bn_fwd: bad arguments");
  MAU_REQUIRE(Co >= 1 && Co <= HEAD_MAX_CO && C <= mau_head_bn_max_channels(), "head_bn_fwd: out_channels in [1,%d], C <= %d", HEAD_MAX_CO,
              mau_head_bn_max_channels());
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= round_up(C, 8), "head_bn_fwd: bad ld");
  MAU_REQUIRE(HW >= 32 && (int64_t)N * HW < ((int64_t)1 << 31), "head_bn_fwd: images of at least 32 pixels, fewer than 2^31 pixels in all");
  const int64_t npix = (int64_t)N * HW;
  const int pixb = npix >= ((int64_t)1 << 20) ? 1024 : 128;
  const dim3 grid(ceil_div(npix, pixb));
  // (the reference's head has two outputs: MC = 2 holds half the weight / accumulator registers of the general form; EX: Co == MC)
#define MAU_HEAD_FWD(MC_, EX_)                                                                                                              \
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((head_bn_fwd_kernel<T, MC_, EX_>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, scale, shift, w, b, \
                                       out, tanh0, HW, C, Co, npix, pixb))
  if (Co == 2) MAU_HEAD_FWD(2, true);
  else if (Co == 1) MAU_HEAD_FWD(2, false);
  else if (Co == HEAD_MAX_CO) MAU_HEAD_FWD(HEAD_MAX_CO, true);
  else MAU_HEAD_FWD(HEAD_MAX_CO, false);
#undef MAU_HEAD_FWD
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
  MAU_REQUIRE(HW >= 32 && (int64_t)N * HW < ((int64_t)1 << 31), "head_bn_bwd_reduce: images of at least 32 pixels, fewer than 2^31 pixels in all");
  const int64_t npix = (int64_t)N * HW;
#define MAU_HEAD_RED(MC_, EX_)                                                                                                              \
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((head_bn_bwd_reduce_kernel<T, MC_, EX_>), dim3(ceil_div(npix, BWD_PIX_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, \
                                       (const T*)y, ldy, scale, shift, mean, invstd, w, out, dout, bn_slab, ldslab, head_slab, tanh0, HW, C, C8, Co, npix))
  if (Co == 2) MAU_HEAD_RED(2, true);
  else if (Co == 1) MAU_HEAD_RED(2, false);
  else if (Co == HEAD_MAX_CO) MAU_HEAD_RED(HEAD_MAX_CO, true);
  else MAU_HEAD_RED(HEAD_MAX_CO, false);
#undef MAU_HEAD_RED
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
  MAU_REQUIRE(HW >= 32 && (int64_t)N * HW < ((int64_t)1 << 31), "head_bn_bwd_apply: images of at least 32 pixels, fewer than 2^31 pixels in all");
  const int64_t npix = (int64_t)N * HW;
  // 64 (8) pixels per thread: the coefficient set-up is paid once per thread.  (Round 6, same call: 1024 pixels per workgroup 144 us, 2048
  //  131 us, 4096 161 us; the arithmetic on register pairs +-0 -- profiles/r6/head_apply_variants.txt.)
  const int pixb = npix >= ((int64_t)1 << 20) ? 2048 : 256;
#define MAU_HEAD_APPLY(MC_, EX_)                                                                                                            \
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((head_bn_bwd_apply_kernel<T, MC_, EX_>), dim3(ceil_div(npix, pixb)), dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, \
                                       scale, shift, mean, invstd, sums, count > 0 ? 1.0 / count : 0.0, w, out, dout, (T*)dy, lddy, tanh0, HW, C, Co, npix, pixb))
  if (Co == 2) MAU_HEAD_APPLY(2, true);
  else if (Co == 1) MAU_HEAD_APPLY(2, false);
  else if (Co == HEAD_MAX_CO) MAU_HEAD_APPLY(HEAD_MAX_CO, true);
  else MAU_HEAD_APPLY(HEAD_MAX_CO, false);
#undef MAU_HEAD_APPLY
  return check_launch("head_bn_bwd_apply_kernel");
}

}  // extern "C"
