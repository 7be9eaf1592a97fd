// mau_common.h -- shared device helpers for libmau_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mau_hip.h"

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace mau {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Launch with a clean error slate: hipGetLastError() is sticky per thread and would otherwise report a
// benign earlier error of some other component in the process as this launch's failure.
#define MAU_LAUNCH(...)               \
  do {                                \
    (void)hipGetLastError();          \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

#define MAU_REQUIRE(cond, ...)                      \
  do {                                              \
    if (!(cond)) {                                  \
      ::mau::set_error(__VA_ARGS__);                \
      return MAU_ERR_ARG;                           \
    }                                               \
  } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- 8-channel vector access on NHWC-ld tensors (16 B for bf16, 32 B for f32) ----
struct F8 {
  float v[8];
};

// fp16 <-> fp32 conversions stay conversions: the backend otherwise folds them into v_fma_mix_f32 / v_fma_mixlo_f16 wherever an
// FMA is adjacent, and v_fma_mixlo_f16 does not round an f16-SUBNORMAL result like v_cvt_f16_f32 does (an exact tie came out
// odd: 761 instead of 762 units of 2^-24) -- two kernels computing the same value then disagree depending on what the
// compiler found next to the conversion.  The empty asm makes the fp32 value opaque at the conversion (no instruction).
__device__ __forceinline__ float opaque(float v) {
  asm("" : "+v"(v));
  return v;
}

template <typename T>
__device__ __forceinline__ F8 load8(const T* p);
template <>
__device__ __forceinline__ F8 load8<float>(const float* p) {
  F8 r;
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[i] = a[i];
    r.v[4 + i] = b[i];
  }
  return r;
}
template <>
__device__ __forceinline__ F8 load8<bf16>(const bf16* p) {
  F8 r;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
  return r;
}
template <>
__device__ __forceinline__ F8 load8<f16>(const f16* p) {
  F8 r;
  f16x8 a = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = opaque((float)a[i]);
  return r;
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const F8& r);
template <>
__device__ __forceinline__ void store8<float>(float* p, const F8& r) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = r.v[i];
    b[i] = r.v[4 + i];
  }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <>
__device__ __forceinline__ void store8<bf16>(bf16* p, const F8& r) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)r.v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}

template <>
__device__ __forceinline__ void store8<f16>(f16* p, const F8& r) {
  f16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (f16)opaque(r.v[i]);
  *reinterpret_cast<f16x8*>(p) = a;
}

// non-temporal (streaming) variants for data touched exactly once by a kernel
template <typename T>
__device__ __forceinline__ F8 load8_nt(const T* p);
template <>
__device__ __forceinline__ F8 load8_nt<float>(const float* p) {
  F8 r;
  const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[i] = a[i];
    r.v[4 + i] = b[i];
  }
  return r;
}
template <>
__device__ __forceinline__ F8 load8_nt<bf16>(const bf16* p) {
  F8 r;
  const bf16x8 a = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)a[i];
  return r;
}
template <>
__device__ __forceinline__ F8 load8_nt<f16>(const f16* p) {
  F8 r;
  const f16x8 a = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p));
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = opaque((float)a[i]);
  return r;
}
template <typename T>
__device__ __forceinline__ void store8_nt(T* p, const F8& r);
template <>
__device__ __forceinline__ void store8_nt<float>(float* p, const F8& r) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = r.v[i];
    b[i] = r.v[4 + i];
  }
  __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p));
  __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p + 4));
}
template <>
__device__ __forceinline__ void store8_nt<bf16>(bf16* p, const F8& r) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)r.v[i];
  __builtin_nontemporal_store(a, reinterpret_cast<bf16x8*>(p));
}

template <>
__device__ __forceinline__ void store8_nt<f16>(f16* p, const F8& r) {
  f16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (f16)opaque(r.v[i]);
  __builtin_nontemporal_store(a, reinterpret_cast<f16x8*>(p));
}

// Per-channel coefficient of channel c (0 beyond C).  The load is UNCONDITIONAL (clamped index) and the zero a select: written as
// `c < C ? p[c] : 0.f` every element became an exec-masked branch around its own load (and, for the fp64 sums, a full
// s_waitcnt vmcnt(0) inside it): 16-48 serialized memory round trips at the head of every thread of the streaming kernels.
__device__ __forceinline__ float coef(const float* __restrict__ p, int c, int C) {
  const float v = p[c < C ? c : C - 1];
  return c < C ? v : 0.f;
}
__device__ __forceinline__ double coef(const double* __restrict__ p, int c, int C) {
  const double v = p[c < C ? c : C - 1];
  return c < C ? v : 0.0;
}

__device__ __forceinline__ F8 zero8() {
  F8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0.f;
  return r;
}

template <typename T>
__device__ __forceinline__ float to_f(T x) {
  return (float)x;
}

// Pixel-block sizes of the slab reductions of bn.hip / head.hip and of their fused forms in bn_fused.hip: a workgroup reduces
// BWD_PIX_PER_BLOCK consecutive pixels as (8 channel vectors) x (32 pixel slots, stride 32) -- the fused kernels keep this
// geometry and per-thread order, so their sums are bit-identical to the two-pass forms they replace.
constexpr int BWD_PIX_PER_BLOCK = 1024;
constexpr int HEAD_MAX_CO = 4;
constexpr int HEAD_PIX_PER_BLOCK = BWD_PIX_PER_BLOCK;

// Shape of the device the calling thread is on, queried once per device: compute units and XCDs (an XCD of gfx950 has 32
// active CUs and its own L2; SPX mode = 8 XCDs = 256 CUs, CPX partitions = 1).  Persistent grids, split counts and the
// XCD-contiguous work-item orders are sized from this, not from literals.  (No device: the MI355X SPX shape.)
struct DeviceShape {
  int cus, xcds, xcd_shift;
};
static inline DeviceShape device_shape() {
  static DeviceShape cache[64];
  static bool have[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DeviceShape{256, 8, 3};
  if (!have[dev]) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    int x = 1, s = 0;
    while (x < 8 && x * 2 * 32 <= cus) {
      x *= 2;
      ++s;
    }
    cache[dev] = DeviceShape{cus, x, s};
    have[dev] = true;
  }
  return cache[dev];
}
// Compute units a persistent grid / a split-K count is sized for: the whole device.  (Round 5 measured a per-thread budget that
// let two chains of the backward pass share the chip by CU: a loss at every split, EXPERIMENTS.md "Round 5 #1"; removed in ABI 5.)
static inline int launch_cus() { return device_shape().cus; }
// dynamic-LDS opt-in of a kernel, once per device (the attribute is per device, a process may drive several)
#define MAU_LDS_ATTR(bytes, ...)                                                                                        \
  do {                                                                                                                  \
    static bool done_[64];                                                                                              \
    int d_ = 0;                                                                                                         \
    if (hipGetDevice(&d_) == hipSuccess && d_ >= 0 && d_ < 64 && !done_[d_]) {                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)(bytes));                                                                          \
      done_[d_] = true;                                                                                                 \
    }                                                                                                                   \
  } while (0)

// grid sizing for streaming kernels: cap at 8 blocks per CU and grid-stride the rest
static inline int stream_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  const int64_t cap = (int64_t)device_shape().cus * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// bn_fused.hip: the 1x1 head on an activation tensor (no BatchNorm), fast path of mau_head_fwd
int head_fwd_fast(const void* a, int lda, const float* w, const float* b, float* out, int tanh0, int dtype, int64_t npix, int HW, int C, int Co,
                  hipStream_t stream);

}  // namespace mau

#define MAU_DISPATCH_DTYPE(dtype, ...)              \
  do {                                              \
    if ((dtype) == MAU_F32) {                       \
      using T = float;                              \
      __VA_ARGS__;                                  \
    } else if ((dtype) == MAU_BF16) {               \
      using T = bf16;                               \
      __VA_ARGS__;                                  \
    } else if ((dtype) == MAU_F16) {                \
      using T = f16;                                \
      __VA_ARGS__;                                  \
    } else {                                        \
      ::mau::set_error("bad dtype %d", (int)dtype); \
      return MAU_ERR_ARG;                           \
    }                                               \
  } while (0)
