// igemm_bf16_util.h -- LDS / wait-counter helpers shared by the bf16 implicit-GEMM convolution kernels (gfx950).
#pragma once
#include <type_traits>
#include "conv_common.h"

namespace mau {
namespace igemm {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(1))) const void* glb_ptr;

// ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (same for
// lanes 32-63).  The assignment "MFMA row/column index i -> tile row" is ours to choose: perm32 sends
// the first group to rows 0..15 and the second to rows 16..31, so every group reads 16 CONSECUTIVE
// 32-byte rows; the chunk swizzles below then spread such a run over the 16 distinct 16-byte bank slots.
__device__ __forceinline__ int perm32(int i) {
  return i < 4 ? i : i < 12 ? i + 12 : i < 16 ? i - 8 : i < 20 ? i + 8 : i < 28 ? i - 12 : i;
}
// counted wait on the vector-memory counter (LDS-DMA included); N must be a compile-time constant
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Operand-fragment reads are issued from inline asm and retired with hand-counted lgkmcnt waits.  Reason: the
// compiler models global_load_lds as a FLAT operation touching both VMEM and LDS, and while one is pending (always,
// in this loop) its waitcnt pass turns every LDS dependency into s_waitcnt lgkmcnt(0) -- each MFMA group then eats a
// full LDS round trip.  Reads issued here are invisible to that pass; land<N>() is the wait, and it re-defines the
// fragments so that their consumers cannot be scheduled above it.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// Epilogue staging through LDS, also from inline asm: for a compiler-visible LDS access that may alias an LDS-DMA
// destination the waitcnt pass emits s_waitcnt vmcnt(0), which here would wait for the wave's OWN global stores
// (stores count in vmcnt) -- one HBM write round trip per 128-byte row group, 37 % of a level-0 layer's time.
template <int OFF>
__device__ __forceinline__ void lds_write_b16(unsigned addr, unsigned v) {
  asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write_b32(unsigned addr, unsigned v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// the accumulator hand-over of the K-group variants (once per item): 16 bytes per lane, waited for by the caller
__device__ __forceinline__ void lds_write_f32x4(unsigned addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 lds_read_f32x4(unsigned addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of two bf16 (round to nearest even): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of two 16-bit values of the activation type (round to nearest even)
template <bool F16>
__device__ __forceinline__ unsigned pack_lp2(f32x2 v) {
  if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  else return pack_bf16x2(v);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_u128(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
__device__ __forceinline__ void lds_land(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// the reads that returned a..d were followed by (at least) 16 more DS operations of this wave: DS operations complete in order, so
// "at most 15 outstanding" means the reads have landed (scalar loads share the counter but only ever make the wait longer)
__device__ __forceinline__ void lds_land_behind16(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Epilogue flavours (compile-time, so the training kernels carry no inference code and vice versa)
constexpr int EPI_PLAIN = 0;   // y = conv (+bias)                      : data gradient
constexpr int EPI_STATS = 1;   // + BatchNorm partial sums              : training forward
constexpr int EPI_POST = 2;    // y = relu(scale*(conv+bias) + shift)   : inference (eval-mode BN + ReLU folded in)

}  // namespace igemm
}  // namespace mau
