// conv_common.h -- declarations shared by the convolution translation units.
#pragma once
#include "mau_common.h"

namespace mau {

// 16 zero bytes in global memory: DMA source of padding pixels / channels (one copy per translation unit)
static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 16-bit operand fragments travel as raw 128-bit registers (typed bf16x8); F16 selects how the matrix core reads them
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct ConvP {
  const void* x;
  int ldx, C0;
  const void* x1;       // optional second tensor source: input channels [C0, C0 + C1) (virtual channel concat), else nullptr
  int ldx1, C1;
  const float* emb;     // fp32 (N,E) broadcast source (float kernels)
  const void* emb_lp;   // the same in the activation dtype (bf16 kernel), filled by the C entry point
  int E;
  const void* w;
  const float* bias;
  const float* post_scale;   // optional: out = relu(post_scale[co] * (conv + bias) + post_shift[co])  (eval-mode BN + ReLU)
  const float* post_shift;
  void* y;
  int ldy, Cout, CoutPad;
  float* slab;
  int N, H, W, tilesX, tilesY, nChunks;
  int xcdShift;         // log2(XCDs of the device) for the XCD-contiguous work-item order (bf16 kernel, set by its launcher)
  int fast;             // (bf16 kernel, set by its launcher) every 16-channel stage lies inside one tensor: buffer-addressed loader
  void* pool = nullptr; // optional (inference epilogue of the 16x16x32 tilings): MaxPool2d(2,2) of the activation, (N, H/2, W/2, ldpool)
  int ldpool = 0;
};

// C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }


// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
struct WgradP {
  const void* x;
  int ldx, C0;
  const void* x1;       // optional second tensor source (channels [C0, C0 + C1) of the convolution's input)
  int ldx1, C1;
  const float* emb;
  const void* emb_lp;   // embedding in the activation dtype (bf16 kernel)
  int E;
  const void* dy;
  int lddy, Cout, CoutPad, Cin, CinPad;
  float* acc;
  int N, H, W, tilesX, tilesY, nTiles;
};

// packed-weight K-chunk per dtype (both 16: one k16 bf16 MFMA step / eight k2 fp32 MFMA steps per tap)
template <typename T>
struct PackKC {
  static constexpr int value = 16;
};

int launch_conv_bf16_v2(const ConvP& p, bool f16, hipStream_t st);   // 16-bit kernels: bf16 or (f16 = true) fp16 operands
int conv_bf16_v2_num_pixel_tiles(int N, int H, int W, int Cout);
void conv_bf16_v2_variant(int N, int H, int W, int Cin, int Cout, int* th, int* nw, int* bn, int* kg);
int launch_wgrad_bf16_v2(const WgradP& p, bool f16, hipStream_t st);      // writes nsplit partial slabs (plain stores)
int wgrad_bf16_v2_splits(int N, int H, int W, int Cout, int Cin);

}  // namespace mau
