// conv3x3_bf16.hip -- the throughput kernel: bf16 3x3 convolution (forward and data gradient) as an
// im2col-free implicit GEMM on v_mfma_f32_32x32x16_bf16, written for gfx950.
//
//   workgroup : 16x16 output pixels x BN output channels (BN = 64: 4 waves, BN = 128: 8 waves)
//   wave      : 64 pixels (4 tile rows) x 64 channels = 2x2 accumulator tiles of 32x32
//               -> one ds_read_b128 per MFMA (LDS array half idle), 64 accumulator registers
//   K loop    : stages of KC = 16 input channels; a stage = the 18x18-pixel halo tile (32 B per
//               pixel) + the 9 x BN x 16 weight slab, i.e. 9 taps x one k16 MFMA step
//   staging   : global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), two LDS stages: the loads of
//               stage k+1 are in flight while stage k is multiplied; one barrier per stage.
//               The LDS image is lane-linear (DMA rule) and XOR-swizzled THROUGH THE SOURCE ADDRESS:
//               16-byte half `hf` of row r lives in slot hf ^ ((r >> 3) & 1), which makes the
//               ds_read_b128 of 16 consecutive rows hit 16 distinct 16-byte bank slots.
//   padding   : out-of-image halo pixels and channels beyond the tensor DMA from a 16-byte zero page
//   broadcast : channels >= C0 DMA from the per-image embedding vector (fuse_embeddings, reference
//               src/model.py:248-259) -- the tiled map never exists in HBM
//   epilogue  : bias, BatchNorm partial statistics from the fp32 accumulators (wave shuffles + LDS),
//               output tile staged through LDS and stored as whole 128-byte pixel rows
//   grid      : 1-D, remapped so that the cout tiles of one pixel tile run back to back on ONE XCD
//               (its L2 serves the re-read of the halo tile); pure speed, no correctness dependence.
//
// Replaces nn.Conv2d(.,.,3,padding=1) + the statistics half of nn.BatchNorm2d of VGGBlock
// (reference src/model.py:12-15).
#include <stdlib.h>
#include "conv_common.h"

namespace mau {

namespace v2 {
constexpr int TS = 16;                 // spatial tile side
constexpr int HS = TS + 2;             // halo side
constexpr int HPIX = HS * HS;          // 324
constexpr int KC = 16;                 // channels per stage
constexpr int ROWB = KC * 2;           // 32 bytes per LDS row (pixel or weight row)
constexpr int HALO_Q = (HPIX * 2 + 63) / 64;   // 11 wave-DMAs (1 KiB each) for the halo tile
constexpr int HALO_BYTES = HALO_Q * 1024;

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(1))) const void* glb_ptr;

__device__ __forceinline__ int swz_off(int row, int half) { return row * ROWB + 16 * (half ^ ((row >> 3) & 1)); }

// ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (same for
// lanes 32-63).  The assignment "MFMA row/column index i -> tile row" is ours to choose: perm32 sends
// the first group to rows 0..15 and the second to rows 16..31, so every group reads 16 CONSECUTIVE
// 32-byte rows, which the (row>>3)&1 chunk swizzle spreads over the 16 distinct 16-byte bank slots.
__device__ __forceinline__ int perm32(int i) {
  return i < 4 ? i : i < 12 ? i + 12 : i < 16 ? i - 8 : i < 20 ? i + 8 : i < 28 ? i - 12 : i;
}

// counted wait on the vector-memory counter (LDS-DMA included); N must be a compile-time constant
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN, int NSTAGE>
__global__ __launch_bounds__(BN * 4) void conv3x3_bf16_kernel(ConvP p, int nPixTiles, int nCt) {
  constexpr int NW = BN / 16;                         // waves per workgroup
  constexpr int W_Q = 9 * BN * 2 / 64;                // wave-DMAs for the weight slab
  constexpr int TOT_Q = HALO_Q + W_Q;
  constexpr int STAGE = HALO_BYTES + W_Q * 1024;      // bytes per LDS stage
  constexpr int PER_WAVE = (TOT_Q + NW - 1) / NW;     // every wave issues exactly this many DMAs per stage
  constexpr int DUMP = NSTAGE * STAGE;                // 1 KiB dump slot for the padding DMAs
  constexpr int DIST = NSTAGE - 1;                    // stages in flight ahead of the one being multiplied
  static_assert(NSTAGE == 2 || NSTAGE == 3, "2 or 3 LDS stages");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NSTAGE * STAGE + 1 KiB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;            // 4 (pixel rows) x BN/64 (channels)
  const int i32 = perm32(lane & 31), h = lane >> 5;   // tile row (pixel / channel) this lane's operands come from

  // ---- XCD-aware block -> (pixel tile, cout tile) map ----
  const int P = blockIdx.x;
  const int xcd = P & 7, seq = P >> 3;
  const int pixTile = xcd + 8 * (seq / nCt);
  const int ct = seq % nCt;
  if (pixTile >= nPixTiles) return;                   // whole workgroup leaves before any barrier
  int t = pixTile;
  const int txi = t % p.tilesX;
  t /= p.tilesX;
  const int tyi = t % p.tilesY;
  const int n = t / p.tilesY;
  const int ty0 = tyi * TS, tx0 = txi * TS;
  const int co0 = ct * BN;

  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(p.w);
  const bf16* __restrict__ embn = reinterpret_cast<const bf16*>(p.emb_lp) + (size_t)n * p.E;
  const bf16* zero = reinterpret_cast<const bf16*>(g_zero_page);

  // ---- per-lane source descriptors of this wave's DMAs (constant over stages) ----
  const bf16* src_base[PER_WAVE];   // halo: pixel base (channel 0) or nullptr when outside the image; weights: row base
  int src_c[PER_WAVE];              // halo: 8 * logical half;  weights: unused
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int q = wave + j * NW;
    src_base[j] = nullptr;
    src_c[j] = 0;
    if (q < HALO_Q) {
      const int slot = q * 64 + lane;
      const int hp = slot >> 1, ph = slot & 1;
      const int lh = ph ^ ((hp >> 3) & 1);
      src_c[j] = 8 * lh;
      if (hp < HPIX) {
        const int gy = ty0 + hp / HS - 1, gx = tx0 + hp % HS - 1;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) src_base[j] = xg + ((size_t)(n * p.H + gy) * p.W + gx) * (size_t)p.ldx;
      }
    } else if (q < TOT_Q) {
      const int slot = (q - HALO_Q) * 64 + lane;
      const int row = slot >> 1, ph = slot & 1;       // row = tap * BN + co
      const int lh = ph ^ ((row >> 3) & 1);
      const int tap = row / BN, co = row % BN;
      src_base[j] = wg + ((size_t)tap * p.CoutPad + co0 + co) * KC + 8 * lh;
    }
  }
  const size_t w_stage_stride = (size_t)9 * p.CoutPad * KC;

  auto issue = [&](int stage, int chunk) {
    const int c0 = chunk * KC;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      const int q = wave + j * NW;                    // wave-uniform
      const bf16* src = zero;
      int dst = DUMP;                                 // padding DMA (keeps the per-wave count uniform for vmcnt)
      if (q < TOT_Q) {
        dst = stage * STAGE + q * 1024;
        if (q < HALO_Q) {
          const int c = c0 + src_c[j];
          if (src_base[j] != nullptr) {
            if (c < p.C0 || (p.E == 0 && c < p.ldx)) src = src_base[j] + c;
            else if (c < p.C0 + p.E) src = embn + (c - p.C0);
          }
        } else {
          src = src_base[j] + (size_t)chunk * w_stage_stride;
        }
      }
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(smem + dst), 16, 0, 0);
    }
  };

  // ---- per-lane LDS read offsets (constant over stages) ----
  int aoff[2][9];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hp = (wm * 4 + mt * 2 + (i32 >> 4) + tap / 3) * HS + (i32 & 15) + tap % 3;
      aoff[mt][tap] = swz_off(hp, h);
    }
  const int boff0 = HALO_BYTES + swz_off(wn * 64 + i32, h);          // (tap*BN is a multiple of 16 rows: swizzle unchanged)
  const int boff1 = HALO_BYTES + swz_off(wn * 64 + 32 + i32, h);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // Pipeline: DIST stages are in flight ahead of the one being multiplied.  Per stage: counted wait for
  // THIS wave's DMAs of the stage (the newer ones stay in flight) -> raw barrier (every wave's DMAs have
  // landed, and everyone has finished reading the buffer that is refilled next) -> issue the stage
  // DIST ahead -> multiply.  __syncthreads() is avoided on purpose: its fence would drain vmcnt to 0.
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (d < p.nChunks) issue(d, d);
  int stage = 0;
  for (int chunk = 0; chunk < p.nChunks; ++chunk) {
    const int newer = min(DIST - 1, p.nChunks - 1 - chunk);      // younger stages that may stay in flight
    if (newer >= 1) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (chunk + DIST < p.nChunks) {
      int ns = stage + DIST;
      if (ns >= NSTAGE) ns -= NSTAGE;
      issue(ns, chunk + DIST);
    }
    const unsigned char* sb = smem + stage * STAGE;
    // register double-buffered operands: the reads of tap t+1 are issued before the MFMAs of tap t, so the
    // compiler's LDS waits become counted (lgkmcnt(4)) instead of draining to 0 in front of every MFMA group
    bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sb + aoff[0][0]);
    bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sb + aoff[1][0]);
    bf16x8 b0 = *reinterpret_cast<const bf16x8*>(sb + boff0);
    bf16x8 b1 = *reinterpret_cast<const bf16x8*>(sb + boff1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf16x8 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (tap < 8) {
        na0 = *reinterpret_cast<const bf16x8*>(sb + aoff[0][tap + 1]);
        na1 = *reinterpret_cast<const bf16x8*>(sb + aoff[1][tap + 1]);
        nb0 = *reinterpret_cast<const bf16x8*>(sb + boff0 + (tap + 1) * BN * ROWB);
        nb1 = *reinterpret_cast<const bf16x8*>(sb + boff1 + (tap + 1) * BN * ROWB);
      }
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    // pin the interleave in the emitted code (LLVM SchedGroupMask: 0x100 = DS read, 0x008 = MFMA):
    // 4 reads of tap 0, then per tap {1 MFMA of tap t, 1 read of tap t+1} x 4; the last tap is MFMA only.
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    stage = stage + 1 == NSTAGE ? 0 : stage + 1;
  }
  __syncthreads();                                     // everyone is done with the stage buffers (no DMA pending)

  // ---- epilogue ----
  bf16* stg = reinterpret_cast<bf16*>(smem) + wave * (64 * 64);      // this wave's 64 pixels x 64 channels
  float* red = reinterpret_cast<float*>(smem + NW * 64 * 64 * 2);    // [4 wm][2][BN]
  float s[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int co = co0 + wn * 64 + nt * 32 + i32;
    const float bv = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int prow = perm32(acc_row(r, h));
        const int pw = mt * 32 + prow;                               // pixel inside the wave tile
        const int gy = ty0 + wm * 4 + (pw >> 4), gx = tx0 + (pw & 15);
        const float v = acc[mt][nt][r] + bv;
        if (gy < p.H && gx < p.W) {
          s[nt] += v;
          q2[nt] += v * v;
        }
        stg[pw * 64 + nt * 32 + i32] = (bf16)v;
      }
    }
  }
  if (p.slab != nullptr) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      s[nt] += __shfl_xor(s[nt], 32);
      q2[nt] += __shfl_xor(q2[nt], 32);
      if (h == 0) {
        red[(wm * 2 + 0) * BN + wn * 64 + nt * 32 + i32] = s[nt];
        red[(wm * 2 + 1) * BN + wn * 64 + nt * 32 + i32] = q2[nt];
      }
    }
  }
  __syncthreads();
  if (p.slab != nullptr && tid < 2 * BN) {
    const int which = tid / BN, c = tid % BN;
    const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c] + red[(2 * 2 + which) * BN + c] +
                    red[(3 * 2 + which) * BN + c];
    p.slab[((size_t)pixTile * 2 + which) * p.CoutPad + co0 + c] = v;
  }
  // whole 128-byte rows: lane (pixel = pass*8 + lane/8, 16-byte vector = lane%8)
  bf16* __restrict__ yg = reinterpret_cast<bf16*>(p.y);
  const int cv = co0 + wn * 64 + (lane & 7) * 8;
  if (cv < p.ldy) {
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int pw = pass * 8 + (lane >> 3);
      const int gy = ty0 + wm * 4 + (pw >> 4), gx = tx0 + (pw & 15);
      if (gy < p.H && gx < p.W) {
        const uint4 v = *reinterpret_cast<const uint4*>(stg + pw * 64 + (lane & 7) * 8);
        *reinterpret_cast<uint4*>(yg + ((size_t)(n * p.H + gy) * p.W + gx) * p.ldy + cv) = v;
      }
    }
  }
}

template <int BN, int NSTAGE>
static int launch(const ConvP& p, hipStream_t st) {
  constexpr int W_Q = 9 * BN * 2 / 64;
  constexpr int STAGE = HALO_BYTES + W_Q * 1024;
  constexpr int NW = BN / 16;
  constexpr size_t epi = (size_t)NW * 64 * 64 * 2 + 4 * 2 * BN * sizeof(float);
  constexpr size_t ring = (size_t)NSTAGE * STAGE + 1024;
  constexpr size_t lds = ring > epi ? ring : epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<BN, NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tilesX = ceil_div(p.W, TS), tilesY = ceil_div(p.H, TS);
  ConvP q = p;
  q.tilesX = tilesX;
  q.tilesY = tilesY;
  q.nChunks = ceil_div(p.C0 + p.E, KC);
  const int nPixTiles = p.N * tilesX * tilesY;
  const int nCt = p.CoutPad / BN;
  const int grid = round_up(nPixTiles, 8) * nCt;
  hipLaunchKernelGGL((conv3x3_bf16_kernel<BN, NSTAGE>), dim3(grid), dim3(BN * 4), lds, st, q, nPixTiles, nCt);
  return check_launch("conv3x3_bf16_kernel");
}
}  // namespace v2

int conv_bf16_v2_num_pixel_tiles(int N, int H, int W) { return N * ceil_div(H, v2::TS) * ceil_div(W, v2::TS); }

int launch_conv_bf16_v2(const ConvP& p, hipStream_t st) {
  static const int stages128 = getenv("MAU_CONV_STAGES128") ? atoi(getenv("MAU_CONV_STAGES128")) : 2;
  static const int stages64 = getenv("MAU_CONV_STAGES64") ? atoi(getenv("MAU_CONV_STAGES64")) : 2;
  if (p.CoutPad % 128 == 0) return stages128 == 3 ? v2::launch<128, 3>(p, st) : v2::launch<128, 2>(p, st);
  return stages64 == 3 ? v2::launch<64, 3>(p, st) : v2::launch<64, 2>(p, st);
}

}  // namespace mau
