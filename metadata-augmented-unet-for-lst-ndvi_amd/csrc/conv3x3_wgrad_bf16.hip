// conv3x3_wgrad_bf16.hip -- bf16 weight gradient of the 3x3 convolution on gfx950:
//     dW[tap][co][ci] = sum over pixels  dY[pix][co] * X[pix + tap][ci]
// as 9 GEMMs that share their operands:  M = co, N = ci, K = pixels.
//
//   workgroup : BCO (64 | 128) output channels x 64 input channels x all 9 taps
//   wave      : 32 co x 32 ci x 9 taps = 9 accumulator tiles (144 accumulator registers);
//               the dY fragment of a k16 step is read once and used for the 9 taps
//   K loop    : stages of one 8x16 pixel tile: dY tile [128 pix][BCO] + X halo tile [180 pix][64]
//   staging   : LDS-DMA (global_load_lds_dwordx4), two stages in flight, one barrier per stage
//   operands  : both GEMM operands are K(pixel)-major in memory (NHWC) but the MFMA wants 8
//               consecutive k per lane -> ds_read_b64_tr_b16 (hardware transpose) straight from the
//               [pixel][channel] image; rows are XOR-swizzled in 16-byte chunks through the DMA
//               SOURCE address so that the 4 rows x 64 bytes a half-wave reads hit distinct banks
//   reduction : every workgroup owns a pixel range (split-K) and writes its partial with plain
//               coalesced stores into slab [split][9][Cout64][Cin64]; the splits are summed in
//               fixed order by the unpack kernel -> bitwise reproducible, no float atomics
//               (the guide prices float atomics at ~1.3 TB/s chip-wide vs ~6 TB/s plain stores).
//
// Replaces autograd's weight gradient of nn.Conv2d(.,.,3,padding=1) (reference src/model.py:12,14
// under loss.backward(), src/train.py:252).
#include "conv_common.h"

namespace mau {

namespace wg2 {
constexpr int TH = 8, TW = 16, HW_ = TW + 2, HALO = (TH + 2) * (TW + 2);   // 180
constexpr int BCI = 64;
constexpr int XROW = BCI * 2;                       // 128-byte rows of the X halo image

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;
typedef __attribute__((address_space(1))) const void* glb_ptr;

// chunk swizzles (16-byte chunks): 4 consecutive rows x 64 bytes must cover all 64 banks
__device__ __forceinline__ int swz128(int row) { return ((row >> 1) & 1) << 2; }   // 128-byte rows
__device__ __forceinline__ int swz256(int row) { return (row & 3) << 2; }          // 256-byte rows

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo, int hi_delta) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(lo + hi_delta));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// KG = number of wave groups that split the K (pixel) range of every stage between them: KG = 2 gives the
// 64-channel variant 8 waves (two per SIMD) instead of 4; each group keeps its own accumulators and writes
// its own partial slab (the split-sum downstream adds them like any other split).
template <int BCO, int KG>
__global__ __launch_bounds__(BCO * 4 * KG) void wgrad_bf16_kernel(WgradP p, int nsplit) {
  constexpr int NW = BCO / 16 * KG;                  // waves: (BCO/32) x 2 x KG
  constexpr int WCO = BCO / 32;
  constexpr int DYROW = BCO * 2;                     // bytes per dY row
  constexpr int DY_CPR = BCO / 8;                    // 16-byte chunks per dY row
  constexpr int DY_Q = TH * TW * DY_CPR / 64;        // wave-DMAs for the dY tile (16 | 32)
  constexpr int X_Q = (HALO * 8 + 63) / 64;          // 23 wave-DMAs for the halo tile
  constexpr int DY_BYTES = DY_Q * 1024;
  constexpr int STAGE = DY_BYTES + X_Q * 1024;
  constexpr int TOT_Q = DY_Q + X_Q;
  constexpr int PER_WAVE = (TOT_Q + NW - 1) / NW;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave % WCO, wci = (wave / WCO) & 1, kg = wave / (2 * WCO);
  const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
  const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ dyg = reinterpret_cast<const bf16*>(p.dy);
  const bf16* __restrict__ emb = reinterpret_cast<const bf16*>(p.emb_lp);
  const bf16* zero = reinterpret_cast<const bf16*>(g_zero_page);

  // ---- per-lane DMA slots (tile independent part) ----
  int s_row[PER_WAVE], s_c[PER_WAVE];               // row inside the tile image, first channel of the 16-byte chunk
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int q = wave + j * NW;
    s_row[j] = -1;
    s_c[j] = 0;
    if (q < DY_Q) {
      const int slot = q * 64 + lane;
      const int row = slot / DY_CPR, pc = slot % DY_CPR;
      const int lc = pc ^ (BCO == 64 ? swz128(row) : swz256(row));
      s_row[j] = row;
      s_c[j] = co0 + 8 * lc;
    } else if (q < TOT_Q) {
      const int slot = (q - DY_Q) * 64 + lane;
      const int row = slot >> 3, pc = slot & 7;
      const int lc = pc ^ swz128(row);
      s_row[j] = row < HALO ? row : -1;
      s_c[j] = ci0 + 8 * lc;
    }
  }

  auto issue = [&](int stage, int tile) {
    int tt = tile;
    const int txi = tt % p.tilesX;
    tt /= p.tilesX;
    const int tyi = tt % p.tilesY;
    const int n = tt / p.tilesY;
    const int ty0 = tyi * TH, tx0 = txi * TW;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      const int q = wave + j * NW;                   // wave-uniform
      if (q < TOT_Q) {
        const bf16* src = zero;
        const int row = s_row[j], c = s_c[j];
        if (q < DY_Q) {
          const int gy = ty0 + (row >> 4), gx = tx0 + (row & 15);
          if (gy < p.H && gx < p.W && c < p.lddy) src = dyg + ((size_t)(n * p.H + gy) * p.W + gx) * (size_t)p.lddy + c;
        } else if (row >= 0) {
          const int gy = ty0 + row / HW_ - 1, gx = tx0 + row % HW_ - 1;
          if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            if (c < p.C0 || (p.E == 0 && c < p.ldx)) src = xg + ((size_t)(n * p.H + gy) * p.W + gx) * (size_t)p.ldx + c;
            else if (c < p.C0 + p.E) src = emb + (size_t)n * p.E + (c - p.C0);
          }
        }
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(smem + stage * STAGE + q * 1024), 16, 0, 0);
      }
    }
  };

  // ---- per-lane transposed-read offsets ----
  // lane 4q+p of a 16-lane group addresses row q, 4 channels at 4p; group g = (lane>>4)&1 takes the
  // next 16 channels; h = lane>>5 takes k (pixel) 8..15 of the k16 step.
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1, th = lane >> 5;
  // A (dY): row = trow*16 + 8h + 4s + q  -> swizzle term depends on q only
  const int a_chunk = wco * 4 + 2 * tg + (tp >> 1);
  const int a_row0 = 8 * th + tq;
  const int a_off = a_row0 * DYROW + 16 * (a_chunk ^ (BCO == 64 ? swz128(a_row0) : swz256(a_row0))) + 8 * (tp & 1);
  // B (X halo): row = (trow+dy)*18 + dx + 8h + 4s + q; (row>>1)&1 depends on ((trow+dy)&1, dx, q)
  const int b_chunk = wci * 4 + 2 * tg + (tp >> 1);
  // address = (lane_base ^ swizzle_bit6) + row constants: the swizzle only toggles byte-offset bit 6 and
  // depends on ((trow+dy)&1, dx, q) -> 6 per-lane bases (dx = 0..2, parity c = 0..1); everything else is an
  // immediate (multiples of the 128-byte row).
  const int b_lane = DY_BYTES + (8 * th + tq) * XROW + 16 * b_chunk + 8 * (tp & 1);
  int b_base[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int c = 0; c < 2; ++c) b_base[dx][c] = b_lane ^ (((((dx + tq) >> 1) ^ c) & 1) << 6);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- split-K: this workgroup's pixel tiles ----
  const int split = blockIdx.x;
  const int per = (p.nTiles + nsplit - 1) / nsplit;
  const int t0 = split * per, t1 = min(p.nTiles, t0 + per);
  if (t0 < t1) {
    issue(0, t0);
    __syncthreads();
    int stage = 0;
    for (int tile = t0; tile < t1; ++tile) {
      if (tile + 1 < t1) issue(stage ^ 1, tile + 1);
      const unsigned char* sb = smem + stage * STAGE;
      constexpr int ROWS = TH / KG;                    // tile rows of this wave group
      const unsigned char* sbk = sb + kg * ROWS * HW_ * XROW;
      const unsigned char* sak = sb + a_off + kg * ROWS * 16 * DYROW;
#pragma unroll
      for (int tr = 0; tr < ROWS; ++tr) {              // (trow = kg*ROWS + tr; ROWS is even, so parity(trow) = parity(tr))
        const bf16x8 a = tr_pair(sak + tr * 16 * DYROW, 4 * DYROW);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int dy = tap / 3, dx = tap % 3;
          const bf16x8 b = tr_pair(sbk + b_base[dx][(tr + dy) & 1] + ((tr + dy) * HW_ + dx) * XROW, 4 * XROW);
          acc[tap] = mfma32(a, b, acc[tap]);
        }
      }
      __syncthreads();
      stage ^= 1;
    }
  }

  // ---- partial slab: [split][tap][CoutPad][CinPad], 128 contiguous bytes per half-wave ----
  float* out = p.acc + (size_t)(split * KG + kg) * 9 * p.CoutPad * p.CinPad;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + acc_row(r, th);
      const int ci = ci0 + wci * 32 + (lane & 31);
      out[((size_t)tap * p.CoutPad + co) * p.CinPad + ci] = acc[tap][r];
    }
}

template <int BCO, int KG>
static int launch(const WgradP& p, int nsplit, hipStream_t st) {
  constexpr int DY_Q = TH * TW * (BCO / 8) / 64;
  constexpr int X_Q = (HALO * 8 + 63) / 64;
  constexpr size_t lds = 2 * (size_t)(DY_Q + X_Q) * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<BCO, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(nsplit / KG, p.CoutPad / BCO, p.CinPad / BCI);      // nsplit counts slabs: KG per workgroup
  MAU_LAUNCH((wgrad_bf16_kernel<BCO, KG>), grid, dim3(BCO * 4 * KG), lds, st, p, nsplit / KG);
  return check_launch("wgrad_bf16_kernel");
}
}  // namespace wg2

int wgrad_bf16_v2_splits(int N, int H, int W, int Cout, int Cin) {
  const int CoutPad = round_up(Cout, 64), CinPad = round_up(Cin, 64);
  const int bco = (CoutPad % 128 == 0) ? 128 : 64;
  const int outTiles = (CoutPad / bco) * (CinPad / 64);
  const int nTiles = N * ceil_div(H, wg2::TH) * ceil_div(W, wg2::TW);
  // one workgroup per CU is resident (LDS / accumulator budget): pick the split count whose total
  // workgroup count fills whole rounds of 256 CUs, preferring fewer splits (less slab traffic) and
  // at least 4 pixel tiles per workgroup (pipeline fill).
  const size_t slab_bytes = (size_t)9 * CoutPad * CinPad * sizeof(float);
  size_t cap = ((size_t)256 << 20) / slab_bytes;                 // keep the partial slabs under 256 MiB
  if (cap < 1) cap = 1;
  int smax = nTiles / 4;
  if (smax < 1) smax = 1;
  if (smax > 1024) smax = 1024;
  if ((size_t)smax > cap) smax = (int)cap;
  const int kg = bco == 64 ? 2 : 1;                // slabs written per workgroup (wave groups splitting K)
  int best = 1;
  double best_score = -1.0;
  for (int s = 1; s <= smax; ++s) {                // s = workgroups along the split axis
    if ((size_t)s * kg > cap) break;
    const long blocks = (long)outTiles * s;
    const long rounds = (blocks + 255) / 256;
    const double eff = (double)blocks / (double)(rounds * 256);
    const double score = eff - 0.0015 * s * kg;
    if (score > best_score + 1e-9) {
      best_score = score;
      best = s;
    }
  }
  return best * kg;
}

int launch_wgrad_bf16_v2(const WgradP& p, hipStream_t st) {
  const int nsplit = wgrad_bf16_v2_splits(p.N, p.H, p.W, p.Cout, p.Cin);
  WgradP q = p;
  q.tilesX = ceil_div(p.W, wg2::TW);
  q.tilesY = ceil_div(p.H, wg2::TH);
  q.nTiles = p.N * q.tilesX * q.tilesY;
  if (p.CoutPad % 128 == 0) return wg2::launch<128, 1>(q, nsplit, st);
  return wg2::launch<64, 2>(q, nsplit, st);
}

}  // namespace mau
