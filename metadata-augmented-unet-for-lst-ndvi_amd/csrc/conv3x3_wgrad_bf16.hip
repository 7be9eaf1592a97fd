// conv3x3_wgrad_bf16.hip -- 16-bit (bf16 / fp16) weight gradient of the 3x3 convolution on gfx950:
//     dW[tap][co][ci] = sum over pixels  dY[pix][co] * X[pix + tap][ci]
// as 9 GEMMs that share their operands:  M = co, N = ci, K = pixels.
//
//   workgroup : BCO (64 | 128) output channels x 64 input channels x all 9 taps
//   wave      : 32 co x 32 ci x 9 taps = 9 accumulator tiles (144 accumulator registers)
//   K loop    : stages of one 8x16 pixel tile: dY tile [128 pix][BCO] + X halo tile [180 pix][64]
//   operands  : both GEMM operands are K(pixel)-major in memory (NHWC) but the MFMA wants 8
//               consecutive k per lane -> ds_read_b64_tr_b16 (hardware transpose) straight from the
//               [pixel][channel] image; rows are XOR-swizzled in 16-byte chunks through the DMA
//               SOURCE address so that the 4 rows x 64 bytes a half-wave reads hit distinct banks.
//               The loop walks HALO rows: the X fragment of (halo row R, column shift dx) is read ONCE and
//               multiplied with the dY fragments of the (up to) three tile rows R, R-1, R-2 it is a tap of
//               (dy = 0, 1, 2): 38 fragment reads per 72 MFMAs instead of 80.
//   pipeline  : the fragment reads are inline-asm ds_read_b64_tr_b16 into a register ring, DEPTH steps
//               ahead of the MFMAs that consume them, retired by hand-counted s_waitcnt lgkmcnt(N) (hipcc
//               models an LDS-DMA as touching LDS and turns every compiler-visible LDS dependency into a
//               full drain while one is pending).  Staging: LDS-DMA (global_load_lds_dwordx4), two stages;
//               a stage = s_waitcnt vmcnt(0) + raw s_barrier at its TOP (the stage's own DMAs were issued a
//               whole stage earlier), then the next stage's wave-DMAs are issued one per step from the head of
//               the stage, between the MFMA groups, with select-only (branch-free) source addressing: they stay
//               in flight while the current stage multiplies, and no wait on the vector-memory counter sits
//               between their issue and the MFMAs.  (A third stage, template NS, bought nothing once the DMAs
//               were issued at the head.)
//   reduction : every workgroup owns a pixel range (split-K) and writes its partial with plain
//               coalesced stores into slab [split][9][Cout64][Cin64]; the splits are summed in
//               fixed order by the unpack kernel -> bitwise reproducible, no float atomics
//               (the guide prices float atomics at ~1.3 TB/s chip-wide vs ~6 TB/s plain stores).
//               The 64-channel variant runs two wave groups that split each tile's rows; their
//               accumulators are added through LDS in a fixed order before the store (one slab per workgroup).
//
// Replaces autograd's weight gradient of nn.Conv2d(.,.,3,padding=1) (reference src/model.py:12,14
// under loss.backward(), src/train.py:252).
#include <stdlib.h>
#include "igemm_bf16_util.h"

namespace mau {

namespace wg2 {
constexpr int TH = 8, TW = 16, HW_ = TW + 2, HALO = (TH + 2) * (TW + 2);   // 180
constexpr int BCI = 64;
constexpr int XROW = BCI * 2;                       // 128-byte rows of the X halo image
#ifndef WG_DEPTH
#define WG_DEPTH 2
#endif
constexpr int DEPTH = WG_DEPTH;                     // fragment reads run this many steps ahead of their MFMAs

using igemm::glb_ptr;
using igemm::lds_ptr;
using igemm::static_for;
using igemm::wait_vmcnt;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// chunk swizzles (16-byte chunks): 4 consecutive rows x 64 bytes must cover all 64 banks
__device__ __forceinline__ int swz128(int row) { return ((row >> 1) & 1) << 2; }   // 128-byte rows
__device__ __forceinline__ int swz256(int row) { return (row & 3) << 2; }          // 256-byte rows

// one k16 operand fragment = two transposed 8-byte reads (k 0..3 | 4..7 of the lane half's 8 pixels), joined into the
// MFMA's 4-register operand AT THE READ: joined later (after the counted wait had taken the two halves as separate
// in/out operands) the halves lived in unrelated register pairs and every fragment cost 2-4 v_mov -- 97 copies per stage
// next to 72 MFMAs, in a kernel whose clock falls with every vector instruction.
struct Frag {
  u32x4 v;
};
template <int OFF, int HI>
__device__ __forceinline__ void tr_read(Frag& f, unsigned addr) {
  static_assert(OFF >= 0 && OFF + HI < 65536, "ds offset field");
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi)       // early-clobber: the second read still needs the address register (conv3x3_first.hip tr_frag)
               : "v"(addr), "n"(OFF), "n"(OFF + HI));
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// wait until at most N of this wave's LDS operations are outstanding; re-defines the fragments about to be
// consumed so that their MFMAs cannot be scheduled above the wait
template <int N>
__device__ __forceinline__ void land(Frag& a, Frag& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.v), "+v"(b.v) : "n"(N));
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_frag(const Frag& a, const Frag& b, f32x16 c) {
#ifdef WG_HACK16        // TIMING ONLY (garbage results): the same operands through two v_mfma_f32_16x16x32 on quarters of the accumulator
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[4], c[5], c[6], c[7]}, q2 = {c[8], c[9], c[10], c[11]}, q3 = {c[12], c[13], c[14], c[15]};
  q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), q0, 0, 0, 0);
  q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), q1, 0, 0, 0);
  // (a 32x32x16 MFMA = 16384 MACs = TWO 16x16x32; the other two quarters pass through)
  return f32x16{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
#endif
  if constexpr (F16) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.v), __builtin_bit_cast(f16x8, b.v), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
}

// Step schedule of one stage for a wave group that owns ROWS tile rows: step s = (halo row R = s / 3, dx = s % 3);
// the step with dx == 0 of a halo row R < ROWS also carries the dY fragment of tile row R.
template <int ROWS>
struct Sched {
  static constexpr int NSTEP = (ROWS + 2) * 3;
  static constexpr bool carries_a(int s) { return s % 3 == 0 && s / 3 < ROWS; }
  static constexpr int reads(int s) { return s < NSTEP ? (carries_a(s) ? 4 : 2) : 0; }
  // LDS operations issued after those of step s when step s is consumed
  static constexpr int behind(int s) {
    int n = 0;
    for (int j = s + 1; j <= s + DEPTH; ++j) n += reads(j);
    return n;
  }
};

// KG = number of wave groups that split the tile rows of every stage between them: KG = 2 gives the
// 64-channel variant 8 waves (two per SIMD) instead of 4.
template <int BCO, int KG, bool F16, int NS>
__global__ __launch_bounds__(BCO * 4 * KG) void wgrad_bf16_kernel(WgradP p, int nsplit, int xcd_shift) {
  constexpr int NW = BCO / 16 * KG;                  // waves: (BCO/32) x 2 x KG
  constexpr int WCO = BCO / 32;
  constexpr int DYROW = BCO * 2;                     // bytes per dY row
  constexpr int DY_CPR = BCO / 8;                    // 16-byte chunks per dY row
  constexpr int DY_Q = TH * TW * DY_CPR / 64;        // wave-DMAs for the dY tile (16 | 32)
  constexpr int X_Q = (HALO * 8 + 63) / 64;          // 23 wave-DMAs for the halo tile
  constexpr int DY_BYTES = DY_Q * 1024;
  constexpr int STAGE = DY_BYTES + X_Q * 1024;
  constexpr int TOT_Q = DY_Q + X_Q;
  constexpr int PER_WAVE = (TOT_Q + NW - 1) / NW;    // every wave issues exactly PER_WAVE DMAs per stage (pad slots: zero page)
  constexpr int ROWS = TH / KG;                      // tile rows of a wave group
  using S = Sched<ROWS>;
  constexpr int NSTEP = S::NSTEP;
  static_assert(PER_WAVE <= NSTEP, "one DMA per step at most");
  static_assert(ROWS % 2 == 0, "row parity of the halo swizzle");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NS * max(STAGE, PER_WAVE * NW KiB)
  constexpr int STRIDE = PER_WAVE * NW * 1024 > STAGE ? PER_WAVE * NW * 1024 : STAGE;   // stage pitch incl. pad slots

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave % WCO, wci = (wave / WCO) & 1, kg = wave / (2 * WCO);
  // ---- work item: (pixel split, co tile, ci tile) from the 1-D workgroup id ----
  // The workgroups of one split stream the SAME pixels (dY tile shared by the ci tiles, X halo shared by the co tiles).
  // Hardware deals workgroup ids round-robin over the XCDs (id % 8; speed only): with xcd_shift > 0 (nsplit a multiple of the XCD
  // count) XCD k owns the splits = k (mod 8) and walks their tiles in order, so all tiles of a split are resident on ONE
  // XCD at about the same time and its L2 serves the re-reads.  With id = split + nsplit * tile (the other form), a split
  // count like 85 or 21 scattered the tiles of a split over all XCDs: conv0_1.conv1 read its 268 MB dY three times.
  int split, tile_id;
  {
    const int nCo = p.CoutPad / BCO, tilesOut = nCo * (p.CinPad / BCI);
    const int id = blockIdx.x;
    if (xcd_shift > 0) {
      const int k = id & ((1 << xcd_shift) - 1), j = id >> xcd_shift;
      tile_id = j % tilesOut;
      split = ((j / tilesOut) << xcd_shift) + k;
    } else if (xcd_shift < 0) {
      // any other split count: the (split, tile) list in split-major order is cut into one contiguous run per XCD -- the tiles of a
      // split are still resident on one XCD (two at a run boundary) at about the same time.  The grid is rounded up to a whole number
      // of workgroups per XCD; the (at most XCDs - 1) surplus workgroups leave here, before any barrier.
      const int sh = -xcd_shift, k = id & ((1 << sh) - 1), j = id >> sh;
      const int total = nsplit * tilesOut, per = (total + (1 << sh) - 1) >> sh;
      const int w = k * per + j;
      if (w >= total) return;
      split = w / tilesOut;
      tile_id = w - split * tilesOut;
    } else {
      split = id % nsplit;
      tile_id = id / nsplit;
    }
  }
  const int co0 = (tile_id % (p.CoutPad / BCO)) * BCO, ci0 = (tile_id / (p.CoutPad / BCO)) * BCI;
  const unsigned short* __restrict__ xg = reinterpret_cast<const unsigned short*>(p.x);
  const unsigned short* __restrict__ x1g = reinterpret_cast<const unsigned short*>(p.x1);
  const unsigned short* __restrict__ dyg = reinterpret_cast<const unsigned short*>(p.dy);
  const unsigned short* __restrict__ emb = reinterpret_cast<const unsigned short*>(p.emb_lp);
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero_page);

  // ---- per-lane DMA slots (tile independent part).  Slot q = wave + j * NW of a stage: q < DY_Q is a dY slot, which
  // for every wave means j < DY_Q / NW (compile time); source selection is selects only, no control flow at issue time ----
  static_assert(DY_Q % NW == 0, "dY / halo slots must split at a compile-time j");
  constexpr int DY_J = DY_Q / NW;
  int s_ry[PER_WAVE], s_rx[PER_WAVE], s_c[PER_WAVE];   // pixel offset inside the tile (halo: -1..), first channel of the 16-byte chunk
  bool s_t[PER_WAVE], s_t2[PER_WAVE], s_e[PER_WAVE];   // source class: tensor (dY or X) / second X tensor / X broadcast embedding / none = zero page
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int q = wave + j * NW;
    s_ry[j] = s_rx[j] = s_c[j] = 0;
    s_t[j] = s_t2[j] = s_e[j] = false;
    if (j < DY_J) {
      const int slot = q * 64 + lane;
      const int row = slot / DY_CPR, pc = slot % DY_CPR;
      const int lc = pc ^ (BCO == 64 ? swz128(row) : swz256(row));
      s_ry[j] = row >> 4;
      s_rx[j] = row & 15;
      s_c[j] = co0 + 8 * lc;
      s_t[j] = s_c[j] < p.lddy;
    } else if (q < TOT_Q) {
      const int slot = (q - DY_Q) * 64 + lane;
      const int row = slot >> 3, pc = slot & 7;
      const int lc = pc ^ swz128(row);
      const int c = ci0 + 8 * lc;
      s_ry[j] = row / HW_ - 1;
      s_rx[j] = row % HW_ - 1;
      s_c[j] = c;
      // input channels: [0, C0) from x | [C0, C0 + C1) from x1 (virtual concat) | then E broadcast channels | zero
      s_t[j] = row < HALO && (c < p.C0 || (p.C1 == 0 && p.E == 0 && c < p.ldx));
      s_t2[j] = row < HALO && c >= p.C0 && c < p.C0 + p.C1;
      s_e[j] = row < HALO && c >= p.C0 + p.C1 && c < p.C0 + p.C1 + p.E;
    }
  }
  // one wave-DMA (1 KiB); (n, ty0, tx0) wave-uniform; predicates and selects only (as conv3x3_bf16.hip's issue_slot)
  auto issue_slot = [&](auto jc, int stage, int n, int ty0, int tx0) {
    constexpr int j = decltype(jc)::value;
    const int q = wave + j * NW;
    const int gy = ty0 + s_ry[j], gx = tx0 + s_rx[j];
    const bool inb = ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
    const size_t pix = (size_t)((n * p.H + gy) * p.W + gx);
    const bool is_t = inb & s_t[j];
    const unsigned short* src;
    if constexpr (j < DY_J) {
      const unsigned short* pt = dyg + pix * (size_t)p.lddy + s_c[j];
      src = is_t ? pt : zero;
    } else {
      const bool is_t2 = inb & s_t2[j];
      // one 64-bit multiply-add on selected operands serves both tensors
      const unsigned short* tb = is_t2 ? x1g : xg;
      const size_t tld = is_t2 ? (size_t)p.ldx1 : (size_t)p.ldx;
      const int tc = is_t2 ? s_c[j] - p.C0 : s_c[j];
      const unsigned short* pt = tb + (pix * tld + tc);
      const unsigned short* pe = emb + (n * p.E + (s_c[j] - p.C0 - p.C1));
      const bool is_e = inb & s_e[j];
      src = (is_t | is_t2) ? pt : (is_e ? pe : zero);
    }
    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(smem + stage * STRIDE + q * 1024), 16, 0, 0);
  };

  // ---- per-lane transposed-read offsets ----
  // lane 4q+p of a 16-lane group addresses row q, 4 channels at 4p; group g = (lane>>4)&1 takes the
  // next 16 channels; h = lane>>5 takes k (pixel) 8..15 of the k16 step.
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1, th = lane >> 5;
  // A (dY): row = trow*16 + 8h + 4s + q  -> swizzle term depends on q only
  const int a_chunk = wco * 4 + 2 * tg + (tp >> 1);
  const int a_row0 = 8 * th + tq;
  const int a_off = a_row0 * DYROW + 16 * (a_chunk ^ (BCO == 64 ? swz128(a_row0) : swz256(a_row0))) + 8 * (tp & 1) + kg * ROWS * 16 * DYROW;
  // B (X halo): row = R*18 + dx + 8h + 4s + q; the swizzle only toggles byte-offset bit 6 and depends on
  // (R & 1, dx, q) -> 6 per-lane bases (dx = 0..2, parity c = 0..1); everything else is an immediate.
  const int b_chunk = wci * 4 + 2 * tg + (tp >> 1);
  const int b_lane = DY_BYTES + (8 * th + tq) * XROW + 16 * b_chunk + 8 * (tp & 1) + kg * ROWS * HW_ * XROW;
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

  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

  // ---- split-K: this workgroup's pixel tiles ----
  const int per = (p.nTiles + nsplit - 1) / nsplit;
  const int t0 = split * per, t1 = min(p.nTiles, t0 + per);
  if (t0 < t1) {
    // (n, ty0, tx0) of the tile being LOADED, advanced incrementally (wave-uniform scalars); past the last tile the cursor
    // stays there: the tile is re-fetched into an idle buffer (nobody reads it), which keeps the DMA count per stage fixed
    int lcur = t0, ltx = t0 % p.tilesX, lty = (t0 / p.tilesX) % p.tilesY, ln = t0 / (p.tilesX * p.tilesY);
    auto advance = [&]() {
      if (lcur + 1 < t1) {
        ++lcur;
        if (++ltx == p.tilesX) {
          ltx = 0;
          if (++lty == p.tilesY) {
            lty = 0;
            ++ln;
          }
        }
      }
    };
    static_for<0, NS - 1>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (st > 0) advance();
      static_for<0, PER_WAVE>([&](auto jc) { issue_slot(jc, st, ln, lty * TH, ltx * TW); });
    });
    int stage = 0;
    for (int tile = t0; tile < t1; ++tile) {
      wait_vmcnt<(NS - 2) * PER_WAVE>();               // this wave's DMAs of the stage about to be multiplied (issued NS - 1 stages ago)
      __builtin_amdgcn_s_barrier();                    // ... everyone's have landed, and everyone is done reading the buffer refilled next
      advance();
      const int lstage = stage == 0 ? NS - 1 : stage - 1;   // buffer multiplied in the previous iteration
      const int nty0 = lty * TH, ntx0 = ltx * TW, nn = ln;
      const unsigned sb = lds0 + stage * STRIDE;
      const unsigned aa = sb + a_off;
      const unsigned bb[3][2] = {{sb + b_base[0][0], sb + b_base[0][1]}, {sb + b_base[1][0], sb + b_base[1][1]}, {sb + b_base[2][0], sb + b_base[2][1]}};
      Frag fa[4], fb[DEPTH + 1];
      auto fetch = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < NSTEP) {
          constexpr int R = s / 3, dx = s % 3;
          if constexpr (S::carries_a(s)) tr_read<R * 16 * DYROW, 4 * DYROW>(fa[R & 3], aa);
          tr_read<(R * HW_ + dx) * XROW, 4 * XROW>(fb[s % (DEPTH + 1)], bb[dx][R & 1]);
        }
      };
      static_for<0, DEPTH>([&](auto sc) { fetch(sc); });
      static_for<0, NSTEP>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int R = s / 3, dx = s % 3;
        fetch(std::integral_constant<int, s + DEPTH>{});
        land<S::behind(s)>(fa[(R < ROWS ? R : ROWS - 1) & 3], fb[s % (DEPTH + 1)]);
        static_for<0, 3>([&](auto dc) {
          constexpr int dy = decltype(dc)::value, tr = R - dy;
          if constexpr (tr >= 0 && tr < ROWS) acc[dy * 3 + dx] = mfma_frag<F16>(fa[tr & 3], fb[s % (DEPTH + 1)], acc[dy * 3 + dx]);
        });
        __builtin_amdgcn_sched_barrier(0);
        // The next stage's DMAs: one per step from the HEAD of the stage.  Measured (scripts/variant_bench.py, same box):
        // spread evenly over the stage +5 % time (the late ones stall the next stage's head); the two waves of a SIMD
        // taking turns (one issues while the other multiplies) +4 %: time to land, not issue cost, is what matters.
        static_for<0, PER_WAVE>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (s == j) {
            issue_slot(jc, lstage, nn, nty0, ntx0);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      });
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    wait_vmcnt<0>();                                   // the idle re-fetch must land before the LDS is reused / released
  }

  float* out = p.acc + (size_t)split * 9 * p.CoutPad * p.CinPad;
  if constexpr (KG == 2) {
    // ---- the two wave groups add their accumulators through LDS (fixed order: group 0 + group 1), 3 taps per round ----
    __builtin_amdgcn_s_barrier();
    float* red = reinterpret_cast<float*>(smem) + (size_t)(wave % (NW / 2)) * (3 * 16 * 64);
#pragma unroll
    for (int round = 0; round < 3; ++round) {
      if (kg == 1) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[round * 3 + t][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[round * 3 + t][r] += red[(t * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
    if (kg != 0) return;
  }
  // ---- partial slab: [split][tap][CoutPad][CinPad], 128 contiguous bytes per half-wave ----
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + acc_row(r, th);
      const int ci = ci0 + wci * 32 + (lane & 31);
      out[((size_t)tap * p.CoutPad + co) * p.CinPad + ci] = acc[tap][r];
    }
}

// The kernels' work-item order for a split count (speed only: every (split, tile) computes the same slab whatever its workgroup id).
//   > 0 : XCD k owns the splits = k (mod XCDs) -- split counts that are whole numbers per XCD (round 3);
//   < 0 : the split-major (split, tile) list is cut into one contiguous run per XCD -- ANY other count (round 6); the grid is rounded
//         up to a whole number of workgroups per XCD and the surplus ones return at once;
//     0 : a device with one XCD: plain order id = split + nsplit * tile.
// Either way the workgroups of one split -- which stream the SAME pixels -- are resident on one XCD at about the same time.
static inline bool wgrad_xcd_order() { return true; }
static inline int wgrad_xcd_shift(int nsplit, int tilesOut) {
  const DeviceShape ds = device_shape();
  if (ds.xcds <= 1) return 0;
  return nsplit % ds.xcds == 0 ? ds.xcd_shift : -ds.xcd_shift;
}
static inline int wgrad_grid(int nsplit, int tilesOut, int xcd_shift) {
  const int total = nsplit * tilesOut;
  return xcd_shift < 0 ? (((total + (1 << -xcd_shift) - 1) >> -xcd_shift) << -xcd_shift) : total;
}

template <int BCO, int KG, bool F16, int NS>
static int launch(const WgradP& p, int nsplit, hipStream_t st) {
  constexpr int NW = BCO / 16 * KG;
  constexpr int DY_Q = TH * TW * (BCO / 8) / 64;
  constexpr int X_Q = (HALO * 8 + 63) / 64;
  constexpr int PER_WAVE = (DY_Q + X_Q + NW - 1) / NW;
  constexpr size_t stage = (size_t)(PER_WAVE * NW > DY_Q + X_Q ? PER_WAVE * NW : DY_Q + X_Q) * 1024;
  constexpr size_t red = KG == 2 ? (size_t)(NW / 2) * 3 * 16 * 64 * sizeof(float) : 0;
  constexpr size_t lds = NS * stage > red ? NS * stage : red;
  static_assert(lds <= 160 * 1024, "LDS budget");
  MAU_LDS_ATTR(lds, &wgrad_bf16_kernel<BCO, KG, F16, NS>);
  const int xcd_shift = wgrad_xcd_shift(nsplit, (p.CoutPad / BCO) * (p.CinPad / BCI));
  dim3 grid(wgrad_grid(nsplit, (p.CoutPad / BCO) * (p.CinPad / BCI), xcd_shift));
  MAU_LAUNCH((wgrad_bf16_kernel<BCO, KG, F16, NS>), grid, dim3(BCO * 4 * KG), lds, st, p, nsplit, xcd_shift);
  return check_launch("wgrad_bf16_kernel");
}
}  // namespace wg2

// Which kernel multiplies a layer: the 16x16x32 kernel (conv3x3_wgrad16.hip) works on 4 x 32 pixel tiles, this file's 32x32x16 kernel on
// 8 x 16 ones.  The first is ~9 % faster per MFMA (clock), so it takes every layer unless its tiles waste more than that on
// pixels outside the image (16-pixel-wide bottleneck images: twice the work).  MAU_WGRAD16=0: the 32x32x16 kernel everywhere.
int launch_wgrad16(const WgradP& q, bool f16, bool mixed, int nsplit, int xcd_shift, hipStream_t st);
struct WgVariant {
  bool k16;
  int th, tw;
};
static WgVariant wgrad_variant(int H, int W, bool addressable = true) {
  const char* e = getenv("MAU_WGRAD16");
  const bool allow = addressable && (e == nullptr || atoi(e) != 0);
  const double a16 = (double)ceil_div(H, 4) * 4 * ceil_div(W, 32) * 32, a32 = (double)ceil_div(H, wg2::TH) * wg2::TH * ceil_div(W, wg2::TW) * wg2::TW;
  if (allow && a16 <= 1.08 * a32) return {true, 4, 32};
  return {false, wg2::TH, wg2::TW};
}

int wgrad_bf16_v2_splits(int N, int H, int W, int Cout, int Cin) {
  const int CoutPad = round_up(Cout, 64), CinPad = round_up(Cin, 64);
  const int bco = (CoutPad % 128 == 0) ? 128 : 64;
  const int outTiles = (CoutPad / bco) * (CinPad / 64);
  const WgVariant v = wgrad_variant(H, W);
  const int nTiles = N * ceil_div(H, v.th) * ceil_div(W, v.tw);
  // one workgroup per CU is resident (LDS / accumulator budget): pick the split count whose total
  // workgroup count fills whole rounds of the device's CUs, preferring fewer splits (less slab traffic) and
  // at least 4 pixel tiles per workgroup (pipeline fill).
  const size_t slab_bytes = (size_t)9 * CoutPad * CinPad * sizeof(float);
  size_t cap = ((size_t)256 << 20) / slab_bytes;                 // keep the partial slabs under 256 MiB
  if (cap < 1) cap = 1;
  int smax = nTiles / 4;
  if (smax < 1) smax = 1;
  if (smax > 1024) smax = 1024;
  if ((size_t)smax > cap) smax = (int)cap;
  if (const char* f = getenv("MAU_WGRAD_SPLITS_FORCE")) {      // timing experiments only (read per call): a given split count, clipped
    const int s = atoi(f);
    if (s >= 1) return s > smax ? smax : s;
  }
  // Which split count: a workgroup costs its nTiles / s pixel-tile iterations plus a fixed overhead OV (pipeline fill, the 147-295 KB
  // slab it writes and the second stage reads back), and the launch runs in ceil(workgroups / CUs) rounds:
  //     cost(s) = rounds(s) * (nTiles / s + OV),     OV = 16 iterations of the 4 x 32 tiling, 10 of the 8 x 16 one
  // (fitted to scripts/wgrad_split_probe.py, profiles/r5/wgrad_split_probe.txt: wgrad + second stage per layer over forced split
  // counts).  Round 4's rule -- fill whole rounds, prefer fewer splits -- agrees on sixteen of the U-Net's eighteen layers; TWO change,
  // both K-heavy layers that had bought the last points of grid fill with more rounds of slabs: 768->256 at 64^2 s = 32 -> 8
  // (363 -> 358 us, 226 -> 57 MB of slabs) and 576->1024 at 16^2 s = 7 -> 3 (111 -> 91 us, 149 -> 64 MB).  Ties go to the smaller s
  // (less slab traffic).  A split count that is not a whole number per XCD runs in the plain work-item order, where every XCD reads
  // every pixel range: such counts are charged 10 % -- which is why 1536->512 at 32^2 STAYS at s = 8: the probe's fastest count there
  // (s = 5: 355 us against 371) moved 702 MB through the fabric where s = 8 moves 428 (profiles/r5/layer_traffic.txt of the first
  // records), 4 % of one launch that the step does not see.
  int best = 1;
  double best_cost = 1e300;
  const long cus = launch_cus();
  const double ov = v.k16 ? 16.0 : 10.0;
  for (int s = 1; s <= smax; ++s) {                // s = workgroups along the split axis = partial slabs
    const long blocks = (long)outTiles * s;
    const long rounds = (blocks + cus - 1) / cus;      // (every split count has an XCD-contiguous work-item order: wgrad_xcd_shift)
    const double cost = (double)rounds * ((double)ceil_div(nTiles, s) + ov);
    if (cost < best_cost - 1e-9 * (best_cost < 0 ? -best_cost : best_cost)) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

int launch_wgrad_bf16_v2(const WgradP& p, bool f16, hipStream_t st) {
  const int nsplit = wgrad_bf16_v2_splits(p.N, p.H, p.W, p.Cout, p.Cin);
  // the 16x16x32 kernel addresses its sources through buffer resources (31-bit byte offsets); a 64-channel block of input channels
  // that straddles two sources (x | x1 | broadcast embedding off the 64-channel grid) takes its MIXED loader
  const long long px = (long long)p.N * p.H * p.W;
  const bool addressable = px * p.ldx * 2 < (1ll << 31) && px * p.ldx1 * 2 < (1ll << 31) && px * p.lddy * 2 < (1ll << 31) && (p.E == 0 || p.emb_lp != nullptr);
  const bool mixed = !((p.C1 == 0 || p.C0 % 64 == 0) && (p.E == 0 || (p.C0 + p.C1) % 64 == 0));
  const WgVariant v = wgrad_variant(p.H, p.W, addressable);
  WgradP q = p;
  q.tilesX = ceil_div(p.W, v.tw);
  q.tilesY = ceil_div(p.H, v.th);
  q.nTiles = p.N * q.tilesX * q.tilesY;
  if (v.k16) {
    const int bco = p.CoutPad % 128 == 0 ? 128 : 64;
    return launch_wgrad16(q, f16, mixed, nsplit, wg2::wgrad_xcd_shift(nsplit, (p.CoutPad / bco) * (p.CinPad / 64)), st);
  }
#ifndef WG_NS64
#define WG_NS64 2
#endif
  if (p.CoutPad % 128 == 0) return f16 ? wg2::launch<128, 1, true, 2>(q, nsplit, st) : wg2::launch<128, 1, false, 2>(q, nsplit, st);
  return f16 ? wg2::launch<64, 2, true, WG_NS64>(q, nsplit, st) : wg2::launch<64, 2, false, WG_NS64>(q, nsplit, st);
}

}  // namespace mau
