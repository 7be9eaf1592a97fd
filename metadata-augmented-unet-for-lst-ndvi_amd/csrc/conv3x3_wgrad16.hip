// conv3x3_wgrad16.hip -- the weight gradient of the 3x3 convolution on v_mfma_f32_16x16x32 (gfx950), 16-bit operands:
//     dW[tap][co][ci] = sum over pixels  dY[pix][co] * X[pix + tap][ci]            (9 GEMMs, M = co, N = ci, K = pixels)
//
// Same split-K structure, LDS-DMA staging, transposed fragment reads and slab output as conv3x3_wgrad_bf16.hip (which stays the
// kernel for images narrower than a 32-pixel tile row); what changes is the MFMA shape and everything that follows from it.
// Under load the chip holds a higher clock on 16x16x32 than on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS
// give-back (7)); timed in the old kernel's own loop with the operands fed to 16x16x32 MFMAs: -9 % over the U-Net's 18 layers.
//
//   K step    : 32 pixels = ONE ROW of a 4 x 32 pixel tile: lane group g = lane / 16 holds pixels 8g .. 8g+7 of the row (the
//               operand map of 16x16x32: lane l, element j <-> k = 8 (l >> 4) + j), lane l % 16 the channel.  One fragment = two
//               ds_read_b64_tr_b16 (pixels 8g+0..3 | 8g+4..7 of 16 channels), exactly as before
//   workgroup : BCO (64 | 128) output channels x 64 input channels x 9 taps; wave = 64 co x 16 ci x 9 taps = 36 accumulator
//               tiles of 16 x 16 (144 registers): a dY fragment (16 co) meets 9 taps, an X fragment (16 ci) 4 co tiles x up to
//               3 taps.  BCO = 64: two wave groups split the tile's rows and add their accumulators through LDS (fixed order)
//   loop      : walks HALO rows: the X fragment of (halo row R, shift dx) is read once for the tile rows R, R-1, R-2 it is a tap
//               of (dy = 0, 1, 2); a tile row's 4 dY fragments are read when its halo row R comes up, behind the MFMAs of the two
//               older rows (a 3-row register ring): 34 fragment reads per 144 MFMAs
//   images    : dY [128 px][BCO], X halo [6 rows][48-px pitch][64 ch] (34 valid pixels per row; the pitch keeps every swizzle key
//               a per-lane constant); 32-byte chunks XOR-swizzled through the DMA source address so that the 8 pixel rows a
//               half-wave reads sit in 8 distinct bank octets (keys: pixel bits (1,3) for 128-byte rows, (0,1,3) for 256-byte
//               rows -- brute-force checked conflict-free for every fragment the loop reads)
//   output    : split-K partial slabs [split][9][Cout64][Cin64], plain stores; conv3x3.hip's unpack kernel adds them in a
//               fixed order (bitwise reproducible)
//
// Replaces autograd's weight gradient of nn.Conv2d(.,.,3,padding=1) (reference src/model.py:12,14 under loss.backward(),
// src/train.py:252).
#include <stdlib.h>
#include "igemm_bf16_util.h"

namespace mau {
namespace wg3 {
constexpr int TH = 4, TW = 32, HSP = 48, HROWS = TH + 2, HVAL = TW + 2;
constexpr int BCI = 64, XROW = BCI * 2;                  // 128-byte rows of the X halo image
constexpr int XGRP = 5;                                  // 8-pixel DMA groups per halo row (pixels 0..39; 34..39 are zeros)
constexpr int X_Q = HROWS * XGRP;                        // 30 wave-DMAs
constexpr int X_BYTES = HROWS * HSP * XROW;              // 36864
constexpr int DEPTH = 2;                                 // X fragment reads run this many steps ahead of their MFMAs

using igemm::static_for;
using igemm::wait_vmcnt;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// swizzle keys (in 32-byte chunks) of pixel row t
__device__ __forceinline__ int key2(int t) { return ((t >> 1) & 1) | (((t >> 3) & 1) << 1); }      // 128-byte rows
__device__ __forceinline__ int key3(int t) { return (t & 3) | (((t >> 3) & 1) << 2); }             // 256-byte rows

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
template <int N>
__device__ __forceinline__ void land1(Frag& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a.v) : "n"(N));
}
template <int N>
__device__ __forceinline__ void land4(Frag& a, Frag& b, Frag& c, Frag& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.v), "+v"(b.v), "+v"(c.v), "+v"(d.v) : "n"(N));
}
// in place (vdst = srcC): consecutive MFMAs never touch the same accumulator, the epilogue reads them long after
template <bool F16>
__device__ __forceinline__ void mfma(const Frag& a, const Frag& b, f32x4v& c) {
  if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a.v), "v"(b.v));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a.v), "v"(b.v));
}

// LDS-read schedule of one stage for a wave group that owns ROWS tile rows.  Step s = (halo row R = s / 3, dx = s % 3) issues, in
// this order: the X fragment of step s + DEPTH (2 reads), then the dY fragments (4 co tiles = 8 reads) of every tile row whose
// fetch step is s.  Row 0 is fetched in step 0 (nothing to hide behind), rows 1 and 2 one step group EARLY (their ring slots
// are still empty), later rows when their halo row comes up -- behind the MFMAs of the two older rows, whose last use of the
// slot was the halo row before.  DS operations return in order, so "fragment has landed" = at most N operations issued after
// it are outstanding; N comes from these counts.
template <int ROWS>
struct Sched16 {
  static constexpr int NSTEP = (ROWS + 2) * 3;
  static constexpr int fstep(int R) { return R == 0 ? 0 : (R <= 2 ? (R - 1) * 3 + 1 : R * 3); }
  static constexpr int issued(int s) {                 // reads issued up to and including step s (the prologue is step -1)
    int n = 2 * DEPTH;
    for (int k = 0; k <= s; ++k) {
      if (k + DEPTH < NSTEP) n += 2;
      for (int R = 0; R < ROWS; ++R)
        if (fstep(R) == k) n += 8;
    }
    return n;
  }
  static constexpr int pos_b(int k) {                  // reads issued up to and including the X fragment of step k
    if (k < DEPTH) return 2 * (k + 1);
    return issued(k - DEPTH - 1) + 2;                  // (first thing step k - DEPTH issues)
  }
  // (the counter holds 0..15: a smaller N only waits for more than necessary)
  static constexpr int behind_b(int s) { return issued(s) - pos_b(s) > 15 ? 15 : issued(s) - pos_b(s); }
  static constexpr int behind_a(int R) { return issued(3 * R) - issued(fstep(R)) > 15 ? 15 : issued(3 * R) - issued(fstep(R)); }
};

// MIXED: a 64-channel block of input channels may straddle the sources (tensor x | tensor x1 | broadcast embedding; channel counts off
// the 64-channel grid): an X slot then issues one exec-masked DMA per source instead of one DMA.
template <int BCO, int KG, bool F16, bool MIXED>
__global__ __launch_bounds__((BCO / 64) * 4 * KG * 64) void wgrad16_kernel(WgradP p, int nsplit, int xcd_shift) {
  constexpr int WCO = BCO / 64;                      // co halves of 64
  constexpr int NW = WCO * 4 * KG;                   // waves: co half x ci quarter x row group
  constexpr int DYROW = BCO * 2;                     // bytes per dY row
  constexpr int DY_CPR = BCO / 8;                    // 16-byte chunks per dY row
  constexpr int DY_Q = TH * TW * DY_CPR / 64;        // wave-DMAs for the dY tile (32 | 16)
  constexpr int DY_BYTES = DY_Q * 1024;
  constexpr int STAGE = DY_BYTES + X_BYTES;
  constexpr int TOT_Q = DY_Q + X_Q;
  constexpr int PER_WAVE = (TOT_Q + NW - 1) / NW;    // every wave issues exactly PER_WAVE DMAs per stage
  constexpr int ROWS = TH / KG;                      // tile rows of a wave group
  constexpr int NSTEP = (ROWS + 2) * 3;
  static_assert(DY_Q % NW == 0, "dY / halo slots must split at a compile-time j");
  static_assert(PER_WAVE * NW - TOT_Q <= HROWS, "pad slots land in the unused 8-pixel group of a halo row");
  static_assert(PER_WAVE <= NSTEP, "one DMA per step at most");
  constexpr int DY_J = DY_Q / NW;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wcoh = wave % WCO, wci = (wave / WCO) & 3, kg = wave / (4 * WCO);
  // ---- work item (see conv3x3_wgrad_bf16.hip): XCD k owns the splits = k (mod 8) and walks their tiles in order ----
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
  // ---- the X source of this workgroup's 64 input channels: tensor x | tensor x1 (virtual concat) | broadcast embedding.  The
  // launcher guarantees that the block lies inside ONE source (C0 and C0 + C1 on 64-channel boundaries where a next source follows).
  const int Ctot = p.C0 + p.C1;
  const bool from_e = p.E > 0 && ci0 >= Ctot, from_x1 = !from_e && p.C1 > 0 && ci0 >= p.C0;
  const void* xsrc = from_e ? p.emb_lp : (from_x1 ? p.x1 : p.x);
  const int xld = from_e ? 0 : (from_x1 ? p.ldx1 : p.ldx);              // pixel stride (elements); the embedding is per image
  const int xcb = from_e ? Ctot : (from_x1 ? p.C0 : 0);                  // first channel of the source
  const int xlim = from_e ? p.E : (from_x1 ? p.ldx1 : ((p.C1 == 0 && p.E == 0) ? p.ldx : p.C0));   // channels readable from it
  const int xbytes = from_e ? p.N * p.E * 2 : p.N * p.H * p.W * xld * 2;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xsrc), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.N * p.H * p.W * p.lddy * 2, 0x00020000);
  // (MIXED: one resource per source; the class of a lane's 8-channel chunk rides in bits 0-1 of its offset constant)
  const __amdgpu_buffer_rsrc_t rs_m0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.N * p.H * p.W * p.ldx * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x1), 0, p.N * p.H * p.W * p.ldx1 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.emb_lp), 0, p.N * p.E * 2, 0x00020000);

  // ---- per-lane DMA slots.  Slot q = wave + j * NW: j < DY_J is a dY slot; then X slot q - DY_Q = (halo row R, 8-pixel group);
  // beyond TOT_Q a pad slot (zeros into the unused pixel group 40..47 of halo row q - TOT_Q).
  // A wave-DMA is ONE buffer_load ... lds: address = resource base + per-lane byte offset; a lane whose offset is >= the tensor's
  // size reads zeros (hardware range check: image border, zero channels, pad slots).  The offset of a lane is
  //     (tile origin pixel) * pixel stride  [scalar, per stage]  +  s_lc[j]  [per lane, constant]
  // and s_lc[j] = 0x80000000 marks a lane that never reads (the sum then stays beyond every tensor: < 2 GiB, launcher).  Of the
  // pixel position only the COLUMN needs a per-lane test; the row of a slot is wave-uniform (scalar unit).
  // (With 64-bit address arithmetic and three-way selects per DMA -- ~25 vector instructions -- this kernel was no faster than
  //  the 32x32x16 one: a 16x16x32 MFMA leaves 8 of its 16 cycles for other vector instructions, a 32x32x16 one 24 of its 32.)
  constexpr int NEVER = (int)0x80000000u;
  int s_lc[PER_WAVE];
  const int l8 = lane >> 3, l16 = lane >> 4;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int q = wave + j * NW;
    s_lc[j] = NEVER;
    if (j < DY_J) {
      const int slot = q * 64 + lane;
      const int row = slot / DY_CPR, pc = slot % DY_CPR;                 // row = pixel 32 * r + x of the tile
      const int lc = pc ^ ((BCO == 64 ? key2(row) : key3(row)) << 1);
      const int c = co0 + 8 * lc;
      if (c < p.lddy) s_lc[j] = (((row >> 5) * p.W + (row & 31)) * p.lddy + c) * 2;
    } else if (q < TOT_Q) {
      const int xq = q - DY_Q, R = xq / XGRP, grp = xq % XGRP;
      const int hx = grp * 8 + l8, pc = lane & 7;
      const int lc = pc ^ (key2(hx) << 1);
      if constexpr (!MIXED) {
        const int c = ci0 + 8 * lc - xcb;                                 // channel inside the source
        if (hx < HVAL && c < xlim) s_lc[j] = (((R - 1) * p.W + (hx - 1)) * xld + c) * 2;   // (relative to the tile origin; may be negative)
      } else {
        const int c = ci0 + 8 * lc, rel = (R - 1) * p.W + (hx - 1);
        if (hx < HVAL) {                                                  // class 0: x (and lanes that read nothing), 1: x1, 2: embedding
          if (c < p.C0 || (p.C1 == 0 && p.E == 0 && c < p.ldx)) s_lc[j] = (rel * p.ldx + c) * 2;
          else if (c >= p.C0 && c < p.C0 + p.C1) s_lc[j] = ((rel * p.ldx1 + c - p.C0) * 2) | 1;
          else if (c >= Ctot && c < Ctot + p.E) s_lc[j] = ((c - Ctot) * 2) | 2;
        }
      }
    }
  }
  // one wave-DMA (1 KiB); (n, ty0, tx0) wave-uniform.  ~6 vector instructions.
  auto issue_slot = [&](auto jc, int stage, int n, int ty0, int tx0) {
    constexpr int j = decltype(jc)::value;
    const int q = wave + j * NW;
    // destination (wave-uniform, scalar unit): dY slots are dense; X slot = (halo row, 8-pixel group); a pad slot lands in the
    // unused pixel group 40..47 of a halo row
    const int dst = j < DY_J ? q * 1024
                             : (q < TOT_Q ? DY_BYTES + (((q - DY_Q) / XGRP) * HSP + ((q - DY_Q) % XGRP) * 8) * XROW
                                          : DY_BYTES + ((q - TOT_Q) * HSP + XGRP * 8) * XROW);
    igemm::lds_ptr ldst = (igemm::lds_ptr)(smem + stage * STAGE + dst);
    if constexpr (j < DY_J) {
      // slot rows: DY_CPR = 16: row = 4 q + lane / 16; DY_CPR = 8: row = 8 q + lane / 8
      const int ry = DY_CPR == 16 ? (q >> 3) : (q >> 2);                 // (scalar)
      const int rx = DY_CPR == 16 ? ((q & 7) * 4 + l16) : ((q & 3) * 8 + l8);
      const bool rowok = ty0 + ry < p.H;
      const int base = ((n * p.H + ty0) * p.W + tx0) * p.lddy * 2;
      int voff = base + s_lc[j];
      voff = (rowok && tx0 + rx < p.W) ? voff : -1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, ldst, 16, voff, 0, 0, 0);
    } else {
      const int xq = q - DY_Q, R = xq / XGRP, grp = xq % XGRP;           // (scalar; a pad slot's s_lc is NEVER)
      const int gy = ty0 + R - 1, gx = tx0 + grp * 8 + l8 - 1;
      const bool rowok = (unsigned)gy < (unsigned)p.H;
      if constexpr (!MIXED) {
        const int base = from_e ? n * p.E * 2 : ((n * p.H + ty0) * p.W + tx0) * xld * 2;
        int voff = base + s_lc[j];
        voff = (rowok && (unsigned)gx < (unsigned)p.W) ? voff : -1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, ldst, 16, voff, 0, 0, 0);
      } else {
        // one DMA per source, each under the exec mask of its lanes (a masked lane writes nothing; class 0 also carries the lanes
        // that read nothing: offset NEVER -> zeros).  Every wave issues all three: the DMA count per stage stays wave-uniform.
        const int tpix = (n * p.H + ty0) * p.W + tx0, cls = s_lc[j] & 3, lc = s_lc[j] & ~3;
        const bool inb = rowok && (unsigned)gx < (unsigned)p.W;
        if (cls == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m0, ldst, 16, inb ? tpix * p.ldx * 2 + lc : -1, 0, 0, 0);
        if (cls == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m1, ldst, 16, inb ? tpix * p.ldx1 * 2 + lc : -1, 0, 0, 0);
        if (cls == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m2, ldst, 16, inb ? n * p.E * 2 + lc : -1, 0, 0, 0);
      }
    }
  };

  // ---- per-lane transposed-read offsets: lane 4q+pp of 16-lane group g addresses pixel row 8g + q (+4 for the second read),
  // 4 channels at 4 pp of the fragment's 16-channel (32-byte) chunk ----
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = lane >> 4;
  const int arow = 8 * tg + tq;                            // pixel inside the 32-pixel tile row; keys do not depend on the +4
  const int akey = BCO == 64 ? key2(arow) : key3(arow);
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  // A (dY): chunk = wcoh * 4 + cs; the swizzle XORs the cs bits too: co tile k of the lane's pixel row sits at chunk k ^ (low key
  // bits), i.e. offset a_off ^ 32 k (bits 5-6 of the offset hold the chunk) -- four per-lane addresses, one per co tile.
  // All read addresses are RUNNING values of the stage being multiplied (they flip by +-STAGE after every stage): 10 registers.
  const int a_off = (kg * ROWS * 32 + arow) * DYROW + 16 * ((2 * (wcoh * 4) + (tp >> 1)) ^ (akey << 1)) + 8 * (tp & 1);
  unsigned aa[4] = {lds0 + (unsigned)a_off, lds0 + (unsigned)(a_off ^ 32), lds0 + (unsigned)(a_off ^ 64), lds0 + (unsigned)(a_off ^ 96)};
  // B (X halo): pixel t = dx + 8g + q (+4): the key depends on (dx, second read), 6 per-lane addresses
  unsigned bb[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int t = dx + arow + 4 * s;
      bb[dx][s] = lds0 + DY_BYTES + (kg * ROWS * HSP + t) * XROW + 16 * ((2 * wci + (tp >> 1)) ^ (key2(t) << 1)) + 8 * (tp & 1);
    }

  f32x4v acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // ---- split-K: this workgroup's pixel tiles ----
  const int per = (p.nTiles + nsplit - 1) / nsplit;
  const int t0 = split * per, t1 = min(p.nTiles, t0 + per);
  if (t0 < t1) {
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
    static_for<0, PER_WAVE>([&](auto jc) { issue_slot(jc, 0, ln, lty * TH, ltx * TW); });
    int stage = 0;
    for (int tile = t0; tile < t1; ++tile) {
      wait_vmcnt<0>();                                   // this wave's DMAs of the stage about to be multiplied (issued a stage ago)
      __builtin_amdgcn_s_barrier();                      // ... everyone's have landed, and everyone is done reading the other buffer
      advance();
      const int nty0 = lty * TH, ntx0 = ltx * TW, nn = ln;
      Frag fa[3][4], fb[DEPTH + 1];
      // X fragment of step s = (halo row R = s / 3, dx = s % 3)
      auto fetchB = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < NSTEP) {
          constexpr int R = s / 3, dx = s % 3;
          u32x2 lo, hi;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(bb[dx][0]), "n"(R * HSP * XROW));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(bb[dx][1]), "n"(R * HSP * XROW));   // (its base holds the +4 pixels)
          fb[s % (DEPTH + 1)].v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
        }
      };
      // the 4 dY fragments (co tiles) of tile row R: address k reads the LOGICAL co tile k for every lane, because the key's low
      // bits were XORed into the per-lane base (aa[k] = aa0 ^ 32 k)
      auto fetchA = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        static_for<0, 4>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          tr_read<R * 32 * DYROW, 4 * DYROW>(fa[R % 3][k], aa[k]);
        });
      };
      using SC = Sched16<ROWS>;
      static_for<0, DEPTH>([&](auto sc) { fetchB(sc); });
      static_for<0, NSTEP>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int R = s / 3, dx = s % 3;
        fetchB(std::integral_constant<int, s + DEPTH>{});
        static_for<0, ROWS>([&](auto rc) {
          if constexpr (SC::fstep(decltype(rc)::value) == s) fetchA(rc);
        });
        static_assert(SC::behind_b(s) >= 0 && SC::behind_b(s) <= 15, "lgkmcnt range");
        land1<SC::behind_b(s)>(fb[s % (DEPTH + 1)]);
        // older rows first (their dY fragments are in registers), the row of this halo row last
        static_for<1, 3>([&](auto dc) {
          constexpr int dy = decltype(dc)::value, tr = R - dy;
          if constexpr (tr >= 0 && tr < ROWS)
            static_for<0, 4>([&](auto kc) { mfma<F16>(fa[tr % 3][decltype(kc)::value], fb[s % (DEPTH + 1)], acc[dy * 3 + dx][decltype(kc)::value]); });
        });
        if constexpr (R < ROWS) {
          if constexpr (dx == 0) {
            static_assert(SC::behind_a(R) >= 0 && SC::behind_a(R) <= 15, "lgkmcnt range");
            land4<SC::behind_a(R)>(fa[R % 3][0], fa[R % 3][1], fa[R % 3][2], fa[R % 3][3]);
          }
          static_for<0, 4>([&](auto kc) { mfma<F16>(fa[R % 3][decltype(kc)::value], fb[s % (DEPTH + 1)], acc[dx][decltype(kc)::value]); });
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next stage's DMAs: one per step from the head of the stage (conv3x3_wgrad_bf16.hip: time to land is what matters)
        static_for<0, PER_WAVE>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (s == j) {
            issue_slot(jc, stage ^ 1, nn, nty0, ntx0);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      });
      // the read addresses move to the other buffer
      const unsigned flip = stage == 0 ? (unsigned)STAGE : (unsigned)(0u - (unsigned)STAGE);
#pragma unroll
      for (int k = 0; k < 4; ++k) aa[k] += flip;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        bb[dx][0] += flip;
        bb[dx][1] += flip;
      }
      stage ^= 1;
    }
    wait_vmcnt<0>();                                     // the idle re-fetch must land before the LDS is reused / released
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
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((t * 4 + c) * 4 + r) * 64 + lane] = acc[round * 3 + t][c][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[round * 3 + t][c][r] += red[((t * 4 + c) * 4 + r) * 64 + lane];
      }
      __syncthreads();
    }
    if (kg != 0) return;
  }
  // ---- partial slab [split][tap][CoutPad][CinPad]; C/D of 16x16: column (ci) = lane % 16, row (co) = 4 (lane / 16) + register ----
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wcoh * 64 + c * 16 + 4 * tg + r;
        const int ci = ci0 + wci * 16 + (lane & 15);
        out[((size_t)tap * p.CoutPad + co) * p.CinPad + ci] = acc[tap][c][r];
      }
}

template <int BCO, int KG, bool F16, bool MIXED>
static int launch(const WgradP& p, int nsplit, int xcd_shift, hipStream_t st) {
  constexpr int NW = (BCO / 64) * 4 * KG;
  constexpr size_t stage = (size_t)(TH * TW * BCO * 2) + X_BYTES;
  constexpr size_t red = KG == 2 ? (size_t)(NW / 2) * 3 * 16 * 64 * sizeof(float) : 0;
  constexpr size_t lds = 2 * stage > red ? 2 * stage : red;
  static_assert(lds <= 160 * 1024, "LDS budget");
  MAU_LDS_ATTR(lds, &wgrad16_kernel<BCO, KG, F16, MIXED>);
  const int total = nsplit * (p.CoutPad / BCO) * (p.CinPad / BCI);
  // (xcd_shift < 0: a whole number of workgroups per XCD; the surplus ones return at once -- conv3x3_wgrad_bf16.hip wgrad_grid)
  dim3 grid(xcd_shift < 0 ? (((total + (1 << -xcd_shift) - 1) >> -xcd_shift) << -xcd_shift) : total);
  MAU_LAUNCH((wgrad16_kernel<BCO, KG, F16, MIXED>), grid, dim3(NW * 64), lds, st, p, nsplit, xcd_shift);
  return check_launch("wgrad16_kernel");
}
}  // namespace wg3

template <bool F16>
static int launch16(const WgradP& q, bool mixed, int nsplit, int xcd_shift, hipStream_t st) {
  if (q.CoutPad % 128 == 0) return mixed ? wg3::launch<128, 1, F16, true>(q, nsplit, xcd_shift, st) : wg3::launch<128, 1, F16, false>(q, nsplit, xcd_shift, st);
  return mixed ? wg3::launch<64, 2, F16, true>(q, nsplit, xcd_shift, st) : wg3::launch<64, 2, F16, false>(q, nsplit, xcd_shift, st);
}

// mixed: some 64-channel block of input channels straddles two sources
int launch_wgrad16(const WgradP& q, bool f16, bool mixed, int nsplit, int xcd_shift, hipStream_t st) {
  return f16 ? launch16<true>(q, mixed, nsplit, xcd_shift, st) : launch16<false>(q, mixed, nsplit, xcd_shift, st);
}

}  // namespace mau
