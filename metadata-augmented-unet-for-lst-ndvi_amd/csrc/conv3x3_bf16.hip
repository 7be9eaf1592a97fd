// conv3x3_bf16.hip -- the throughput kernel: 16-bit (bf16 / fp16) 3x3 convolution, forward and data gradient, as an
// im2col-free implicit GEMM on the gfx950 matrix cores.
//
//   workgroup : TH x 16 output pixels (TH = 8 | 16 | 32 | 64) x BN output channels (64 | 128), 4 or 8 waves (Geo<BN, MT, NW>);
//               persistent, 1-D grid; every XCD owns a contiguous range of pixel tiles (image order) and runs the cout tiles of
//               one pixel tile back to back: its L2 serves halo overlap and input re-reads (pure speed)
//   wave      : MT x 32 pixels x 64 channels; 64 or 128 fp32 accumulator registers
//   stage     : KC = 16 input channels: the (TH + 2) x 18 halo tile (32 B per pixel) + the 9 x BN x 16 weight slab, staged by
//               LDS-DMA (no VGPR round trip) into two LDS buffers: stage k + 1 lands while stage k is multiplied
//   loader    : FAST = buffer_load ... lds with (resource, scalar stage offset, per-lane item-constant offset): tensor x |
//               tensor x1 (virtual concat) | per-image embedding vector (fuse_embeddings, reference src/model.py:248-259:
//               the tiled map never exists in HBM); the hardware range check supplies the zero padding.  !FAST = 64-bit
//               select-only addresses + a zero page (channel counts off the 16-channel grid)
//   multiply  : M16 (big tiles, even stage count): v_mfma_f32_16x16x32, two taps per MFMA, stages walked in pairs (PairSched);
//               otherwise v_mfma_f32_32x32x16, dx-major with shared halo-row fragments (StageSched).  Both: fragment reads
//               (inline-asm ds_read_b128, counted lgkmcnt) and wave-DMAs sit BETWEEN the MFMAs, from constexpr schedules.
//               The kernel is bound by the clock the chip grants under MFMA load, not by issue slots
//   image     : lane-linear (DMA rule); 32x32x16 path: XOR-swizzled THROUGH THE SOURCE ADDRESS (16-byte half hf of row r in
//               slot hf ^ ((r >> 3) & 1): 16 consecutive rows hit 16 distinct bank slots); 16x16x32 path: unswizzled (its
//               lane groups are conflict-free as they are)
//   epilogue  : bias, BatchNorm partial statistics from the fp32 accumulators, or the inference affine + ReLU; 16-bit
//               conversion, output tile staged through LDS and stored as whole 128-byte pixel rows
//   build     : 120 instantiations; one (epilogue, operand type) slice per translation unit -- see "translation units" below
//
// Replaces nn.Conv2d(.,.,3,padding=1) + the statistics half of nn.BatchNorm2d of VGGBlock
// (reference src/model.py:12-15), torch.cat([skip, up]) (:279-282) and fuse_embeddings (:248-259).
#include <stdlib.h>
#include "igemm_bf16_util.h"

namespace mau {

namespace v2 {
constexpr int TW = 16;                 // tile width (pixels); the tile height TH depends on the variant
constexpr int HS = TW + 2;             // halo row pitch (pixels)
constexpr int KC = 16;                 // channels per stage
constexpr int ROWB = KC * 2;           // 32 bytes per LDS row (pixel or weight row)

using namespace igemm;

// weight rows: swizzle keyed on the row index (rows come in aligned runs of 16)
__device__ __forceinline__ int w_off(int row, int half) { return row * ROWB + 16 * (half ^ ((row >> 3) & 1)); }
// halo pixels: swizzle keyed on the halo COLUMN hx (0..17).  A 16-pixel run hx..hx+15 of one halo row holds
// every residue of hp mod 8 twice, and the two always differ in bit 3 of hx -> 16 distinct slots; and because
// the key ignores the halo row, a tap's (dy, mt) row shift is a pure immediate offset of the read address.
__device__ __forceinline__ int halo_swz(int hx) { return (hx >> 3) & 1; }

// wait until at most N of this wave's LDS operations are outstanding; a, b: the fragments that have landed by then
template <int N>
__device__ __forceinline__ void landed(bf16x8& a, bf16x8& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
// ---- The multiply schedule of one stage (16 input channels x 9 taps) of a wave ----
// Taps are walked dx-major.  For one dx the pixel fragment of halo ROW PAIR r (rows r, r + 1 of the wave's strip, shifted by
// dx) is the operand of every (tile mt, dy) with 2 * mt + dy = r -- up to two tiles -- so it is read ONCE: 2 * MT + 1
// pixel fragments per dx instead of 3 * MT, 9 instead of 12 for MT = 4: 45 fragment reads per stage instead of 54.  (Measured
// against the tap-major loop with 54: no difference in time -- the kernel is clock-bound, DESIGN.md "the clock is the
// bound", and the clock did not move; kept for the LDS bytes.)  A fragment read is issued LEAD MFMAs before its first
// use, in need order; DS operations return in order, so "fragment k has landed" is a counted lgkmcnt.
template <int MT>
struct StageSched {
  static constexpr int NG = 3, ROWS = 2 * MT + 1;
  static constexpr int NM = 18 * MT;                    // MFMAs of a stage
  static constexpr int NR = NG * (ROWS + 6);            // fragment reads of a stage
#ifdef MAU_CONV_LEAD
  static constexpr int LEAD = MAU_CONV_LEAD, RING = MAU_CONV_RING, BRING = MAU_CONV_BRING;
#else
  static constexpr int LEAD = 8;                        // MFMAs between a fragment's read and its first use
  static constexpr int RING = 6, BRING = 10;            // pixel / weight fragment registers (rings in read order)
#endif
  struct MF { int g, r, mt, dy, nt, ka, kb; };          // ka / kb: read index of its pixel / weight fragment
  struct RD { int isA, g, r, dy, nt, need, last, iss, seq; };   // need / last: first / last MFMA using it; iss: issued after MFMA iss (-1: stage head)
  MF mf[NM] = {};
  RD rd[NR] = {};
  int lo[NM + 1] = {};                                  // reads issued after MFMA M: [lo[M + 1], lo[M + 2]) ... see rd_lo()
  constexpr StageSched() {
    int a_idx[NG][ROWS] = {}, b_idx[NG][3][2] = {};
    for (int g = 0; g < NG; ++g) {
      for (int r = 0; r < ROWS; ++r) a_idx[g][r] = -1;
      for (int dy = 0; dy < 3; ++dy) b_idx[g][dy][0] = b_idx[g][dy][1] = -1;
    }
    int m = 0, k = 0, aseq = 0, bseq = 0;
    for (int g = 0; g < NG; ++g)
      for (int r = 0; r < ROWS; ++r)
        for (int which = 0; which < 2; ++which) {       // the tiles row pair r belongs to: (r/2, dy 0 | 1), then (r/2 - 1, dy 2)
          const int dy = (r & 1) ? 1 : (which == 0 ? 0 : 2);
          const int mt = (r & 1) ? (r - 1) / 2 : (which == 0 ? r / 2 : r / 2 - 1);
          if ((r & 1) && which == 1) continue;
          if (mt < 0 || mt >= MT) continue;
          for (int nt = 0; nt < 2; ++nt) {
            if (b_idx[g][dy][nt] < 0) {
              rd[k] = {0, g, r, dy, nt, m, m, 0, bseq++};
              b_idx[g][dy][nt] = k++;
            }
            if (a_idx[g][r] < 0) {
              rd[k] = {1, g, r, dy, nt, m, m, 0, aseq++};
              a_idx[g][r] = k++;
            }
            mf[m] = {g, r, mt, dy, nt, a_idx[g][r], b_idx[g][dy][nt]};
            rd[a_idx[g][r]].last = m;
            rd[b_idx[g][dy][nt]].last = m;
            ++m;
          }
        }
    for (int i = 0; i < NR; ++i) rd[i].iss = rd[i].need - LEAD < -1 ? -1 : rd[i].need - LEAD;
    int c = 0;                                          // lo[M] = number of reads issued before MFMA M is issued
    for (int M = 0; M <= NM; ++M) {
      while (c < NR && rd[c].iss < M) ++c;
      lo[M] = c;
    }
  }
  constexpr bool ok() const {                           // counts, order, and the register ring never overwrites a live fragment
    int m = 0, k = 0;
    for (int i = 0; i < NM; ++i) m += mf[i].mt >= 0;
    for (int i = 0; i < NR; ++i) {
      if (i > 0 && (rd[i].need < rd[i - 1].need || rd[i].iss < rd[i - 1].iss)) return false;
      for (int j = 0; j < i; ++j)                       // the register this read lands in: its previous fragment is done
        if (rd[j].isA == rd[i].isA && rd[j].seq + (rd[i].isA ? RING : BRING) == rd[i].seq && rd[j].last > rd[i].iss) return false;
      ++k;
    }
    return m == NM && k == NR && lo[NM] == NR;
  }
};

// ---- The same stage pair on v_mfma_f32_16x16x32 (M16 variants) ----
// Under load the chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per FLOP
// (MI355X_MICROARCH.md, DVFS give-back (7); timed in this kernel's own loop: +7..10 %).  K = 32 of one MFMA =
// 16 channels of TWO taps: lanes 0-31 (k-slices 0, 1) read tap t of the step, lanes 32-63 (k-slices 2, 3) tap t'; the
// fragment of a lane is still one 16-byte read at its own address, so the LDS image and the loader do not change.  Nine
// taps do not pair inside one 16-channel stage, so two stages are walked together: stage A taps (0,1)(2,3)(4,5)(6,7), the
// CROSS step (tap 8 of A | tap 0 of B) -- after the barrier that publishes B's buffer, before the one that releases A's
// -- then stage B taps (1,2)(3,4)(5,6)(7,8): 9 steps of 32 MFMAs for 2 x 9 taps (three barriers per two stages, not two).
// wave tile = 8 pixel rows of 16 x 4 channel tiles of 16; acc[y][n], lane l: channel column l % 16, pixels 4 * (l / 16) + r.
typedef float f32x4v __attribute__((ext_vector_type(4)));
// In-place (vdst = srcC) by construction: through the builtin the 288 unrolled updates of a stage pair were renamed freely
// and the 32 accumulators came back to their loop-carried registers through 128 v_mov per pair (and spills).  Consecutive
// MFMAs of the schedule never touch the same accumulator; the epilogue reads them long after the last one.
template <bool F16>
__device__ __forceinline__ void mfma32k(const bf16x8& a, const bf16x8& b, f32x4v& c) {
  if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
struct PairSched {
  static constexpr int NS = 9, NM = NS * 32, NR = NS * 12;
  static constexpr int B1 = 127, B2 = 159;              // barriers follow these MFMAs (end of stage A's pairs; end of the cross step)
#ifdef MAU_CONV_M16_LEAD
  static constexpr int LEAD = MAU_CONV_M16_LEAD, RA = MAU_CONV_M16_RA, RB = MAU_CONV_M16_RB;
#else
  static constexpr int LEAD = 12, RA = 6, RB = 8;
#endif
  struct RD { int isA, s, idx, need, last, iss, seq; };   // idx: pixel row y (A) / channel tile n (B)
  RD rd[NR] = {};
  int lo[NM + 1] = {};
  int ka[NS][8] = {}, kb[NS][4] = {};
  constexpr PairSched() {
    int k = 0, aseq = 0, bseq = 0;
    for (int s = 0; s < NS; ++s) {
      const int rel = s <= 3 ? -1 : B1;                  // steps 4.. read stage B's buffer: not before the barrier after MFMA B1
      for (int o = 0; o < 12; ++o) {                     // need order: B0 A0 B1 B2 B3 A1 .. A7
        const bool isA = o == 1 || o >= 5;
        const int idx = o == 0 ? 0 : o == 1 ? 0 : o <= 4 ? o - 1 : o - 4;
        const int need = 32 * s + (isA ? 4 * idx : idx), last = 32 * s + (isA ? 4 * idx + 3 : 28 + idx);
        int iss = need - LEAD;
        if (iss < rel) iss = rel;
        rd[k] = {isA ? 1 : 0, s, idx, need, last, iss, isA ? aseq++ : bseq++};
        if (isA) ka[s][idx] = k;
        else kb[s][idx] = k;
        ++k;
      }
    }
    int c = 0;
    for (int M = 0; M <= NM; ++M) {
      while (c < NR && rd[c].iss < M) ++c;
      lo[M] = c;
    }
  }
  constexpr bool ok() const {
    for (int i = 0; i < NR; ++i) {
      if (i > 0 && (rd[i].need < rd[i - 1].need || rd[i].iss < rd[i - 1].iss)) return false;
      for (int j = 0; j < i; ++j)
        if (rd[j].isA == rd[i].isA && rd[j].seq + (rd[i].isA ? RA : RB) == rd[i].seq && rd[j].last > rd[i].iss) return false;
    }
    return lo[NM] == NR;
  }
};
inline constexpr PairSched kPairSched{};

template <int MT>
inline constexpr StageSched<MT> kStageSched{};

// Two K groups per workgroup (KG = 2 below): which tilings have the form, and when the launcher takes it -- the layer leaves every CU
// at most ONE 4-wave workgroup (nItems <= CUs: single-tile inference at the deep levels), an even stage count (same barriers in both
// groups) of at least four.  The rule depends on the batch and on the device's CU count, and the two groups' partial sums are added in
// a fixed order (group 0 + group 1): the same tile can therefore differ in the last bits between B = 1 and B = 8.
// MAU_CONV_KG=0 switches it off (the A/B and test switch).
constexpr bool has_k_groups(int bn, int mt, int nw) { return bn == 64 && nw == 4 && mt <= 2; }
static inline bool k_groups_rule(int nChunks, int nItems) {
  static const bool kg_on = getenv("MAU_CONV_KG") == nullptr || atoi(getenv("MAU_CONV_KG")) != 0;
  return kg_on && nChunks % 2 == 0 && nChunks >= 4 && nItems <= device_shape().cus;
}

// One (pixel tile, cout tile) work item.
struct Item {
  int pixTile, n, ty0, tx0, co0;
};

// Geometry of a variant: NW waves = WM (pixel direction) x WN (= BN/64 channel direction); a wave owns
// MT x 32 pixels (MT x 2 tile rows of 16) x 64 channels = MT x 2 accumulator tiles of 32x32.
//   <128, 2, 8>: 16x16 px tile     <128, 4, 8>: 32x16 px tile (128 accumulator registers per lane)
//   < 64, 2, 4>: 16x16             < 64, 2, 8>: 32x16             < 64, 4, 8>: 64x16 px tile (128 accumulator registers)
//   < 64, 4, 4>: 32x16 (two workgroups per CU)                      < 64, 1, 4>:  8x16 (under-filled layers: single-tile inference)
// Larger tiles move fewer LDS-DMA bytes per MFMA (weights are shared by more pixels, the halo by more
// channels): the ablation in DESIGN.md prices each DMA stream at ~12 % of the kernel time.
template <int BN, int MT, int NW>
struct Geo {
  static constexpr int WN = BN / 64, WM = NW / WN;
  static constexpr int TH = WM * MT * 2;                       // tile rows
  static constexpr int HPIX = (TH + 2) * HS;                   // halo pixels
  static constexpr int HALO_Q = (HPIX * 2 + 63) / 64;          // wave-DMAs (1 KiB each) for the halo tile
  static constexpr int HALO_BYTES = HALO_Q * 1024;
  static constexpr int W_Q = 9 * BN * 2 / 64;                  // wave-DMAs for the weight slab
  static constexpr int TOT_Q = HALO_Q + W_Q;
  static constexpr int PER_WAVE = (TOT_Q + NW - 1) / NW;       // every wave issues exactly PER_WAVE DMAs per stage:
  static constexpr int STAGE = PER_WAVE * NW * 1024;           // slots >= TOT_Q are padding fed from the zero page
  static_assert(PER_WAVE <= 18, "at most two DMAs per tap");
  static constexpr size_t LDS = 2 * (size_t)STAGE;
  static_assert(NW % WN == 0 && (size_t)NW * 32 * 64 * 2 <= STAGE, "epilogue staging must fit one stage buffer");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// wave-uniform operands of the loader (see the kernel's "per-lane DMA slots" comment)
struct LoaderArgs {
  unsigned long long xa, x1a, wa, w_stage_bytes, zero_a;
  unsigned ld0, ld1;
  int C0, Ctot, E, lim0, lim1;
  bool hasC1;
};
// Source address of one wave-DMA, selects only.  halo / chunk are wave-uniform: the tensor of a 16-channel stage, its
// pixel stride and channel offset are computed on the scalar unit.  off32: halo slot = linear pixel index, weight slot =
// element offset inside the chunk's slab, -1 = zero page.
template <int KC_>
__device__ __forceinline__ void dma_issue(const LoaderArgs la, unsigned long long embn_a, bool halo, int chunk, int slot_c, int off32,
                                          unsigned char* lds_dst) {
  const int c0 = chunk * KC_;
  const bool second = la.hasC1 && c0 >= la.C0;
  const unsigned long long ub = halo ? (second ? la.x1a : la.xa) : la.wa + (unsigned long long)chunk * la.w_stage_bytes;
  const unsigned um = halo ? (second ? la.ld1 : la.ld0) : 1u;
  const int ua = halo ? c0 - (second ? la.C0 : 0) : 0;
  const int ulim = !halo ? 0x7fffffff : (second ? la.lim1 : la.lim0);
  const int c = c0 + slot_c;
  const int add = halo ? ua + slot_c : 0;
  const bool valid = off32 >= 0;
  unsigned long long pt = ub + 2ull * ((unsigned long long)(unsigned)off32 * um + (unsigned long long)add);
  unsigned long long pe = embn_a + 2ull * (unsigned long long)(long long)(c - la.Ctot);
#ifndef MAU_CONV_BRANCHY_LOADER
  // both candidates are materialised unconditionally: left to itself the compiler sinks the 64-bit multiply-add into an
  // exec-masked branch per DMA (s_and_saveexec + s_cbranch_execz in the middle of the MFMA stream)
  asm volatile("" : "+v"(pt), "+v"(pe));
#endif
  const bool is_t = valid & (c < ulim);
  const bool is_e = valid & halo & ((unsigned)(c - la.Ctot) < (unsigned)la.E);
  const unsigned long long src = is_t ? pt : (is_e ? pe : la.zero_a);
#ifdef MAU_CONV_ABL_ADDRONLY        // timing-only: the address arithmetic without the transfer
  asm volatile("" ::"v"(src), "v"(lds_dst));
#else
  __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)lds_dst, 16, 0, 0);
#endif
}

// timing-only ablations (scripts/build_variants.sh): results are garbage, the instruction streams are what is measured
#ifdef MAU_CONV_ABL_NODMA
constexpr bool ABL_NODMA = true;
#else
constexpr bool ABL_NODMA = false;
#endif
#ifdef MAU_CONV_ABL_NOWDMA          // timing-only: the weight slabs are never moved (what weights resident in LDS would save at most)
constexpr bool ABL_NOWDMA = true;
#else
constexpr bool ABL_NOWDMA = false;
#endif
// TIMING-ONLY probe (round 6; never in the product build): what "BatchNorm-apply + ReLU fused into the consumer's load" would cost with
// THIS loader -- an LDS-DMA has no register stage, so the affine map + ReLU would have to run as a fix-up pass over the halo image in
// LDS between "the stage has landed" and "its fragments are read": every wave rewrites its share of the (TH + 2) x 18 pixels x 16
// channels (16-byte vectors: read, unpack, fma, max, pack, write; all-zero vectors -- the padding ring -- stay zero), then one more
// workgroup barrier per stage.  Coefficients are register constants here (a real kernel reloads 32 of them per stage): a LOWER bound.
// Results are garbage (relu of the input); scripts/archive/r6_c5.sh times it.  EXPERIMENTS.md "Round 6", DESIGN.md section 8.
#ifdef MAU_CONV_PROBE_LDSFIX
constexpr bool PROBE_LDSFIX = true;
#else
constexpr bool PROBE_LDSFIX = false;
#endif
template <int VECS, int THREADS, bool F16>
__device__ __forceinline__ void probe_lds_fix(unsigned buf, int tid, float sc, float sh) {
#pragma unroll
  for (int it = 0; it < (VECS + THREADS - 1) / THREADS; ++it) {
    const int v = tid + it * THREADS;
    if (v < VECS) {
      const unsigned a = buf + (unsigned)v * 16u;
      u32x4 r = lds_read_u128<0>(a);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r));
      const bool keep = (r[0] | r[1] | r[2] | r[3]) != 0u;
      u32x4 o;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        f32x2 x = {__uint_as_float(r[d] << 16), __uint_as_float(r[d] & 0xffff0000u)};
        x = __builtin_elementwise_fma(x, f32x2{sc, sc}, f32x2{sh, sh});
        x = f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
        o[d] = keep ? pack_lp2<F16>(x) : 0u;
      }
      asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(o) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// The same wave-DMA through a buffer resource: address = base (SGPR resource) + soff (SGPR: the stage's channel / slab
// offset) + voff (per lane, constant over the stages of an item); a lane whose voff is 0xffffffff is out of range and
// receives zeros (hardware range check: the zero padding of the convolution and of the slot grid, no zero page).  One
// vector instruction per DMA instead of ~20 (64-bit multiply-add, range tests, three-way select): in this kernel VALU
// instructions cost clock, not only issue slots ("the clock is the bound").
typedef __attribute__((address_space(3))) void* lds_vptr;
__device__ __forceinline__ void dma_issue_buf(__amdgpu_buffer_rsrc_t rs, int voff, int soff, unsigned char* lds_dst) {
#ifdef MAU_CONV_ABL_ADDRONLY
  asm volatile("" ::"v"(voff), "s"(soff), "v"(lds_dst));
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)lds_dst, 16, voff, soff, 0, 0);
#endif
}

// ONE: the layer reads ONE tensor (no second source, no broadcast embedding) -- every data gradient and all but five forward launches
// of a step.  The per-DMA choice of the source then disappears from the multiply loop (it was 38 scalar compare-and-branch pairs per
// stage pair between the MFMAs: -1.5...3 % of a launch), with it the second source's lane offsets (5 registers).
// KG = 2 (K groups): the workgroup is TWO wave groups of NW waves on the SAME work item, group g multiplying the 16-channel stages
// g, g + 2, ... out of LDS buffers of its own; group 1 hands its accumulators to group 0 through LDS, group 0 runs the epilogue.
// For under-filled layers (single-tile inference at the deep levels: fewer items than half the CUs, one 4-wave workgroup per CU): a
// wave's dependent chain of 9 * Cin / 16 MFMA groups halves and every SIMD holds two waves that hide each other's latencies.
template <int BN, int MT, int NW, int EPI, bool F16, bool FAST, bool M16, bool ONE = false, int KG = 1>
__global__ __launch_bounds__(NW * KG * 64, 2) void conv3x3_bf16_kernel(ConvP p, int nPixTiles, int nCt, int nItems) {
  static_assert(!M16 || (MT == 4 && (NW == 8 || (NW == 4 && BN == 64)) && FAST), "the 16x16x32 loop: 128-pixel wave strips, buffer-addressed loader");
  static_assert(!ONE || M16, "single-source instantiations exist for the 16x16x32 variants only");
  static_assert(KG == 1 || (KG == 2 && !M16 && NW == 4 && BN == 64 && FAST && EPI != EPI_STATS), "K groups: the 4-wave 64-channel forms of the 32x32x16 loop");
  using G = Geo<BN, MT, NW>;
  constexpr int WN = G::WN, WM = G::WM, TH = G::TH, HPIX = G::HPIX, HALO_Q = G::HALO_Q, HALO_BYTES = G::HALO_BYTES;
  constexpr int TOT_Q = G::TOT_Q, STAGE = G::STAGE, PER_WAVE = G::PER_WAVE;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];   // KG x (2 * STAGE)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 1 ? 0 : wave_all / NW;          // K group of this wave; `wave` is its index INSIDE the group everywhere below
  const int wave = KG == 1 ? wave_all : wave_all % NW;
  unsigned char* const smem = smem_all + kg * (2 * STAGE);
#ifdef MAU_CONV_SETPRIO      // experiment (MI355X_MICROARCH.md, two waves per SIMD, item 4): static priority for the second-dispatched half
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int wm = wave % WM, wn = wave / WM;
  const int i32 = perm32(lane & 31), h = lane >> 5;   // tile row (pixel / channel) this lane's operands come from

  // (16-bit elements are only moved, never interpreted, outside the matrix core and the epilogue's conversion)

  // ---- work items: XCD-aware order; a persistent workgroup walks I = blockIdx.x, +gridDim.x, ...
  // (gridDim.x is a multiple of the XCD count, so a workgroup stays on one XCD's slice of the item list) ----
  // Hardware places workgroup id on XCD id % 8.  Every XCD owns a CONTIGUOUS range of pixel tiles (in image order),
  // and its workgroups walk that range together: tiles that share halo rows/columns, and the cout tiles of one pixel
  // tile, are in flight on the same XCD at about the same time, so its L2 serves the overlap instead of HBM
  // (round-robin placement of neighbouring tiles on different XCDs cost 1.17-1.28x the input bytes at level 0).
  const int xs = p.xcdShift, nx = 1 << xs;           // (3, 8 on an MI355X in SPX mode)
  const int tilesPerXcd = (nPixTiles + nx - 1) >> xs;
  auto decode = [&](int I, Item& it) -> bool {
    const int xcd = I & (nx - 1), seq = I >> xs;
    const int local = seq / nCt;
    it.pixTile = xcd * tilesPerXcd + local;
    if (local >= tilesPerXcd || it.pixTile >= nPixTiles) return false;
    it.co0 = (seq % nCt) * BN;
    int t = it.pixTile;
    const int txi = t % p.tilesX;
    t /= p.tilesX;
    const int tyi = t % p.tilesY;
    it.n = t / p.tilesY;
    it.ty0 = tyi * TH;
    it.tx0 = txi * TW;
    return true;
  };
  auto next_valid = [&](int I, Item& it) -> int {
    for (; I < nItems; I += gridDim.x)
      if (decode(I, it)) return I;
    return -1;
  };

  // ---- per-lane DMA slots: the tile-independent part.  The address of a wave-DMA is computed WITHOUT
  // control flow (selects only), so that the tap sequence of a stage stays one basic block. ----
  // Input channels come from up to two tensors and a broadcast vector (virtual concat: torch.cat([skip, up], 1) of the
  // decoder, reference src/model.py:279-282, and fuse_embeddings :248-259, never materialised):
  //   [0, C0) from x (pixel stride ldx) | [C0, C0 + C1) from x1 (ldx1) | [C0 + C1, C0 + C1 + E) from emb[n] | zero page.
  // A stage is one 16-channel chunk; with two tensors C0 % 16 == 0, so the tensor a chunk reads is wave-uniform.
  // slot j of this lane: hp = halo pixel index inside the halo tile (or -1) / weight row = tap*BN + co; c = 8 * logical
  // 16-byte half.  Recomputed where needed (once per item, and per DMA on the general path) instead of held in 2 * PER_WAVE
  // registers across the K loop.
  const size_t w_stage_stride = (size_t)9 * p.CoutPad * KC;
  const int Ctot = p.C0 + p.C1;     // tensor channels; the broadcast embedding follows
  auto slot_of = [wave, lane](int j, int& hp_or_row, int& c) {
    const int q = wave + j * NW;
    hp_or_row = -1;
    c = 0;
    if (q < HALO_Q) {
      const int slot = q * 64 + lane;
      const int hp = slot >> 1, ph = slot & 1;
      c = 8 * (M16 ? ph : ph ^ halo_swz(hp % HS));      // (M16: the lane groups of its reads are conflict-free unswizzled)
      hp_or_row = hp < HPIX ? hp : -1;
    } else if (q < TOT_Q) {
      const int slot = (q - HALO_Q) * 64 + lane;
      const int row = slot >> 1, ph = slot & 1;
      c = 8 * (M16 ? ph : ph ^ ((row >> 3) & 1));
      hp_or_row = row;
    }
  };
  auto slot_of_at = [wave](int j, int ln, int& hp_or_row, int& c) {        // (the same for a given lane index)
    const int q = wave + j * NW;
    hp_or_row = -1;
    c = 0;
    if (q < HALO_Q) {
      const int slot = q * 64 + ln;
      const int hp = slot >> 1, ph = slot & 1;
      c = 8 * (M16 ? ph : ph ^ halo_swz(hp % HS));      // (M16: the lane groups of its reads are conflict-free unswizzled)
      hp_or_row = hp < HPIX ? hp : -1;
    } else if (q < TOT_Q) {
      const int slot = (q - HALO_Q) * 64 + ln;
      const int row = slot >> 1, ph = slot & 1;
      c = 8 * (M16 ? ph : ph ^ ((row >> 3) & 1));
      hp_or_row = row;
    }
  };

  // per-lane source offsets of the item being LOADED (constant over its stages); -1 = zero page
  //   halo slot: linear pixel index (n*H + gy)*W + gx;   weight slot: element offset of the row inside a chunk's slab
  int off32[PER_WAVE];
  // buffer-addressed loader (p.fast): off32 holds BYTE offsets instead -- halo slot: pixel * (2 * ldx) + 2 * channel slot
  // inside source 0, and the same for source 1 in off1 (only the first HJ slots of a wave can be halo slots);
  // weight slots (always buffer-addressed): byte offset of the row inside a stage's slab.  -1 = out of range = zeros.
  constexpr int HJ = (HALO_Q + NW - 1) / NW;
  int off1[HJ];
  constexpr bool fast = FAST;        // (p.fast, resolved by the launcher: the two loaders never share a kernel -- with both in one
                                     //  body the general loader's hoisted 64-bit terms spilled across the K loop of <64,4,8>)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.N * p.H * p.W * p.ldx * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x1), 0, p.N * p.H * p.W * p.ldx1 * 2, 0x00020000);
  // (the broadcast embedding: base = image 0's vector, the scalar offset picks the image and the stage's channels; a lane's
  //  offset is just its 16-byte half, or -1 outside the image -- the broadcast map is zero in the padding ring)
  const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.emb_lp), 0, p.N * p.E * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.nChunks * 9 * p.CoutPad * KC * 2, 0x00020000);
  // the kernel arguments the loader needs, as plain scalars (the lambdas below must not keep the argument struct alive in memory)
  const unsigned long long xa = (unsigned long long)p.x, x1a = (unsigned long long)p.x1, wa = (unsigned long long)p.w;
  const unsigned long long zero_a = (unsigned long long)g_zero_page, emb_a = (unsigned long long)p.emb_lp;
  const int C0v = p.C0, Ev = p.E, Hv = p.H, Wv = p.W, CoutPadv = p.CoutPad;
  const bool hasC1 = p.C1 > 0;
  const unsigned ld0v = (unsigned)p.ldx, ld1v = (unsigned)p.ldx1;
  const int lim0v = (p.C1 == 0 && p.E == 0) ? p.ldx : p.C0, lim1v = p.E == 0 ? p.C0 + p.ldx1 : p.C0 + p.C1;
  unsigned long long embn_a = 0;
  int embn_off = 0;                 // byte offset of the loaded item's image inside the embedding matrix
  // The slot coordinates below (hp / 18, hp % 18, tap * CoutPad + co for up to 10 slots) are loop-invariant, and the compiler hoists
  // them out of the item loop -- 2-3 registers per slot.  Where the register file is full it then SPILLS them: 26 VGPRs in the
  // inference-epilogue instantiation of <64,4,4> (its 16 coefficient registers), 10 in <128,4,8>'s, reloaded from scratch at the
  // head and the tail of every item.  In the INFERENCE-epilogue forms the lane index is therefore made opaque once per call: nothing
  // can be hoisted, the coordinates are recomputed per item (~15 VALU per slot in the shadow of the last stage pair), no spill:
  // the 64->64 layers of the 512 x 512 inference 3-7 % faster (profiles/r5/slot_invariants_ab.txt).  Everywhere else the hoisted form
  // stays, untouched: recomputing costs the two-stage-pair items of level 0 +4-6 % (statistics epilogue), and where those forms
  // spill (8 registers in the two-source <64,4,4>) the reloads sit outside the multiply loop -- a hand-packed one-register-per-slot
  // form was measured too and is 0.3 % slower on the step than what the compiler does by itself.  (Also recomputed: the 32x32x16-loop
  // forms of the 64-wide four-row tilings -- odd stage counts at level 0, rare -- whose spilled coordinates came back BETWEEN MFMAs.)
  constexpr bool SLOT_RECOMPUTE = EPI == EPI_POST || (!M16 && BN == 64 && MT == 4);
  auto setup = [&](const Item& it) {
    embn_a = emb_a + 2ull * (unsigned long long)((long long)it.n * Ev);
    embn_off = 2 * it.n * Ev;
    int lane_o = lane;
    if constexpr (SLOT_RECOMPUTE) asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      const int q = wave + j * NW;
      off32[j] = -1;
      if (!ONE && j < HJ) off1[j < HJ ? j : 0] = -1;
      int hp_or_row, sc;
      if constexpr (SLOT_RECOMPUTE) slot_of_at(j, lane_o, hp_or_row, sc);
      else slot_of(j, hp_or_row, sc);
      if (q < HALO_Q) {
        const int hp = hp_or_row;
        if (hp >= 0) {
          const int gy = it.ty0 + hp / HS - 1, gx = it.tx0 + hp % HS - 1;
          if (gy >= 0 && gy < Hv && gx >= 0 && gx < Wv) {
            const int pix = (it.n * Hv + gy) * Wv + gx;
            off32[j] = fast ? pix * (int)(2 * ld0v) + 2 * sc : pix;
            if (!ONE && j < HJ) off1[j < HJ ? j : 0] = pix * (int)(2 * ld1v) + 2 * sc;
          }
        }
      } else if (q < TOT_Q) {
        const int row = hp_or_row;
        // (the packed rows are pre-permuted: row nt*32 + i of a 64-channel block holds output channel 2*i + nt, so the
        // two accumulator tiles of a lane carry ADJACENT channels -- pack_weights_kernel, conv3x3.hip)
        const int tap = row / BN, co = row % BN;
        off32[j] = 2 * ((tap * CoutPadv + it.co0 + co) * KC + sc);
      }
    }
  };
  // one wave-DMA (1 KiB) of a stage; j is a compile-time constant after unrolling.  Everything that depends only on
  // (slot kind, chunk) is wave-uniform and computed on the scalar unit.
  // One wave-DMA (1 KiB) of a stage: dma_issue() below, a plain function with every operand passed BY VALUE.  (As a
  // [&] lambda, "halo ? (second ? x1 : x) : w" became a load from a SELECTED field of the closure object, which pinned the
  // closure -- and with it wave, slot_c[], off32[] -- in private memory: 300 bytes of scratch traffic inside the K loop.)
  const int w_stage_bytes32 = 2 * 9 * p.CoutPad * KC;
  const LoaderArgs la = {xa, x1a, wa, 2ull * w_stage_stride, zero_a, ld0v, ld1v, C0v, Ctot, Ev, lim0v, lim1v, hasC1};
#define MAU_ISSUE_SLOT(J, STAGE_, CHUNK_)                                                                              \
  {                                                                                                                    \
    unsigned char* dst_ = smem + (STAGE_) * STAGE + (wave + (J) * NW) * 1024;                                        \
    const int c0_ = (CHUNK_) * KC;                                                                                     \
    if ((J) >= HJ || wave + (J) * NW >= HALO_Q) {                                                                      \
      if (!ABL_NOWDMA) dma_issue_buf(rs_w, off32[(J)], (CHUNK_) * w_stage_bytes32, dst_);                             \
    } else if constexpr (FAST) {                                                                                       \
      if constexpr ((J) < HJ) {                                                                                        \
        if constexpr (ONE) dma_issue_buf(rs_x, off32[(J)], 2 * c0_, dst_);                                           \
        else if (c0_ >= Ctot) dma_issue_buf(rs_e, off32[(J)] < 0 ? -1 : (off32[(J)] & 16), embn_off + 2 * (c0_ - Ctot), dst_); \
        else if (hasC1 && c0_ >= C0v) dma_issue_buf(rs_x1, off1[(J) < HJ ? (J) : 0], 2 * (c0_ - C0v), dst_);           \
        else dma_issue_buf(rs_x, off32[(J)], 2 * c0_, dst_);                                                          \
      }                                                                                                                \
    } else if constexpr ((J) < HJ) {                                                                                   \
      int hp_, sc_;                                                                                                    \
      slot_of((J), hp_, sc_);                                                                                          \
      dma_issue<KC>(la, embn_a, true, (CHUNK_), sc_, off32[(J)], dst_);                                               \
    }                                                                                                                  \
  }

  // ---- per-lane LDS read addresses: three bases (one per dx) + immediates for (mt, dy); two for the weights ----
  int abase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hx = (i32 & 15) + dx;
    abase[dx] = ((wm * MT * 2 + (i32 >> 4)) * HS + hx) * ROWB + 16 * (h ^ halo_swz(hx));
  }
  const int boff0 = HALO_BYTES + w_off(wn * 64 + i32, h);          // (tap*BN is a multiple of 16 rows: swizzle unchanged)
  const int boff1 = HALO_BYTES + w_off(wn * 64 + 32 + i32, h);

  // accumulator register r of lane half h holds MFMA row (r&3) + 8*(r>>2) + 4*h, i.e. tile row
  // perm32(that) = rowbase[r>>2] + (r&3): four per-lane bases + immediates address the whole epilogue
  int rowbase[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rowbase[g] = perm32(8 * g + 4 * h);

  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);   // LDS byte address of smem
  Item cur, nxt;
  int I = next_valid(blockIdx.x, cur);
  if (I < 0) return;                                  // whole workgroup leaves before any barrier
  setup(cur);
  static_for<0, PER_WAVE>([&](auto jc) { MAU_ISSUE_SLOT(decltype(jc)::value, 0, kg); });      // (group g starts with stage g)
  int stage = 0;
  unsigned short* __restrict__ yg = reinterpret_cast<unsigned short*>(p.y);

  bool stores_behind = false;      // (wave-uniform) the previous item's epilogue issued exactly its 4 * MT output stores
  while (true) {
    const int In = next_valid(I + gridDim.x, nxt);
    f32x16 acc[M16 ? 1 : MT][2];
    f32x4v acc16[M16 ? 8 : 1][4];
    if constexpr (M16) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc16[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;   // (bias-initialised accumulators make the MT = 4 variants spill 100 registers)
    }

    // ---- K loop.  Between two stages: wait for this wave's DMAs -> raw barrier (every wave's DMAs have landed and
    // everyone is done with the buffer that is refilled next); inside a stage: issue the following stage, which may
    // already belong to the NEXT work item (cross-tile pipelining), and multiply. ----
    // The first stage's wait stands BEFORE the loop: after an interior item's epilogue the 4 * MT output stores (and the
    // statistics row) were issued AFTER the stage's DMAs; the vector-memory counter retires in issue order, so waiting until
    // at most 4 * MT operations are outstanding covers the DMAs without waiting out the stores' HBM write round trip
    // (with vmcnt(0) every item of a short-K layer paid it).  (A flag tested at the loop head made the compiler peel the
    // first iteration, with register reloads and full vmcnt(0) drains inside the peeled copy.)
#ifndef MAU_CONV_ABL_NOWAIT
#ifndef MAU_CONV_NO_COUNTED_EPI
    if (stores_behind) wait_vmcnt<4 * MT>();
    else
#endif
      wait_vmcnt<0>();
#endif
    __builtin_amdgcn_s_barrier();
    if constexpr (M16) {
      using PS = PairSched;
      static_assert(kPairSched.ok(), "pair schedule");
      // a wave-DMA every DSP MFMAs of a half pair: <*,4,8> issue up to 7 per stage, <64,4,4> (half the waves for 0.68 of the bytes) 10
      constexpr int DSP = PER_WAVE <= 7 ? 14 : 12;
      static_assert(PER_WAVE <= 10 && 10 + (PER_WAVE - 1) * DSP < 128, "DMA slots of the pair schedule");
      const bool up = lane >= 32;
      const unsigned laneA = ((wm * 8) * HS + (lane & 15)) * ROWB + 16 * ((lane >> 4) & 1);
      const unsigned laneB = HALO_BYTES + (wn * 64 + (lane & 15)) * ROWB + 16 * ((lane >> 4) & 1);
      for (int chunk = 0; chunk < p.nChunks; chunk += 2) {       // (the launcher sends only even stage counts here)
        const bool moreB = chunk + 2 < p.nChunks;
        const int fchunkB = moreB ? chunk + 2 : 0;
        const unsigned sA = lds0 + stage * STAGE, sB = lds0 + (stage ^ 1) * STAGE;
        if constexpr (PROBE_LDSFIX) {                           // (timing probe: stage A's halo image, published by the barrier above)
          float psc = 1.0009765625f, psh = 0.0009765625f;
          asm volatile("" : "+v"(psc), "+v"(psh));
          probe_lds_fix<HPIX * 2, NW * 64, F16>(sA, tid, psc, psh);
        }
        bf16x8 fa[PS::RA] = {}, fb[PS::RB] = {};
        unsigned aAddr = 0, bAddr = 0;
        auto issue_read = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          constexpr PS::RD R = kPairSched.rd[k];
          if constexpr (!R.isA && R.idx == 0) {                   // first read of step s: its two taps' addresses
            constexpr int st = R.s;
            constexpr int tlo = st <= 3 ? 2 * st : st == 4 ? 8 : 2 * (st - 5) + 1;
            constexpr int thi = st <= 3 ? 2 * st + 1 : st == 4 ? 0 : 2 * (st - 5) + 2;
            const unsigned blo = st <= 4 ? sA : sB, bhi = st <= 3 ? sA : sB;
            aAddr = laneA + (up ? bhi + ((thi / 3) * HS + thi % 3) * ROWB : blo + ((tlo / 3) * HS + tlo % 3) * ROWB);
            bAddr = laneB + (up ? bhi + thi * BN * ROWB : blo + tlo * BN * ROWB);
          }
#ifndef MAU_CONV_ABL_NOREAD
          if constexpr (R.isA) fa[R.seq % PS::RA] = lds_read128<R.idx * HS * ROWB>(aAddr);
          else fb[R.seq % PS::RB] = lds_read128<R.idx * 16 * ROWB>(bAddr);
#endif
        };
        static_for<0, kPairSched.lo[0]>([&](auto kc) { issue_read(kc); });
        static_for<0, PS::NM>([&](auto Mc) {
          constexpr int M = decltype(Mc)::value, st = M / 32, y = (M % 32) / 4, n = M % 4;
          constexpr int kA = kPairSched.ka[st][y], kB = kPairSched.kb[st][n];
          constexpr int as = kPairSched.rd[kA].seq % PS::RA, bs = kPairSched.rd[kB].seq % PS::RB;
          if constexpr (kPairSched.rd[kA].need == M || kPairSched.rd[kB].need == M) {
            constexpr int kmax = kA > kB ? kA : kB;
            constexpr int allowed = kPairSched.lo[M] - (kmax + 1);
            static_assert(allowed >= 0 && allowed <= 15, "lgkmcnt range");
            landed<allowed>(fa[as], fb[bs]);
          }
          mfma32k<F16>(fa[as], fb[bs], acc16[y][n]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (M == PS::B1) {                              // stage B's buffer: every wave's DMAs have landed
#ifndef MAU_CONV_ABL_NOWAIT          // (timing-only: how much of the kernel is WAITING for the DMAs, as opposed to issuing them)
            wait_vmcnt<0>();
#endif
            __builtin_amdgcn_s_barrier();
            if constexpr (PROBE_LDSFIX) {                         // (timing probe: stage B's halo image, just published)
              float psc = 1.0009765625f, psh = 0.0009765625f;
              asm volatile("" : "+v"(psc), "+v"(psh));
              probe_lds_fix<HPIX * 2, NW * 64, F16>(sB, tid, psc, psh);
            }
          }
          if constexpr (M == PS::B2) {                              // stage A's buffer: everyone is done reading it
            __builtin_amdgcn_s_barrier();
            if (!moreB && In >= 0) setup(nxt);
          }
          if constexpr (kPairSched.lo[M + 1] > kPairSched.lo[M]) {
            static_for<kPairSched.lo[M], kPairSched.lo[M + 1]>([&](auto kc) { issue_read(kc); });
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (M < 128 && M % DSP == 10 && M / DSP < PER_WAVE && !ABL_NODMA) {
            MAU_ISSUE_SLOT(M / DSP, stage ^ 1, chunk + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (M >= 160 && (M - 160) % DSP == 10 && (M - 160) / DSP < PER_WAVE && !ABL_NODMA) {
            MAU_ISSUE_SLOT((M - 160) / DSP, stage, fchunkB);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        if (moreB) {
#ifndef MAU_CONV_ABL_NOWAIT
          wait_vmcnt<0>();
#endif
          __builtin_amdgcn_s_barrier();
        }
      }
    } else
    for (int chunk = kg; chunk < p.nChunks; chunk += KG) {        // (KG = 2: the launcher sends only even stage counts -- same barriers in both groups)
      // next stage: the following chunk of this item, or chunk 0 of the next item (cross-tile pipelining);
      // after the very last stage chunk 0 of the current item is re-fetched into the idle buffer (nobody reads it)
      const bool more = chunk + KG < p.nChunks;
      const int fchunk = more ? chunk + KG : kg;
      if (!more && In >= 0) setup(nxt);
      // Operand fragments are double-buffered in registers: the ds_reads of tap t+1 are issued BEFORE the
      // MFMAs of tap t, so the only LDS round trip a wave waits out is the first one of a stage (timing
      // ablation, DESIGN.md: with the reads removed the same loop runs 1.36x faster).
      // The following stage's DMAs are issued ONE PER TAP, after the MFMA group: a wave-DMA costs its issuing
      // wave 60-185 cycles (MI355X_MICROARCH.md); in a burst at the head of the stage all eight waves would
      // pay that with the MFMA pipes idle, spread out the SIMD's other wave multiplies meanwhile.
      const unsigned sbase = lds0 + stage * STAGE;
      const unsigned b0a = sbase + boff0, b1a = sbase + boff1;
      const unsigned aa[3] = {sbase + abase[0], sbase + abase[1], sbase + abase[2]};
#ifdef MAU_CONV_DMA_PER_TAP
      constexpr int DPT = MAU_CONV_DMA_PER_TAP;
#else
      constexpr int DPT = PER_WAVE > 9 ? 2 : 1;
#endif
      // The stage's NM MFMAs in StageSched order; between two MFMAs of the wave (an MFMA holds the vector issue port for
      // 8 of its 32 cycles) go the fragment reads that are LEAD MFMAs from their first use and, every NM / 9 MFMAs, one
      // wave-DMA of the following stage (a wave-DMA costs its issuing wave 60-185 cycles, MI355X_MICROARCH.md: in a burst
      // at the head of the stage all eight waves would pay that with the MFMA pipes idle; measured: 2, 3, 9 per slot are
      // 3-5 % slower than 1).
      using SS = StageSched<MT>;
      static_assert(kStageSched<MT>.ok(), "stage schedule");
      bf16x8 fa[SS::RING] = {}, fb[SS::BRING] = {};
      auto issue_read = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr typename SS::RD R = kStageSched<MT>.rd[k];
#ifndef MAU_CONV_ABL_NOREAD
        if constexpr (R.isA) fa[R.seq % SS::RING] = lds_read128<R.r * HS * ROWB>(aa[R.g]);
        else if constexpr (R.nt == 0) fb[R.seq % SS::BRING] = lds_read128<(R.dy * 3 + R.g) * BN * ROWB>(b0a);
        else fb[R.seq % SS::BRING] = lds_read128<(R.dy * 3 + R.g) * BN * ROWB>(b1a);
#endif
      };
      static_for<0, kStageSched<MT>.lo[0]>([&](auto kc) { issue_read(kc); });            // the stage head: everything needed within LEAD
      static_for<0, SS::NM>([&](auto Mc) {
        constexpr int M = decltype(Mc)::value;
        constexpr typename SS::MF f = kStageSched<MT>.mf[M];
        constexpr int as = kStageSched<MT>.rd[f.ka].seq % SS::RING, bs = kStageSched<MT>.rd[f.kb].seq % SS::BRING;
        if constexpr (kStageSched<MT>.rd[f.ka].need == M || kStageSched<MT>.rd[f.kb].need == M) {
          constexpr int kmax = f.ka > f.kb ? f.ka : f.kb;
          constexpr int allowed = kStageSched<MT>.lo[M] - (kmax + 1);
          static_assert(allowed >= 0 && allowed <= 15, "lgkmcnt range");
          landed<allowed>(fa[as], fb[bs]);
        }
        acc[f.mt][f.nt] = mfma16<F16>(fa[as], fb[bs], acc[f.mt][f.nt]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kStageSched<MT>.lo[M + 1] > kStageSched<MT>.lo[M]) {
          static_for<kStageSched<MT>.lo[M], kStageSched<MT>.lo[M + 1]>([&](auto kc) { issue_read(kc); });
          __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int SLOT = SS::NM / 9;
        if constexpr (M % SLOT == SLOT - 2 && (M / SLOT) * DPT < PER_WAVE && !ABL_NODMA) {
          static_for<0, DPT>([&](auto rc) {
            constexpr int J = (M / SLOT) * DPT + decltype(rc)::value;
            if constexpr (J < PER_WAVE) MAU_ISSUE_SLOT(J, stage ^ 1, fchunk);
          });
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      stage ^= 1;
      if (more) {                                      // (after the last stage the epilogue's barrier takes this place)
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
    }

    // ---- epilogue of `cur` (the next item's first stage is already in flight into buffer `stage`) ----
    __builtin_amdgcn_s_barrier();                      // every wave is done reading buffer stage^1 -> reuse it
    const bool full = cur.ty0 + TH <= p.H && cur.tx0 + TW <= p.W;     // interior tiles (the common case): mask-free, workgroup-uniform
    if constexpr (KG == 2) {
      // group 1 -> group 0: MT * 32 accumulator registers per wave = MT * 32 KB, through the two groups' idle stage^1 buffers (registers
      // 4 j .. 4 j + 3 of wave w: the 1 KB block (w * R4 + j) of the area [group 1's buffer | group 0's buffer]); fixed order: group 0 + group 1.
      constexpr int R4 = MT * 8;
      static_assert((size_t)NW * R4 * 1024 <= 2 * (size_t)STAGE && STAGE % 1024 == 0, "accumulator hand-over area");
      const unsigned lds_all = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_all);
      const unsigned b1 = lds_all + (2 + (stage ^ 1)) * STAGE, b0 = lds_all + (stage ^ 1) * STAGE;
      auto slot = [&](int j) -> unsigned {
        const unsigned o = (unsigned)((wave * R4 + j) * 1024);
        return (o < (unsigned)STAGE ? b1 + o : b0 + (o - (unsigned)STAGE)) + lane * 16;
      };
      if (kg == 1) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
              lds_write_f32x4(slot((a * 2 + b) * 4 + g), v);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (kg == 0) {
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          f32x4 t[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = lds_read_f32x4(slot(a * 8 + j));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][j >> 2][4 * (j & 3) + e] += t[j][e];
        }
      }
      __builtin_amdgcn_s_barrier();                    // the hand-over area overlaps OTHER waves' epilogue staging rows
    }
    if (KG == 1 || kg == 0) {
    // wave-private staging image: 32 pixels x 64 channels bf16 (128-byte rows)
    const unsigned stg = lds0 + (stage ^ 1) * STAGE + wave * (32 * 64 * 2);
    const unsigned rbase = stg + (lane >> 3) * 128 + (lane & 7) * 16;   // read: pixel pass*8 + lane/8, 16-byte vector lane%8
    const int cv = cur.co0 + wn * 64 + (lane & 7) * 8;
    // MaxPool2d(2,2) of the activation (reference src/model.py:218,268-271; inference epilogue) from the SAME registers a tile-row group
    // is stored from: a group is two pixel rows of 16, ov[pass] = row pass / 2, pixel x = 8 (pass % 2) + lane / 8, 8 channels per lane
    // -- the row partner of a pixel is the same lane two passes on, its column partner the lane 8 further (DPP row_ror:8 inside the
    // 16-lane row).  The activation is relu(...) >= 0, and the bit patterns of non-negative bf16 / fp16 values order like signed 16-bit
    // integers (-0 below everything): v_pk_max_i16 on the packed pairs IS the maximum, bit for bit what mau_maxpool2x2_fwd computes
    // from the stored tensor.  ybase (the group's first row) is even; floor mode: windows that leave the image do not exist.
    auto pool_group = [&](const u32x4 (&ov)[4], int ybase, bool whole) {
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      auto mx = [](unsigned a, unsigned b) -> unsigned {
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
      };
      const int Ho = p.H >> 1, Wo = p.W >> 1;
      const int yq = ybase >> 1;
      unsigned char* const pbytes = reinterpret_cast<unsigned char*>(p.pool);
      int lane_p = lane;                       // (opaque per group: nothing below may be hoisted out of the item loop into a
      asm volatile("" : "+v"(lane_p));         //  register that lives across the multiply loop -- the file is full at 256)
#pragma unroll
      for (int half = 0; half < 2; ++half) {                                        // pixels x = 0..7 | 8..15 of the two rows
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned m = mx(ov[half][d], ov[half + 2][d]);
          v[d] = mx(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, false));     // row_ror:8 -> lane ^ 8
        }
        const int xq = (cur.tx0 >> 1) + half * 4 + (lane_p >> 4);
        const bool mine = ((lane_p >> 3) & 1) == 0 && cur.co0 + wn * 64 + (lane_p & 7) * 8 < p.ldpool && (whole || (yq < Ho && xq < Wo));
        if (mine)
          *reinterpret_cast<u32x4*>(pbytes + ((((size_t)cur.n * Ho + yq) * Wo + xq) * p.ldpool + cur.co0 + wn * 64 + (lane_p & 7) * 8) * 2) = v;
      }
    };
    if constexpr (M16) {
#ifdef MAU_CONV_ABL_NOEPI            // timing-only: the accumulators are consumed, nothing is staged, counted or stored
      float t = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) t += acc16[a][b][0] + acc16[a][b][1] + acc16[a][b][2] + acc16[a][b][3];
      if (t == 1234.5f) yg[lane] = 1;
#else
      // lane (q = lane / 16, c = lane % 16): pixels x = 4q + r of row y, channel column c of tile t; the packed rows put
      // channel 32 (t & 1) + 2c + (t >> 1) there, so tiles (t, t + 2) of a lane are an adjacent channel pair
      const int q4 = (lane >> 4) * 4, c16 = lane & 15;
      const int cb = cur.co0 + wn * 64 + 2 * c16;               // + 32 (t & 1) + (t >> 1)
      const unsigned wb16 = stg + q4 * 128 + c16 * 4;
      // Everything per value is PACKED fp32 math on the register pairs the accumulator tiles already are -- (r = 0, 1) and
      // (r = 2, 3) of one tile: bias and the two statistics cost 6 v_pk_* per tile row instead of 12 scalar operations, and
      // interior tiles (FULL, the common case) carry no masks.  (One body with the run-time `full` inside made every value a
      // select: 379 v_cndmask + 224 v_mov per item on top of the arithmetic -- more VALU issue than the item's 576 MFMAs'
      // shadows hide at level 0, where the epilogue comes every four stages.)
      float bb[4];
      f32x2 pscv[4], pshv[4], s2[4], q2[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ch = cb + 32 * (t & 1) + (t >> 1);
        const float bch = (p.bias != nullptr && ch < p.Cout) ? p.bias[ch] : 0.f;
        bb[t] = bch;
        s2[t] = f32x2{0.f, 0.f};
        q2[t] = f32x2{0.f, 0.f};
        if (EPI == EPI_POST) {
          const float sc = ch < p.Cout ? p.post_scale[ch] : 0.f, sh = ch < p.Cout ? p.post_shift[ch] : 0.f;
          pscv[t] = f32x2{sc, sc};
          pshv[t] = f32x2{sh, sh};
        }
      }
      const int xrem = p.W - cur.tx0 - q4;                       // pixel x = q4 + r is inside the image iff r < xrem
      // The four tile-row groups (mt) of a wave go through ONE 4 KB staging image; DS operations of a wave execute in order, so
      // group mt's writes may follow group mt - 1's reads without a wait -- only the global stores need the data those reads
      // return.  The groups are therefore software-pipelined: convert + write(mt), THEN wait for read(mt - 1) (16 writes were issued
      // behind it: "at most 15 LDS operations outstanding" = it has landed), store(mt - 1), read(mt).  The LDS round trip of a group
      // hides behind the next group's arithmetic; with everything waited in place each item paid four of them with all eight waves of
      // the workgroup -- and so the MFMA pipes -- idle.
      // Output addresses: (uniform 64-bit tile base) + (per-lane 32-bit offset, constant over the kernel) -- the compiler keeps the
      // base in SGPRs (global_store saddr form) instead of 7 VALU instructions of 64-bit arithmetic per store.
      const unsigned lane_off = (unsigned)(((lane >> 3) * p.ldy + (lane & 7) * 8) * 2);
      unsigned char* const ybytes = reinterpret_cast<unsigned char*>(yg);
      auto rows16 = [&](auto fullc, auto biasc) {
        constexpr bool FULL = decltype(fullc)::value, BIAS = decltype(biasc)::value;
        u32x4 o[2][4];
        auto store_group = [&](int mt, const u32x4 (&ov)[4]) {
#ifdef MAU_CONV_ABL_NOSTORE
          if (ov[0][0] + ov[1][1] + ov[2][2] + ov[3][3] == 0x12345678u) yg[lane] = 1;
#else
          const int ybase = cur.ty0 + wm * 8 + mt * 2;
          if (cv < p.ldy) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
              const int gyu = ybase + (pass >> 1), gxu = cur.tx0 + (pass & 1) * 8;        // (uniform part of the pixel)
              const size_t uni = (((size_t)(cur.n * p.H + gyu) * p.W + gxu) * p.ldy + cur.co0 + wn * 64) * 2;
              if (FULL || (gyu < p.H && gxu + (lane >> 3) < p.W))
                __builtin_nontemporal_store(ov[pass], reinterpret_cast<u32x4*>(ybytes + uni + lane_off));
            }
          }
          if constexpr (EPI == EPI_POST && ONE) {         // (single-source forms only: an encoder block's second convolution)
            if (p.pool != nullptr) pool_group(ov, ybase, FULL);
          }
#endif
        };
        static_for<0, 4>([&](auto mc) {
          constexpr int mt = decltype(mc)::value;
          const int ybase = cur.ty0 + wm * 8 + mt * 2;
          static_for<0, 2>([&](auto yc) {
            constexpr int yy = decltype(yc)::value;
            const bool yok = FULL || ybase + yy < p.H;
            f32x2 lo[4], hi[4];
            static_for<0, 4>([&](auto tc) {
              constexpr int t = decltype(tc)::value;
              const f32x4v v = acc16[mt * 2 + yy][t];
              lo[t] = f32x2{v[0], v[1]};
              hi[t] = f32x2{v[2], v[3]};
              if (BIAS) {
                lo[t] += bb[t];
                hi[t] += bb[t];
              }
              if (EPI == EPI_POST) {
                lo[t] = __builtin_elementwise_max(__builtin_elementwise_fma(lo[t], pscv[t], pshv[t]), f32x2{0.f, 0.f});
                hi[t] = __builtin_elementwise_max(__builtin_elementwise_fma(hi[t], pscv[t], pshv[t]), f32x2{0.f, 0.f});
              }
#ifndef MAU_CONV_ABL_NOSTATS
              if (EPI == EPI_STATS) {
                f32x2 ml = lo[t], mh = hi[t];
                if (!FULL) {
                  ml = f32x2{yok && 0 < xrem ? ml[0] : 0.f, yok && 1 < xrem ? ml[1] : 0.f};
                  mh = f32x2{yok && 2 < xrem ? mh[0] : 0.f, yok && 3 < xrem ? mh[1] : 0.f};
                }
                s2[t] += ml;
                s2[t] += mh;
                q2[t] = __builtin_elementwise_fma(ml, ml, q2[t]);
                q2[t] = __builtin_elementwise_fma(mh, mh, q2[t]);
              }
#endif
            });
            static_for<0, 2>([&](auto nc) {
              constexpr int n = decltype(nc)::value;
              lds_write_b32<(yy * 16 + 0) * 128 + 64 * n>(wb16, pack_lp2<F16>(f32x2{lo[n][0], lo[n + 2][0]}));
              lds_write_b32<(yy * 16 + 1) * 128 + 64 * n>(wb16, pack_lp2<F16>(f32x2{lo[n][1], lo[n + 2][1]}));
              lds_write_b32<(yy * 16 + 2) * 128 + 64 * n>(wb16, pack_lp2<F16>(f32x2{hi[n][0], hi[n + 2][0]}));
              lds_write_b32<(yy * 16 + 3) * 128 + 64 * n>(wb16, pack_lp2<F16>(f32x2{hi[n][1], hi[n + 2][1]}));
            });
            // (pin the running sums here: the stores are basic-block boundaries, and LLVM's sinking pass otherwise moves the whole
            //  statistics chain behind the last of them -- all 128 biased values stay live, ~90 registers spill)
            if (EPI == EPI_STATS)
              asm volatile("" : "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]), "+v"(q2[0]), "+v"(q2[1]), "+v"(q2[2]), "+v"(q2[3]));
          });
          if constexpr (mt > 0) {
            lds_land_behind16(o[(mt - 1) & 1][0], o[(mt - 1) & 1][1], o[(mt - 1) & 1][2], o[(mt - 1) & 1][3]);
            store_group(mt - 1, o[(mt - 1) & 1]);
          }
          o[mt & 1][0] = lds_read_u128<0 * 1024>(rbase);
          o[mt & 1][1] = lds_read_u128<1 * 1024>(rbase);
          o[mt & 1][2] = lds_read_u128<2 * 1024>(rbase);
          o[mt & 1][3] = lds_read_u128<3 * 1024>(rbase);
        });
        lds_land(o[1][0], o[1][1], o[1][2], o[1][3]);
        store_group(3, o[1]);
      };
      if (p.bias != nullptr) {                         // (workgroup-uniform; the data gradient has no bias to add)
        if (full) rows16(std::true_type{}, std::true_type{});
        else rows16(std::false_type{}, std::true_type{});
      } else {
        if (full) rows16(std::true_type{}, std::false_type{});
        else rows16(std::false_type{}, std::false_type{});
      }
      if (EPI == EPI_STATS) {
        // a lane's pair sums -> the channel's sum over the wave's 128 pixels: the lanes q = 0..3 of a column c hold the rest
        float* srow = p.slab + ((size_t)cur.pixTile * WM + wm) * 2 * p.CoutPad + cb;
        float st[4], qt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          st[t] = s2[t][0] + s2[t][1];
          qt[t] = q2[t][0] + q2[t][1];
          st[t] += __shfl_xor(st[t], 16);
          st[t] += __shfl_xor(st[t], 32);
          qt[t] += __shfl_xor(qt[t], 16);
          qt[t] += __shfl_xor(qt[t], 32);
        }
        if (lane < 16) {
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const f32x2 sv = {st[n], st[n + 2]}, qv = {qt[n], qt[n + 2]};
            *reinterpret_cast<f32x2*>(srow + 32 * n) = sv;
            *reinterpret_cast<f32x2*>(srow + 32 * n + p.CoutPad) = qv;
          }
        }
      }
#endif   // MAU_CONV_ABL_NOEPI
    } else {
    unsigned wbase[4];                                 // write: row rowbase[g] + k, channel pair 2*i32 (one dword)
#pragma unroll
    for (int g = 0; g < 4; ++g) wbase[g] = stg + rowbase[g] * 128 + i32 * 4;
    // lane i32, accumulator tile nt <-> output channel 2*i32 + nt of the wave's 64 (see pack_weights_kernel)
    const int cpair = cur.co0 + wn * 64 + 2 * i32;
    float bv[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bv[nt] = (p.bias != nullptr && cpair + nt < p.Cout) ? p.bias[cpair + nt] : 0.f;
    float s[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};      // statistics of the channel pair
    float psc[2] = {0.f, 0.f}, psh[2] = {0.f, 0.f};
    if (EPI == EPI_POST) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        psc[nt] = cpair + nt < p.Cout ? p.post_scale[cpair + nt] : 0.f;
        psh[nt] = cpair + nt < p.Cout ? p.post_shift[cpair + nt] : 0.f;
      }
    }
    int xlim[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xlim[g] = p.W - cur.tx0 - (rowbase[g] & 15);
#ifdef MAU_CONV_ABL_NOEPI
    {
      float t = 0.f;
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[a][b][r];
      if (t == 1234.5f) yg[lane] = 1;
    }
#else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ybase = cur.ty0 + wm * MT * 2 + mt * 2;
      auto stage_rows = [&](auto fullc, auto biasc) {
        constexpr bool FULL = decltype(fullc)::value, BIAS = decltype(biasc)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bool yok = FULL || ybase + (rowbase[g] >> 4) < p.H;
          static_for<0, 4>([&](auto ic) {
            constexpr int k = decltype(ic)::value;
            // (scalar f32 math on purpose: packed v_pk_* needs the pair in adjacent registers, which made the
            //  128-accumulator variants spill 100 more registers)
            float v0 = acc[mt][0][g * 4 + k], v1 = acc[mt][1][g * 4 + k];
            if (BIAS) {
              v0 += bv[0];
              v1 += bv[1];
            }
            if (EPI == EPI_POST) {
              v0 = fmaxf(fmaf(v0, psc[0], psh[0]), 0.f);
              v1 = fmaxf(fmaf(v1, psc[1], psh[1]), 0.f);
            }
            if (EPI == EPI_STATS && (FULL || (yok && k < xlim[g]))) {
              s[0] += v0;
              s[1] += v1;
              q2[0] = fmaf(v0, v0, q2[0]);
              q2[1] = fmaf(v1, v1, q2[1]);
            }
            const f32x2 v = {v0, v1};
            lds_write_b32<k * 128>(wbase[g], pack_lp2<F16>(v));
          });
        }
      };
      if (p.bias != nullptr) {                         // (workgroup-uniform; the data gradient has no bias to add)
        if (full) stage_rows(std::true_type{}, std::true_type{});
        else stage_rows(std::false_type{}, std::true_type{});
      } else {
        if (full) stage_rows(std::true_type{}, std::false_type{});
        else stage_rows(std::false_type{}, std::false_type{});
      }
      // whole 128-byte rows leave as 16-byte vectors.  DS operations of one wave execute in order, so the reads
      // follow the writes without a barrier; the four global stores are issued back to back, nothing waits for them.
      u32x4 o0 = lds_read_u128<0 * 1024>(rbase), o1 = lds_read_u128<1 * 1024>(rbase);
      u32x4 o2 = lds_read_u128<2 * 1024>(rbase), o3 = lds_read_u128<3 * 1024>(rbase);
      lds_land(o0, o1, o2, o3);
      const u32x4 ov[4] = {o0, o1, o2, o3};
      if (cv < p.ldy) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int prow = pass * 8 + (lane >> 3);
          const int gy = ybase + (prow >> 4), gx = cur.tx0 + (prow & 15);
          if (full || (gy < p.H && gx < p.W))
            __builtin_nontemporal_store(ov[pass], reinterpret_cast<u32x4*>(yg + ((size_t)(cur.n * p.H + gy) * p.W + gx) * p.ldy + cv));
        }
      }
      if constexpr (EPI == EPI_POST) {
        if (p.pool != nullptr) pool_group(ov, ybase, full);
      }
    }
#endif
    if (EPI == EPI_STATS) {
      // one slab row per (pixel tile, wave row): no cross-wave reduction, hence no barrier and no LDS round trip here
      float* srow = p.slab + ((size_t)cur.pixTile * WM + wm) * 2 * p.CoutPad + cpair;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        s[nt] += __shfl_xor(s[nt], 32);
        q2[nt] += __shfl_xor(q2[nt], 32);
      }
      if (h == 0) {
        const f32x2 sv = {s[0], s[1]}, qv = {q2[0], q2[1]};
        *reinterpret_cast<f32x2*>(srow) = sv;
        *reinterpret_cast<f32x2*>(srow + p.CoutPad) = qv;
      }
    }
    }   // (!M16)
    }   // (KG == 1 || kg == 0)
    if (In < 0) break;
    stores_behind = (KG == 1 || kg == 0) && full && cur.co0 + wn * 64 + 64 <= p.ldy;     // every lane stored, 4 * MT store instructions per wave
    cur = nxt;
    I = In;
  }
  wait_vmcnt<0>();                                     // the idle re-fetch of the last stage must land before the LDS is released
#undef MAU_ISSUE_SLOT
}

#ifdef MAU_CONV_DEV_ONE               // development aid: compile ONE instantiation (register / ISA checks in seconds, not minutes)
template __global__ void conv3x3_bf16_kernel<MAU_CONV_DEV_ONE>(ConvP, int, int, int);
}  // namespace v2
}  // namespace mau
#else
template <int BN, int MT, int NW, int EPI, bool F16>
static int launch(const ConvP& p, hipStream_t st) {
  using G = Geo<BN, MT, NW>;
  MAU_LDS_ATTR(G::LDS, &conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, false>);
  MAU_LDS_ATTR(G::LDS, &conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, false, false>);
  if constexpr (MT == 4) {
    MAU_LDS_ATTR(G::LDS, &conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, true>);
    MAU_LDS_ATTR(G::LDS, (&conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, true, true>));
  }
  constexpr bool HAS_KG = has_k_groups(BN, MT, NW) && EPI != EPI_STATS;       // the under-filled-layer forms <64,1,4> / <64,2,4>
  if constexpr (HAS_KG) MAU_LDS_ATTR(2 * G::LDS, (&conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, false, false, 2>));
  const DeviceShape ds = device_shape();
  const int tilesX = ceil_div(p.W, TW), tilesY = ceil_div(p.H, G::TH);
  ConvP q = p;
  q.tilesX = tilesX;
  q.tilesY = tilesY;
  q.nChunks = ceil_div(p.C0 + p.C1 + p.E, KC);
  // buffer-addressed halo loads: every 16-channel stage inside one tensor's (zero-padded) pixel row, 31-bit byte offsets
  const long long px = (long long)p.N * p.H * p.W;
  // (with a broadcast embedding behind the tensors: the tensors end on a stage boundary and the embedding fills whole stages;
  //  the lane offset of an embedding stage is derived from the source-0 offset: its 16-byte-half bit needs 2 * ldx % 32 == 0)
  const bool emb_ok = p.E == 0 || (p.emb_lp != nullptr && p.E % KC == 0 && (p.C0 + p.C1) % KC == 0 && p.ldx % KC == 0);
  const bool one = p.C1 == 0 && p.ldx % KC == 0 && round_up(p.C0, KC) <= p.ldx;
  const bool two = p.C1 > 0 && p.C0 % KC == 0 && p.C0 <= p.ldx && round_up(p.C1, KC) <= p.ldx1;
  q.fast = emb_ok && (one || two) && px * p.ldx * 2 < (1ll << 31) && px * p.ldx1 * 2 < (1ll << 31) ? 1 : 0;
  if ((long long)q.nChunks * 9 * p.CoutPad * KC * 2 >= (1ll << 31)) {
    set_error("conv3x3_fwd: packed weights beyond 2 GiB");
    return MAU_ERR_ARG;
  }
  const int nPixTiles = p.N * tilesX * tilesY;
  const int nCt = p.CoutPad / BN;
  // fewer pixel tiles than XCDs (single-tile inference at the deep levels): "every XCD owns a range of pixel tiles" would leave XCDs
  // without work -- the items go out in plain order instead (item I = (pixel tile I / nCt, cout tile I % nCt) on XCD I % 8)
  const bool spread = nPixTiles < ds.xcds;
  const int xcds = spread ? 1 : ds.xcds;
  q.xcdShift = spread ? 0 : ds.xcd_shift;
  const int nItems = round_up(nPixTiles, xcds) * nCt;
  // persistent: as many workgroups as fit the chip at once
  const int per_cu = (int)((160 * 1024) / G::LDS) >= 2 && NW == 4 ? 2 : 1;
  // (launch_cus(): the device's CUs, or the calling thread's budget -- mau_set_cu_budget)
  int grid = launch_cus() / ds.xcds * ds.xcds * per_cu;      // a multiple of the XCD count, like nItems: a workgroup stays on its XCD's slice
  if (grid > nItems) grid = nItems;
  // the 16x16x32 loop walks stages in pairs: big-tile variants, buffer-addressed loader, an even number of stages
  // (MAU_CONV_M16=0: the 32x32x16 loop everywhere -- the exact big-tile tests run both loops on the production tilings)
  static const bool m16 = getenv("MAU_CONV_M16") == nullptr || atoi(getenv("MAU_CONV_M16")) != 0;
  bool done = false;
  // MaxPool2d(2,2) of the activation (inference, one tensor source): written by the epilogue from the registers it stores; anything else
  // runs the convolution as it is and mau_maxpool2x2_fwd behind it (same bits)
  const bool pool_behind = q.pool != nullptr && !(EPI == EPI_POST && p.C1 == 0 && p.E == 0);
  if (pool_behind) q.pool = nullptr;
  if constexpr (MT == 4) {
    if (m16 && q.fast && q.nChunks % 2 == 0) {
      if (p.C1 == 0 && p.E == 0)
        MAU_LAUNCH((conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, true, true>), dim3(grid), dim3(NW * 64), G::LDS, st, q, nPixTiles, nCt, nItems);
      else
        MAU_LAUNCH((conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, true>), dim3(grid), dim3(NW * 64), G::LDS, st, q, nPixTiles, nCt, nItems);
      done = true;
    }
  }
  if constexpr (HAS_KG) {
    // Two K groups per workgroup where the layer leaves every CU at most ONE 4-wave workgroup (single-tile inference at the deep
    // levels): the stages are dealt to two wave groups, a wave's chain of dependent MFMA groups halves and each SIMD
    // holds two waves.  Even stage counts only (same barriers in both groups); MAU_CONV_KG=0 switches it off (A/B).
    if (!done && q.fast && k_groups_rule(q.nChunks, nItems)) {
      MAU_LAUNCH((conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, false, false, 2>), dim3(nItems), dim3(NW * 2 * 64), 2 * G::LDS, st, q, nPixTiles, nCt, nItems);
      done = true;
    }
  }
  if (done) {
  } else if (q.fast) {
    MAU_LAUNCH((conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, true, false>), dim3(grid), dim3(NW * 64), G::LDS, st, q, nPixTiles, nCt, nItems);
  } else {
    MAU_LAUNCH((conv3x3_bf16_kernel<BN, MT, NW, EPI, F16, false, false>), dim3(grid), dim3(NW * 64), G::LDS, st, q, nPixTiles, nCt, nItems);
  }
  const int rc = check_launch("conv3x3_bf16_kernel");
  if (rc != MAU_OK || !pool_behind) return rc;
  return mau_maxpool2x2_fwd(p.y, p.ldy, p.pool, p.ldpool, F16 ? MAU_F16 : MAU_BF16, p.N, p.H, p.W, p.Cout, (mau_stream_t)st);
}

// Variant choice.  Taller workgroup tiles move fewer LDS-DMA bytes and issue fewer ds_reads per MFMA (measured
// +5..9 % on full grids), but quarter the number of work items: a layer whose items do not fill the 256 CUs (or that
// wastes tile rows on a small image) is better off with 16-row tiles.  Score = grid-fill x tile-fill x variant bonus.
// (Measured and dropped: <64,4,4> = two 4-wave workgroups per CU drifting apart -- the extra weight-slab DMA traffic cost
//  more than the overlap gained, 207 vs 182 us on the 64->64 level-0 layer; <*,8,4> = one wave per SIMD with 256
//  accumulators in AGPRs, 0.48 instead of 0.75 fragment reads per MFMA -- 10-90 % slower, nothing hides its epilogue.)
struct Variant {
  int th, nw, bn;                  // tile rows, waves per workgroup, output channels per workgroup   ((32, 4, 64) = <64,4,4>; (32, 8, 64) = <64,2,8>; (32, 8, 128) = <128,4,8>)
};
static inline Variant pick_variant(int CoutPad, int N, int H, int W, int Cin) {
  const bool wide = CoutPad % 128 == 0;
  // A 128-multiple layer whose 16-row items would keep fewer than half of the CUs busy (single-tile inference at the deep levels:
  // 512 x 512, B = 1 gives 32 items of <128,2,8> at 32 x 32 pixels) runs on the 64-channel workgroups instead: twice the items of
  // half the work, on twice the CUs (B = 1 conv4_0.conv2: 86 -> 45 us).  The packed weights are laid out per 64-channel block, so
  // both widths read the same packs.
  if (wide) {
    const long tiles16 = (long)N * ceil_div(H, 16) * ceil_div(W, TW);
    // (... and 8-row tiles <64,1,4> while even those items are fewer than half of the CUs: the chain of one wave halves again)
    if (tiles16 * (CoutPad / 64) * 2 <= device_shape().cus) return {8, 4, 64};        // <64,1,4>
    if (tiles16 * (CoutPad / 128) * 2 <= device_shape().cus) return {16, 4, 64};      // <64,2,4>
  }
  const int nCt = CoutPad / (wide ? 128 : 64);
  Variant best = {16, wide ? 8 : 4, wide ? 128 : 64};
  double best_score = -1.0;
  // 64-row tiles exist for the 64-wide variant only (<64,4,8>: the per-MFMA LDS-read and DMA ratios of <128,4,8>)
  for (int th = 16; th <= (wide ? 32 : 64); th *= 2) {
    const int nw = wide || th > 16 ? 8 : 4;
    const int slots = device_shape().cus * (nw == 4 ? 2 : 1);      // <64,2,4> runs two workgroups per CU
    const long tilesY = ceil_div(H, th), tilesX = ceil_div(W, TW);
    const long items = (long)N * tilesY * tilesX * nCt;
    const double grid_fill = (double)items / (double)(((items + slots - 1) / slots) * slots);
    const double tile_fill = (double)H * W / (double)(tilesY * th * tilesX * TW);
    const double score = grid_fill * tile_fill * (th == 64 ? 1.10 : th == 32 ? 1.06 : 1.0);
    if (score > best_score) {
      best_score = score;
      best = {th, nw, wide ? 128 : 64};
    }
  }
  // Level 0 (64 output channels, the 64-row tile wins the score): TWO independent 4-wave workgroups per CU, each on a 32 x 16 pixel
  // tile (<64,4,4>, the same 128-pixel wave strips and 16x16x32 stage-pair loop).  At K = 64..192 an item is 2-6 stage pairs and ends
  // in an epilogue that moves 128 KB per CU through the store path with the matrix pipes idle (all eight waves of the one resident
  // workgroup are in it together); two workgroups drift apart, one's epilogue runs beside the other's multiply loop.  Costs: the weight
  // slab is fetched by both (L2 hits; 74 instead of 55 DMA bytes per pixel) -- which is why the choice depends on K (round 6,
  // profiles/r6/level0_tile_form_by_k.txt, same call, alternating): K = 64: <64,4,4> 3-5 % faster; K = 192: equal; K = 208..400 (the
  // U-Net++'s full-resolution nodes): the one 8-wave workgroup on the 64 x 16 tile <64,4,8> 3-5 % faster forward -- the epilogue is a
  // seventh to a thirteenth of such an item and the DMA bytes are what is left.  Both forms write the same statistics slab geometry
  // (conv_bf16_v2_num_pixel_tiles does not know Cin).  Cin = 0: unknown (the slab-geometry query).
  static const int l0 = getenv("MAU_CONV_L0") ? atoi(getenv("MAU_CONV_L0")) : -1;      // test hook: 0 / 1 force <64,4,8> / <64,4,4> (both forms stay under the exact big-tile tests)
  if (!wide && best.th == 64 && (l0 < 0 ? Cin <= 192 : l0 != 0)) best = {32, 4, 64};
  return best;
}
}  // namespace v2

// ---- translation units ----
// The kernel has 120 instantiations (7 tile variants x 2 loaders + 3 big-tile variants x 2 stage-pair forms, x 3 epilogues x 2 operand types) of
// 4-5 s each: one (epilogue, operand type) slice per translation unit (conv3x3_bf16_e<EPI>_<type>.hip define MAU_CONV_TU_EPI /
// MAU_CONV_TU_F16 and include this file), built in parallel; this file alone carries the variant choice and the dispatcher.
#define MAU_CONV_TU_NAME2(e, f) launch_conv_bf16_tu_e##e##_f##f
#define MAU_CONV_TU_NAME(e, f) MAU_CONV_TU_NAME2(e, f)
#ifdef MAU_CONV_TU_EPI
int MAU_CONV_TU_NAME(MAU_CONV_TU_EPI, MAU_CONV_TU_F16)(const ConvP& p, int th, int nw, int bn, hipStream_t st) {
  constexpr int E = MAU_CONV_TU_EPI;
  constexpr bool F = MAU_CONV_TU_F16 != 0;
  if (bn == 128) return th == 32 ? v2::launch<128, 4, 8, E, F>(p, st) : v2::launch<128, 2, 8, E, F>(p, st);
  if (th == 8) return v2::launch<64, 1, 4, E, F>(p, st);
  if (th == 64) return v2::launch<64, 4, 8, E, F>(p, st);
  if (th == 32 && nw == 4) return v2::launch<64, 4, 4, E, F>(p, st);
  return th == 32 ? v2::launch<64, 2, 8, E, F>(p, st) : v2::launch<64, 2, 4, E, F>(p, st);
}
#else
int launch_conv_bf16_tu_e0_f0(const ConvP&, int, int, int, hipStream_t);
int launch_conv_bf16_tu_e1_f0(const ConvP&, int, int, int, hipStream_t);
int launch_conv_bf16_tu_e2_f0(const ConvP&, int, int, int, hipStream_t);
int launch_conv_bf16_tu_e0_f1(const ConvP&, int, int, int, hipStream_t);
int launch_conv_bf16_tu_e1_f1(const ConvP&, int, int, int, hipStream_t);
int launch_conv_bf16_tu_e2_f1(const ConvP&, int, int, int, hipStream_t);

// rows of the BatchNorm partial-sum slab: one per (pixel tile, wave row of the workgroup)
int conv_bf16_v2_num_pixel_tiles(int N, int H, int W, int Cout) {
  const int CoutPad = round_up(Cout, 64);
  const v2::Variant v = v2::pick_variant(CoutPad, N, H, W, 0);     // (the slab geometry must not depend on Cin: see pick_variant's level-0 rule)
  const int th = v.th;
  const int wm = (v.bn == 64 && v.nw == 8) ? v2::Geo<64, 2, 8>::WM : 4;    // <128,*,8>, <64,2,4>, <64,4,4>: 4 wave rows; <64,2,8>, <64,4,8>: 8
  static_assert(v2::Geo<64, 4, 4>::WM == 4 && v2::Geo<64, 4, 4>::TH == 32, "slab rows");
  static_assert(v2::Geo<64, 2, 8>::WM == 8 && v2::Geo<64, 4, 8>::WM == 8 && v2::Geo<64, 4, 8>::TH == 64, "slab rows");
  static_assert(v2::Geo<128, 2, 8>::WM == 4 && v2::Geo<128, 4, 8>::WM == 4 && v2::Geo<64, 2, 4>::WM == 4, "slab rows");
  static_assert(v2::Geo<64, 1, 4>::WM == 4 && v2::Geo<64, 1, 4>::TH == 8, "slab rows");
  return wm * N * ceil_div(H, th) * ceil_div(W, v2::TW);
}

// which tile variant the 16-bit kernel runs for a layer: tile rows, waves per workgroup, output channels per workgroup
// ... and, given the layer's input channels (a single-source layer on the buffer-addressed loader, plain or inference epilogue),
// the number of K groups per workgroup (2 = the under-filled-layer form, see k_groups_rule)
void conv_bf16_v2_variant(int N, int H, int W, int Cin, int Cout, int* th, int* nw, int* bn, int* kg) {
  const int CoutPad = round_up(Cout, 64);
  const v2::Variant v = v2::pick_variant(CoutPad, N, H, W, Cin);
  *th = v.th;
  *nw = v.nw;
  *bn = v.bn;
  const int mt = v.th / (2 * (v.nw / (v.bn / 64)));
  const DeviceShape ds = device_shape();
  const int nPixTiles = N * ceil_div(H, v.th) * ceil_div(W, v2::TW);
  const int nItems = round_up(nPixTiles, nPixTiles < ds.xcds ? 1 : ds.xcds) * (CoutPad / v.bn);
  const bool m16 = mt == 4 && Cin > 0 && ceil_div(Cin, v2::KC) % 2 == 0;
  *kg = (Cin > 0 && Cin % v2::KC == 0 && !m16 && v2::has_k_groups(v.bn, mt, v.nw) && v2::k_groups_rule(ceil_div(Cin, v2::KC), nItems)) ? 2 : 1;
}

int launch_conv_bf16_v2(const ConvP& p, bool f16, hipStream_t st) {
  if (p.post_scale != nullptr && p.slab != nullptr) {
    set_error("conv3x3_fwd: post_scale/post_shift and the statistics slab are mutually exclusive");
    return MAU_ERR_ARG;
  }
  static_assert(v2::EPI_PLAIN == 0 && v2::EPI_STATS == 1 && v2::EPI_POST == 2, "translation-unit names");
  const v2::Variant v = v2::pick_variant(p.CoutPad, p.N, p.H, p.W, p.C0 + p.C1 + p.E);
  const int th = v.th, nw = v.nw, bn = v.bn;
  if (p.post_scale != nullptr) return f16 ? launch_conv_bf16_tu_e2_f1(p, th, nw, bn, st) : launch_conv_bf16_tu_e2_f0(p, th, nw, bn, st);   // (a post-affine launch carries no slab)
  if (p.slab != nullptr) return f16 ? launch_conv_bf16_tu_e1_f1(p, th, nw, bn, st) : launch_conv_bf16_tu_e1_f0(p, th, nw, bn, st);
  return f16 ? launch_conv_bf16_tu_e0_f1(p, th, nw, bn, st) : launch_conv_bf16_tu_e0_f0(p, th, nw, bn, st);
}
#endif

}  // namespace mau
#endif
