// conv3x3_first.hip -- the FIRST convolution of the network (reference src/model.py:222 / :67, conv0_0.conv1: spatial_channels -> 64)
// for inputs of at most 8 channels, 16-bit arithmetic: reads the reference's input tensor AS IT ARRIVES -- (N, C, H, W) fp32, the layout
// of collate_fn, src/dataset.py:99-106 -- and writes the NHWC 16-bit conv output + BatchNorm partial sums.
//
// Why a kernel of its own.  K = 9 taps x 6 channels = 54: the layer is 0.4 % of the network's FLOPs and moves 318 MB (50 MB in, 268 MB
// out at B = 32): pure HBM work.  Through the generic path it cost a layout kernel (NCHW fp32 -> NHWC-8 bf16: 83 MB of traffic) plus a
// 16-channel stage of the implicit-GEMM kernel whose loader, stage barriers and LDS-staged epilogue are built for K in the hundreds
// (76 us against a 50 us floor).  Here:
//   input    : a (16 + 2) x (64 + 2) pixel halo tile is read plane by plane (coalesced 4-byte lanes along W), converted once, and kept
//              in LDS as [pixel][8 channels] (16 bytes per pixel, zero padding outside the image and above C channels)
//   multiply : v_mfma_f32_16x16x32 with the WEIGHTS as the A operand and 16 consecutive pixels of a row as B: one K = 32 step is
//              4 taps x 8 channels (a lane group's 16-byte fragment IS the pixel shifted by its tap), 3 steps cover the 9 taps
//              (the 3 x 4 weight fragments live in registers for the whole persistent workgroup: 48 VGPRs)
//   epilogue : D = [cout][pixel], so a lane holds 16 output channels of ONE pixel: with the cout rows of the A tiles permuted
//              (tile t, row 4q + r <-> channel 32 (t >> 1) + 8 q + 4 (t & 1) + r) they are two runs of 8 consecutive channels = two
//              16-byte stores straight from registers -- no LDS staging; a store instruction covers 16 pixels x 64 contiguous bytes
//   BN sums  : per-lane fp32 running sums (its 16 channels) over every pixel the workgroup processes, reduced across lanes and
//              waves ONCE at the end: one slab row per workgroup (fixed order: bitwise reproducible)
//   by-product (training): the converted input as an NHWC-8 tensor for the weight-gradient kernel (x8), written from the halo pass
#include "igemm_bf16_util.h"

namespace mau {
namespace first {
using namespace igemm;

constexpr int TH = 16, TW = 64, HR = TH + 2, HC = TW + 2, HPIX = HR * HC;     // halo: 18 x 66 pixels x 16 B = 19008 B of LDS
constexpr int NT = 256, NWAVE = NT / 64;
constexpr int LOADS = (HPIX + NT - 1) / NT;                                    // halo pixels per thread (5)
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct P {
  const float* x;
  int Cin;
  const float* w;
  const float* bias;
  const float* post_scale;
  const float* post_shift;
  void* y;
  int ldy, Cout, CoutPad;
  float* slab;
  void* x8;
  int N, H, W, tilesX, tilesY, nTiles;
};

template <bool F16>
__device__ __forceinline__ f32x4v mfma(const bf16x8& a, const bf16x8& b, f32x4v c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// eight fp32 -> one 16-byte vector of the activation type (round to nearest even; the same conversion mau_nchw_to_nhwc applies)
template <bool F16>
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (F16) r[i] = pack_lp2<true>(f32x2{opaque(v[2 * i]), opaque(v[2 * i + 1])});
    else r[i] = pack_lp2<false>(f32x2{v[2 * i], v[2 * i + 1]});
  }
  return r;
}

// the channel (inside a 64-channel output tile) of accumulator register r of A-tile t for lane group q: two runs of 8 per lane
__device__ __forceinline__ int chan_of(int t, int q, int r) { return 32 * (t >> 1) + 8 * q + 4 * (t & 1) + r; }

#ifndef MAU_FIRST_UNROLL
#define MAU_FIRST_UNROLL 2           // pixel groups per trip of the inner loop: one group's LDS round trips and MFMA chain beside the other's VALU
#endif
#ifndef MAU_FIRST_WAVES
#define MAU_FIRST_WAVES 2            // waves per SIMD = 4-wave workgroups per CU the register allocation is held to
#endif
template <bool F16, int EPI>
__global__ __launch_bounds__(NT, MAU_FIRST_WAVES) void first_fwd_kernel(P p) {
  __shared__ __attribute__((aligned(16))) unsigned char halo[HPIX * 16];
  __shared__ float red[NWAVE][2 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, n = lane & 15;
  const int co0 = blockIdx.y * 64;

  // ---- the weights as A fragments (tile t, step s): row m = lane % 16 <-> output channel chan_of(t, m / 4, m % 4), K block
  // lane / 16 <-> tap 4 s + lane / 16 (taps >= 9: zeros), element j <-> input channel j (channels >= Cin: zeros).  The tile's
  // <= 64 x Cin x 9 fp32 weights (<= 18 KB, contiguous in OIHW) pass through the halo buffer: one coalesced sweep, then 96 LDS
  // reads per lane -- read straight from global memory the 96 dependent scalar loads of every workgroup were a third of the kernel ----
  bf16x8 A[3][4];
  {
    float* wl = reinterpret_cast<float*>(halo);
    static_assert(64 * 8 * 9 * sizeof(float) <= sizeof(halo), "weight tile fits the halo buffer");
    const int rowlen = p.Cin * 9, nrows = min(64, p.Cout - co0);
    const float* wsrc = p.w + (size_t)co0 * rowlen;
    for (int i = tid; i < nrows * rowlen; i += NT) wl[i] = wsrc[i];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int col = chan_of(t, n >> 2, n & 3), tau = 4 * s + q;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = tau < 9 && j < p.Cin && col < nrows;
          const float wv = wl[ok ? (col * p.Cin + j) * 9 + tau : 0];
          v[j] = ok ? wv : 0.f;
        }
        A[s][t] = __builtin_bit_cast(bf16x8, pack8<F16>(v));
      }
  }                                                      // (the tile loop's first barrier separates these reads from the halo writes)
  // per-lane epilogue coefficients of the lane's 16 channels
  // (register pairs (r = 0, 1), (r = 2, 3) of an accumulator tile: bias, statistics and the inference affine map are packed fp32 math)
  f32x2 bb2[4][2], psc2[4][2], psh2[4][2], s1[4][2], s2[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int co = co0 + chan_of(t, q, 2 * h + e);
        const float bv = p.bias != nullptr ? coef(p.bias, co, p.Cout) : 0.f;
        bb2[t][h][e] = bv;
        if (EPI == EPI_POST) {                           // relu(sc * (acc + b) + sh) = relu(fma(acc, sc, fma(sc, b, sh))): one coefficient pair
          psc2[t][h][e] = coef(p.post_scale, co, p.Cout);
          psh2[t][h][e] = fmaf(psc2[t][h][e], bv, coef(p.post_shift, co, p.Cout));
        }
      }
      s1[t][h] = s2[t][h] = f32x2{0.f, 0.f};
    }
  // LDS address of the lane's B fragment of step s inside a group: its pixel (n) shifted by its tap (taps 9..11 of step 2 carry zero
  // weights: any finite pixel will do -- tap 8)
  unsigned lane_lds[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tau = min(4 * s + q, 8), dy = tau / 3, dx = tau - 3 * dy;
    lane_lds[s] = (unsigned)((dy * HC + n + dx) * 16);
  }
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // Stores: a lane holds 16 channels of ONE pixel (two runs of 8), so a store of a lane's own runs would cover 16 pixels x 64 bytes --
  // half lines.  Lanes n and n ^ 8 of a group swap their upper runs (one DPP row rotate per dword), then store 1 covers pixels 0..7 of the
  // group and store 2 pixels 8..15, each with all eight 16-byte chunks of the pixel: whole 128-byte lines, 1 KB contiguous per instruction.
  //   lanes n < 8 : store 1 = own lower run   @ (pixel n,     chunk q)      store 2 = partner's upper run @ (pixel n + 8, chunk 4 + q)
  //   lanes n >= 8: store 1 = partner's upper @ (pixel n - 8, chunk 4 + q)  store 2 = own lower run       @ (pixel n,     chunk q)
  const bool lowhalf = n < 8;
  const int ch1 = lowhalf ? q : 4 + q, ch2 = lowhalf ? 4 + q : q;
  const bool st1 = co0 + 8 * ch1 < p.ldy, st2 = co0 + 8 * ch2 < p.ldy;            // the chunk exists in y (narrow layers: ldy < 64)
  const unsigned off1 = (unsigned)(((n & 7) * p.ldy + co0 + 8 * ch1) * 2), off2 = (unsigned)(((8 + (n & 7)) * p.ldy + co0 + 8 * ch2) * 2);
  unsigned char* const yb = reinterpret_cast<unsigned char*>(p.y);
  const size_t plane = (size_t)p.H * p.W;

  // ---- halo pass, split in two so that the NEXT tile's input is in flight while the current tile is multiplied (two workgroups per CU
  // do not hide a load phase that stalls all four waves of one): fetch() issues the tile's <= 5 x Cin plane loads per lane into
  // registers (out-of-image lanes read pixel 0 and discard it: no exec-masked branches), commit() converts them, writes the
  // [pixel][8 channels] LDS image and, for the weight gradient, the NHWC-8 copy of the tile's interior ----
  float hv[LOADS][8];
  auto coords = [&](int item, int& img, int& ty0, int& tx0) {
    const int txi = item % p.tilesX;
    item /= p.tilesX;
    img = item / p.tilesY;
    ty0 = (item - img * p.tilesY) * TH;
    tx0 = txi * TW;
  };
  auto fetch = [&](int item) {
    int img, ty0, tx0;
    coords(item, img, ty0, tx0);
    const float* src[LOADS];
    bool in[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
      const int pidx = tid + k * NT;
      const int hy = pidx / HC, hx = pidx - hy * HC;
      const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
      in[k] = pidx < HPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      src[k] = p.x + (size_t)img * p.Cin * plane + (in[k] ? (size_t)gy * p.W + gx : 0);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < p.Cin) {                                   // (wave-uniform: a scalar branch, not an exec mask)
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
          const float v = src[k][(size_t)c * plane];
          hv[k][c] = in[k] ? v : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) hv[k][c] = 0.f;
      }
    }
  };
  auto commit = [&](int item) {
    int img, ty0, tx0;
    coords(item, img, ty0, tx0);
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
      const int pidx = tid + k * NT;
      if (pidx < HPIX) {
        const u32x4 pk = pack8<F16>(hv[k]);
        *reinterpret_cast<u32x4*>(halo + pidx * 16) = pk;
        if (p.x8 != nullptr && blockIdx.y == 0) {
          const int hy = pidx / HC, hx = pidx - hy * HC;
          const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
          if (hy >= 1 && hy <= TH && hx >= 1 && hx <= TW && gy < p.H && gx < p.W)
            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(p.x8) + (((size_t)img * p.H + gy) * p.W + gx) * 16) = pk;
        }
      }
    }
  };
  static_assert(LOADS == 5, "halo pixels per thread");
#ifdef MAU_FIRST_PREFETCH
  if ((int)blockIdx.x < p.nTiles) fetch(blockIdx.x);
#endif
  for (int item = blockIdx.x; item < p.nTiles; item += gridDim.x) {
    int img, ty0, tx0;
    coords(item, img, ty0, tx0);
    __syncthreads();                                     // everyone is done multiplying from the previous tile
#ifndef MAU_FIRST_NOLOAD             // timing-only ablation: stale LDS
#ifndef MAU_FIRST_PREFETCH
    fetch(item);
#endif
    commit(item);
#endif
    __syncthreads();
#if defined(MAU_FIRST_PREFETCH) && !defined(MAU_FIRST_NOLOAD)
    // (measured: prefetching the next tile's planes into registers across the multiply phase -- 40 more live registers -- changes nothing:
    //  the multiply phase, not the load phase, is what the kernel's time is made of; off by default)
    if (item + (int)gridDim.x < p.nTiles) fetch(item + gridDim.x);
#endif
    // ---- multiply: a wave owns 4 tile rows = 16 groups of 16 pixels.  Everything a group needs is (wave-uniform scalar) + (per-lane
    // constant): LDS address = lane_lds[s] + row / column offset; store address = uniform row pointer + off1 / off2.  Interior tiles
    // (FULL, the common case) carry no validity masks. ----
    const bool full = ty0 + TH <= p.H && tx0 + TW <= p.W;
    auto rows = [&](auto fullc) {
      constexpr bool FULL = decltype(fullc)::value;
#pragma unroll 1
      for (int rr = 0; rr < 4; ++rr) {
        const int ry = wv * 4 + rr, gy = ty0 + ry;
        unsigned char* const rowp = yb + (((size_t)img * p.H + gy) * p.W + tx0) * p.ldy * 2;       // (uniform)
        const bool yok = FULL || gy < p.H;
#pragma unroll MAU_FIRST_UNROLL
        for (int gc = 0; gc < 4; ++gc) {
          const int cx = gc * 16;
          const unsigned la = (unsigned)((ry * HC + cx) * 16);
          f32x4v acc[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const bf16x8 B = *reinterpret_cast<const bf16x8*>(halo + lane_lds[s] + la);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma<F16>(A[s][t], B, acc[t]);
          }
          const bool valid = FULL || (yok && tx0 + cx + n < p.W);
          f32x2 v[4][2];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 o = f32x2{acc[t][2 * h], acc[t][2 * h + 1]};
              if (EPI != EPI_POST) o += bb2[t][h];
              if (EPI == EPI_STATS) {
                const f32x2 m = FULL ? o : (valid ? o : f32x2{0.f, 0.f});
                s1[t][h] += m;
                s2[t][h] = __builtin_elementwise_fma(m, m, s2[t][h]);
              }
              if (EPI == EPI_POST) o = __builtin_elementwise_max(__builtin_elementwise_fma(o, psc2[t][h], psh2[t][h]), f32x2{0.f, 0.f});
              v[t][h] = o;
            }
          const u32x4 plo = {pack_lp2<F16>(v[0][0]), pack_lp2<F16>(v[0][1]), pack_lp2<F16>(v[1][0]), pack_lp2<F16>(v[1][1])};
          const u32x4 phi = {pack_lp2<F16>(v[2][0]), pack_lp2<F16>(v[2][1]), pack_lp2<F16>(v[3][0]), pack_lp2<F16>(v[3][1])};
          u32x4 d1, d2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)phi[i], 0x128, 0xf, 0xf, false);      // row_ror:8 -- lane n ^ 8's upper run
            d1[i] = lowhalf ? plo[i] : other;
            d2[i] = lowhalf ? other : plo[i];
          }
          unsigned char* const base = rowp + (size_t)cx * p.ldy * 2;
          const int gx1 = tx0 + cx + (n & 7);
          const bool ok1 = st1 && (FULL || (yok && gx1 < p.W)), ok2 = st2 && (FULL || (yok && gx1 + 8 < p.W));
#if defined(MAU_FIRST_NOSTORE)        // timing-only ablation (scripts/build_variants.sh): what the stores cost
          if (d1[0] + d2[1] == 0x12345678u) *reinterpret_cast<u32x4*>(base + off1) = d1;
#else
          if (ok1) __builtin_nontemporal_store(d1, reinterpret_cast<u32x4*>(base + off1));
          if (ok2) __builtin_nontemporal_store(d2, reinterpret_cast<u32x4*>(base + off2));
#endif
        }
      }
    };
    if (full) rows(std::true_type{});
    else rows(std::false_type{});
  }
  if (EPI == EPI_STATS) {
    // lanes n = 0..15 of a group hold partial sums of the SAME 16 channels: butterfly over n, then the four waves in fixed order
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a1 = s1[t][r >> 1][r & 1], a2 = s2[t][r >> 1][r & 1];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          a1 += __shfl_xor(a1, d);
          a2 += __shfl_xor(a2, d);
        }
        if (n == 0) {
          red[wave][chan_of(t, q, r)] = a1;
          red[wave][64 + chan_of(t, q, r)] = a2;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      const float tot = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
      if (co0 + c < p.CoutPad) p.slab[(size_t)blockIdx.x * 2 * p.CoutPad + which * p.CoutPad + co0 + c] = tot;
    }
  }
}

static int grid_x(int nTiles, int per_cu = MAU_FIRST_WAVES) {
  const int cap = device_shape().cus * per_cu;           // persistent: as many 4-wave workgroups as are resident at once
  return nTiles < cap ? nTiles : cap;
}

template <bool F16>
static int launch(const P& p, hipStream_t st) {
  dim3 grid(grid_x(p.nTiles), p.CoutPad / 64);
  if (p.post_scale != nullptr) MAU_LAUNCH((first_fwd_kernel<F16, EPI_POST>), grid, dim3(NT), 0, st, p);
  else if (p.slab != nullptr) MAU_LAUNCH((first_fwd_kernel<F16, EPI_STATS>), grid, dim3(NT), 0, st, p);
  else MAU_LAUNCH((first_fwd_kernel<F16, EPI_PLAIN>), grid, dim3(NT), 0, st, p);
  return check_launch("first_fwd_kernel");
}
}  // namespace first
}  // namespace mau

extern "C" {

int mau_conv3x3_first_max_channels(void) { return 8; }

int mau_conv3x3_first_rows(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return mau::first::grid_x(N * mau::ceil_div(H, mau::first::TH) * mau::ceil_div(W, mau::first::TW));
}

int mau_conv3x3_first_fwd(const float* x, int Cin, const float* w, const float* bias, const float* post_scale, const float* post_shift, void* y,
                          int ldy, int Cout, float* slab, void* x8, int dtype, int N, int H, int W, mau_stream_t stream) {
  using namespace mau;
  MAU_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_first_fwd: bad arguments");
  MAU_REQUIRE(Cin >= 1 && Cin <= 8, "conv3x3_first_fwd: %d input channels (this kernel serves <= 8; use mau_conv3x3_fwd)", Cin);
  MAU_REQUIRE(dtype == MAU_BF16 || dtype == MAU_F16, "conv3x3_first_fwd: 16-bit activation types only (fp32 = parity mode: mau_conv3x3_fwd)");
  MAU_REQUIRE(ldy % 8 == 0 && ldy >= Cout, "conv3x3_first_fwd: ldy must be a multiple of 8 and >= Cout");
  MAU_REQUIRE((post_scale == nullptr) == (post_shift == nullptr), "conv3x3_first_fwd: post_scale and post_shift come together");
  MAU_REQUIRE(!(post_scale != nullptr && slab != nullptr), "conv3x3_first_fwd: post_scale/post_shift and the statistics slab are mutually exclusive");
  MAU_REQUIRE((long long)N * H * W * ldy * 2 < (1ll << 40), "conv3x3_first_fwd: tensor too large");
  first::P p;
  p.x = x;
  p.Cin = Cin;
  p.w = w;
  p.bias = bias;
  p.post_scale = post_scale;
  p.post_shift = post_shift;
  p.y = y;
  p.ldy = ldy;
  p.Cout = Cout;
  p.CoutPad = round_up(Cout, 64);
  p.slab = slab;
  p.x8 = x8;
  p.N = N;
  p.H = H;
  p.W = W;
  p.tilesX = ceil_div(W, first::TW);
  p.tilesY = ceil_div(H, first::TH);
  p.nTiles = N * p.tilesX * p.tilesY;
  return dtype == MAU_F16 ? first::launch<true>(p, (hipStream_t)stream) : first::launch<false>(p, (hipStream_t)stream);
}

}  // extern "C"

// ======================================================================================================================
// The first layer's WEIGHT GRADIENT:  dW[co][ci][tap] = sum over pixels  dZ[pix][co] * X[pix + tap][ci],  Cin <= 8.
// (autograd's conv weight gradient of conv0_0.conv1 under loss.backward(), reference src/train.py:252 / src/model.py:12,222)
//
// 14.5 GFLOP against 302 MB of reads (dZ: 268 MB, X as the forward's NHWC-8 by-product: 34 MB): pure HBM work, 50 us at 6 TB/s.
// The generic weight-gradient kernel (64 input channels per workgroup, 30 halo DMAs of mostly padding per stage, three of four
// waves multiplying zeros) needs 117 us.  Here the GEMM is  D[co (64)][n = (tap, ci): six column tiles of 16]  with K = pixels:
//   * a WAVE owns a stream of (row, 32-pixel segment) tiles and two private LDS buffers: no workgroup barrier in the loop, and no
//     global store before the end -- so "my DMAs have landed" is a counted vmcnt, never a wait behind stores (the first-layer
//     forward kernel's lesson, DESIGN.md "Round 4")
//   * per tile 7 wave-DMAs (LDS-DMA; out-of-image lanes read a zero page = the padding): the dZ row segment
//     [32 px][64 co] (4 KB) and three X halo rows [34 px][8 ci] (16 bytes per pixel)
//   * one K = 32 step of v_mfma_f32_16x16x32 per tile: A = dZ^T (4 co tiles), B = the im2col of X, both through ds_read_b64_tr_b16.
//     A B tile's 16 columns are the 32 CONTIGUOUS bytes at (halo row dy, pixel + dx): the 8 channels of tap (dy, dx) and of its right
//     neighbour (dy, dx + 1).  (A transposed read wants the four lanes of a block row on one contiguous 32-byte run: a tile made of
//     taps (0,2) and (1,0) -- two rows of the halo -- returned wrong columns.)  Six tiles: (dy, dx = 0 | 1) and (dy, dx = 2 | junk)
//     per halo row; the junk halves are never summed
//   * 96 accumulator registers; the four waves of a workgroup add up through LDS (fixed order), one compact slab
//     [workgroup][64][80] fp32, a second kernel adds the slabs in workgroup order and writes OIHW: bitwise reproducible
namespace mau {
namespace firstw {
using namespace igemm;
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TWW = 32;                       // pixels per tile (one K step)
constexpr int DZ_BYTES = TWW * 128;           // [32 px][64 co] 16-bit
constexpr int XROW = 1024;                    // one halo row: 34 pixels x 16 B, padded to a whole wave-DMA
constexpr int BUF = DZ_BYTES + 3 * XROW;      // 7168
constexpr int NT = 256, NWAVE = 4, NDMA = 7;
constexpr int NBT = 6, NCOL = 16 * NBT;       // column tiles: per halo row dy one for taps (dy,0),(dy,1) and one for (dy,2),(junk)

struct WP {
  const void* x8;
  const void* dz;
  int lddz, Cout;
  float* ws;           // [gridDim.x][gridDim.y * 64][NCOL]
  int N, H, W, tilesX, nTiles;
};

template <bool F16>
__device__ __forceinline__ void mfma(const u32x4& a, const u32x4& b, f32x4v& c) {
  if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// 4 rows x 16 columns of 16-bit elements, delivered column-major: lane 4q + p of a 16-lane group supplies the address of row q,
// columns 4p .. 4p+3; lane i receives column i of the four rows.  Two of them (rows 0-3 | 4-7 of a lane group's 8 pixels) = one fragment.
template <int OFF_HI>
__device__ __forceinline__ u32x4 tr_frag(unsigned addr_lo, unsigned addr_hi) {
  u32x2 lo, hi;
  // (early-clobber outputs: without '&' the compiler may give the first read's destination the address register the second read still
  //  needs -- `ds_read_b64_tr_b16 v[114:115], v114` followed by `ds_read_b64_tr_b16 v[116:117], v114 offset:512` -- which works only
  //  while the first read's data has not come back when the second issues: sporadically wrong with two waves per SIMD)
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(addr_lo), "v"(addr_hi), "n"(OFF_HI));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

template <bool F16>
__global__ __launch_bounds__(NT, 2) void first_wgrad_kernel(WP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // NWAVE x 2 x BUF, later the cross-wave sum
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co0 = blockIdx.y * 64;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) + wave * (2 * BUF);
  // Out-of-image lanes read 16 zero bytes of global memory (g_zero_page) through a per-lane 64-bit address.  (The big convolution kernels
  // use the buffer-addressed form -- offset -1, zeros from the hardware range check -- for lanes at the image border; here WHOLE
  // wave-DMAs fall outside the image (a halo row above / below it, the tail of a ragged row segment), a case those kernels never
  // produce and this one does not want to depend on; seven address computations per 32 pixels cost nothing.)
  const unsigned long long dz_a = (unsigned long long)p.dz, x_a = (unsigned long long)p.x8, zero_a = (unsigned long long)g_zero_page;

  // per-lane constants of the DMAs: dZ DMA d (0..3) covers pixels 8d .. 8d+7, lane = (pixel % 8) * 8 + 16-byte chunk; X halo DMA: lane = halo column
  const int dz_px = lane >> 3, dz_ch = lane & 7;
  const bool dz_ch_ok = co0 + 8 * dz_ch < p.lddz;                          // (narrow layers: the row holds fewer than 64 channels)
  auto issue = [&](int tile, int buf) {
    const int txi = tile % p.tilesX, rowi = tile / p.tilesX;               // rowi = n * H + y
    const int x0 = txi * TWW, y = rowi % p.H;
    unsigned char* dst = smem + wave * (2 * BUF) + buf * BUF;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int gx = x0 + 8 * d + dz_px;
      unsigned long long src = dz_a + 2ull * ((unsigned long long)((long long)rowi * p.W + gx) * (unsigned)p.lddz + (unsigned)(co0 + 8 * dz_ch));
      asm volatile("" : "+v"(src));                      // (both candidates materialised: no exec-masked branch around the 64-bit multiply)
      src = (gx < p.W && dz_ch_ok) ? src : zero_a;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(dst + d * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int gy = y + dy - 1, gx = x0 + lane - 1;
      const bool in = lane < TWW + 2 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      unsigned long long src = x_a + 16ull * (unsigned long long)((long long)(rowi + dy - 1) * p.W + gx);
      asm volatile("" : "+v"(src));
      src = in ? src : zero_a;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(dst + DZ_BYTES + dy * XROW), 16, 0, 0);
    }
  };
  // per-lane LDS read addresses (relative to a buffer).  Lane l: K block g = l / 16 (pixels 8g .. 8g+7), i = l % 16 -> row q = i / 4 of
  // the 4-pixel block, column quad pq = i % 4.
  const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
  //   A (co tile t): pixel 8g + q (+4), channels 16t + 4pq .. +3
  const unsigned a_lane = (unsigned)((8 * g + q) * 128 + 8 * pq);
  //   B (column tile j): halo row dy = j / 2, first tap dx = 2 (j % 2): the 32 bytes at halo pixel 8g + q (+4) + dx = that tap's 8 channels
  //   and the next pixel's (tap dx + 1; for dx = 2 a pixel that is no tap: columns nobody sums); the lane takes bytes 8 pq .. 8 pq + 7
  unsigned b_lane[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) b_lane[j] = (unsigned)(DZ_BYTES + (j >> 1) * XROW + (8 * g + q + 2 * (j & 1)) * 16 + 8 * pq);
  f32x4v acc[4][NBT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < NBT; ++j) acc[t][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const int nw = gridDim.x * NWAVE, w0 = blockIdx.x * NWAVE + wave;
  int buf = 0;
  if (w0 < p.nTiles) issue(w0, 0);
  for (int tile = w0; tile < p.nTiles; tile += nw) {
    const bool more = tile + nw < p.nTiles;
    if (more) {
      issue(tile + nw, buf ^ 1);
      wait_vmcnt<NDMA>();                                // the NDMA just issued may be in flight; this tile's have landed
    } else {
      wait_vmcnt<0>();
    }
    const unsigned base = lds0 + buf * BUF;
    u32x4 A[4], B[NBT];
#pragma unroll
    for (int t = 0; t < 4; ++t) A[t] = tr_frag<4 * 128>(base + a_lane + 32 * t, base + a_lane + 32 * t);
#pragma unroll
    for (int j = 0; j < NBT; ++j) B[j] = tr_frag<4 * 16>(base + b_lane[j], base + b_lane[j]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < NBT; ++j) mfma<F16>(A[t], B[j], acc[t][j]);
    // (the fragments are in registers: the buffer may be refilled by the DMAs issued at the top of the next trip)
    buf ^= 1;
  }
  wait_vmcnt<0>();
  // ---- the four waves' sums, in wave order, through LDS; D layout: column n = lane % 16 of tile j, rows 4 (lane / 16) + r of tile t ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);            // [64 co][NCOL] = 24 KB <= NWAVE * 2 * BUF
  const int dn = lane & 15, dq = lane >> 4;
  for (int wv = 0; wv < NWAVE; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NBT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* e = red + (16 * t + 4 * dq + r) * NCOL + 16 * j + dn;
            *e = wv == 0 ? acc[t][j][r] : *e + acc[t][j][r];
          }
    }
    __syncthreads();
  }
  float* out = p.ws + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64 * NCOL;
  for (int i = tid; i < 64 * NCOL; i += NT) out[i] = red[i];
}

// ws [nwg][coTiles * 64][NCOL] -> dw (Cout, Cin, 3, 3).  A workgroup = 32 output elements x 8 slices: slice s adds the workgroups'
// slabs s, s + 8, s + 16, ... (8 loads in flight per thread), the slices meet in LDS and are added in slice order: a fixed tree.
constexpr int SUM_SL = 8, SUM_EL = 32;
__global__ __launch_bounds__(SUM_SL * SUM_EL) void first_wgrad_sum_kernel(const float* __restrict__ ws, int nwg, int coTiles, float* __restrict__ dw, int Cout, int Cin) {
  __shared__ float part[SUM_SL][SUM_EL];
  const int el = threadIdx.x % SUM_EL, sl = threadIdx.x / SUM_EL;
  const int i = blockIdx.x * SUM_EL + el, total = Cout * Cin * 9;
  const int ic = i < total ? i : 0;
  const int tap = ic % 9, ci = (ic / 9) % Cin, co = ic / (9 * Cin);
  const int dy = tap / 3, dx = tap - 3 * dy;
  const int col = 16 * (2 * dy + (dx == 2)) + 8 * (dx == 1) + ci;              // column tile (dy, dx = 0|1) or (dy, dx = 2), half, channel
  const float* src = ws + (size_t)co * NCOL + col;
  const size_t stride = (size_t)coTiles * 64 * NCOL;
  float s = 0.f;
  int wgi = sl;
  for (; wgi + 7 * SUM_SL < nwg; wgi += 8 * SUM_SL) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(wgi + u * SUM_SL) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; wgi < nwg; wgi += SUM_SL) s += src[(size_t)wgi * stride];
  part[sl][el] = s;
  __syncthreads();
  if (sl == 0 && i < total) {
    float o = part[0][el];
#pragma unroll
    for (int k = 1; k < SUM_SL; ++k) o += part[k][el];
    dw[i] = o;
  }
}

static int grid_x(int nTiles) {
  const int cap = device_shape().cus * 2;                // two 4-wave workgroups (2 x 56 KB of LDS) per CU
  const int need = ceil_div(nTiles, NWAVE);
  return need < cap ? need : cap;
}
}  // namespace firstw
}  // namespace mau

extern "C" {

size_t mau_conv3x3_first_wgrad_ws_elems(int N, int H, int W, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  return (size_t)mau::firstw::grid_x(N * H * mau::ceil_div(W, mau::firstw::TWW)) * mau::round_up(Cout, 64) * mau::firstw::NCOL;
}

int mau_conv3x3_first_wgrad(const void* x8, const void* dz, int lddz, float* dw, float* ws, int Cin, int Cout, int dtype, int N, int H, int W,
                            mau_stream_t stream) {
  using namespace mau;
  MAU_REQUIRE(x8 && dz && dw && ws && N > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_first_wgrad: bad arguments");
  MAU_REQUIRE(Cin >= 1 && Cin <= 8, "conv3x3_first_wgrad: %d input channels (this kernel serves <= 8; use mau_conv3x3_wgrad2)", Cin);
  MAU_REQUIRE(dtype == MAU_BF16 || dtype == MAU_F16, "conv3x3_first_wgrad: 16-bit activation types only");
  MAU_REQUIRE(lddz % 8 == 0 && lddz >= Cout && ((uintptr_t)x8 % 16) == 0 && ((uintptr_t)dz % 16) == 0, "conv3x3_first_wgrad: bad lddz / alignment");
  firstw::WP p;
  p.x8 = x8;
  p.dz = dz;
  p.lddz = lddz;
  p.Cout = Cout;
  p.ws = ws;
  p.N = N;
  p.H = H;
  p.W = W;
  p.tilesX = ceil_div(W, firstw::TWW);
  p.nTiles = N * H * p.tilesX;
  const int coTiles = round_up(Cout, 64) / 64;
  const dim3 grid(firstw::grid_x(p.nTiles), coTiles);
  constexpr size_t lds = (size_t)firstw::NWAVE * 2 * firstw::BUF;
  static_assert(lds >= 64 * firstw::NCOL * sizeof(float), "the cross-wave sum fits the DMA buffers");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MAU_F16) {
    MAU_LDS_ATTR(lds, &firstw::first_wgrad_kernel<true>);
    MAU_LAUNCH(firstw::first_wgrad_kernel<true>, grid, dim3(firstw::NT), lds, st, p);
  } else {
    MAU_LDS_ATTR(lds, &firstw::first_wgrad_kernel<false>);
    MAU_LAUNCH(firstw::first_wgrad_kernel<false>, grid, dim3(firstw::NT), lds, st, p);
  }
  const int rc = check_launch("first_wgrad_kernel");
  if (rc != MAU_OK) return rc;
  MAU_LAUNCH(firstw::first_wgrad_sum_kernel, dim3(ceil_div(Cout * Cin * 9, firstw::SUM_EL)), dim3(firstw::SUM_SL * firstw::SUM_EL), 0, st, ws, (int)grid.x,
             coTiles, dw, Cout, Cin);
  return check_launch("first_wgrad_sum_kernel");
}

}  // extern "C"
