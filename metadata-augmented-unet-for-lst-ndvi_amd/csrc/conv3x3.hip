// conv3x3.hip -- fp32 PARITY-MODE kernels of the 3x3 / stride 1 / pad 1 convolution on NHWC-ld activations,
// weight packing / gradient unpacking, and the C entry points that dispatch on dtype
// (the bf16 throughput kernels live in conv3x3_bf16.hip and conv3x3_wgrad_bf16.hip).
//
// The parity kernels are im2col-free implicit GEMMs on v_mfma_f32_32x32x2_f32, whose result is
// bit-for-bit a k-ordered fp32 FMA chain (no reduced-precision step anywhere):
//   forward / data-gradient :  M = pixels (8x16 spatial tile per workgroup), N = 64 output channels,
//                              K = 9 taps x 16 input channels per chunk; the (8+2)x(16+2) halo tile is
//                              staged ONCE per chunk in LDS and re-read for the 9 taps at shifted rows.
//   weight-gradient         :  M = 64 output channels, N = 64 input channels (x 9 taps kept in
//                              accumulators), K = pixels; every workgroup owns a strided set of pixel tiles
//                              (split-K) and writes its partial with plain stores into its own slab
//                              [split][9][Cout64][Cin64]; unpack_wgrad_kernel adds the splits in a fixed order
//                              (bitwise reproducible, no float atomics -- as the 16-bit path).
// The templates are written on the element type T but only instantiated for float.
//
// Replaces nn.Conv2d(cin, cout, 3, padding=1) of VGGBlock (reference src/model.py:12,14), the
// first half of train-mode BatchNorm2d (:13,15: per-channel sum / sum of squares, fused into the
// epilogue) and fuse_embeddings (:248-259: the broadcast embedding is a second, spatially
// constant K-source of the loader and is never materialised).
#include "conv_common.h"

namespace mau {

constexpr int TH = 8, TW = 16;               // spatial tile (pixels)
constexpr int HW_ = TW + 2, HH_ = TH + 2;    // halo tile
constexpr int HALO = HW_ * HH_;              // 180 pixels
constexpr int BN = 64;                       // output-channel tile
constexpr int NT = 256;                      // threads per workgroup (4 waves)

template <typename T>
struct Cfg;
template <>
struct Cfg<float> {
  static constexpr int KC = 16;    // channels per K-chunk
  static constexpr int KPL = 1;    // k elements per lane per MFMA operand
  static constexpr int KCP = 17;   // padded LDS row (elements): conflict-free ds_read_b32
  static constexpr int VEC = 4;    // elements per 16-byte global vector
  static constexpr int WP = 64;    // wgrad LDS row (elements)
  using frag = float;
};


template <typename T>
__device__ __forceinline__ typename Cfg<T>::frag lds_frag(const T* p) {
  return *reinterpret_cast<const typename Cfg<T>::frag*>(p);
}

// 16 bytes of T moved global -> register -> LDS
template <typename T>
__device__ __forceinline__ void lds_put(T* dst, const uint4& v);
template <>
__device__ __forceinline__ void lds_put<float>(float* dst, const uint4& v) {
  // rows are padded to 17 floats -> scalar stores
  dst[0] = __uint_as_float(v.x);
  dst[1] = __uint_as_float(v.y);
  dst[2] = __uint_as_float(v.z);
  dst[3] = __uint_as_float(v.w);
}
template <typename T>
__device__ __forceinline__ uint4 pack_emb(const float* e);
template <>
__device__ __forceinline__ uint4 pack_emb<float>(const float* e) {
  return *reinterpret_cast<const uint4*>(e);
}

template <typename T>
__global__ __launch_bounds__(NT) void conv3x3_igemm_kernel(ConvP p) {
  using C = Cfg<T>;
  constexpr int KC = C::KC, KPL = C::KPL, KCP = C::KCP, VEC = C::VEC;
  constexpr int KSTEP = 2 * KPL;             // k covered by one MFMA
  constexpr int VPP = KC / VEC;              // 16-byte vectors per pixel / per weight row
  constexpr int HALO_V = HALO * VPP;         // vectors in the halo tile
  constexpr int HALO_PT = (HALO_V + NT - 1) / NT;
  constexpr int W_V = 9 * BN * VPP;          // vectors in the weight slab
  constexpr int W_PT = W_V / NT;
  static_assert(W_V % NT == 0, "weight slab must divide evenly");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lds_h = reinterpret_cast<T*>(smem);                  // [HALO][KCP]
  T* lds_w = lds_h + HALO * KCP;                          // [9][BN][KCP]
  // (HALO*KCP*sizeof(T) is a multiple of 16 for both configurations)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int i32 = lane & 31, h = lane >> 5;

  int tile = blockIdx.x;
  const int txi = tile % p.tilesX;
  tile /= p.tilesX;
  const int tyi = tile % p.tilesY;
  const int n = tile / p.tilesY;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int co0 = blockIdx.y * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

  // ---- per-thread source descriptors of the halo vectors (constant over chunks) ----
  size_t h_off[HALO_PT];
  int h_dst[HALO_PT];
  int h_kv[HALO_PT];
  bool h_inb[HALO_PT];
#pragma unroll
  for (int j = 0; j < HALO_PT; ++j) {
    const int v = tid + j * NT;
    const int hp = v / VPP, kv = v % VPP;
    const int hy = hp / HW_, hx = hp % HW_;
    const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
    h_inb[j] = (v < HALO_V) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    h_off[j] = ((size_t)(n * p.H + (h_inb[j] ? gy : 0)) * p.W + (h_inb[j] ? gx : 0)) * (size_t)p.ldx;
    h_dst[j] = (v < HALO_V) ? hp * KCP + kv * VEC : -1;
    h_kv[j] = kv * VEC;
  }

  // per-thread weight vectors: v = tid + j*NT -> (tap = j*NT/(BN*VPP) + ..., row, kv); with
  // BN*VPP == NT each j is exactly one tap and (row, kv) depend on tid only.
  static_assert(BN * VPP == NT, "one tap per prefetch slot");
  const int w_row = tid / VPP, w_kv = (tid % VPP) * VEC;
  uint4 hv0 = make_uint4(0, 0, 0, 0), hv1 = hv0, hv2 = hv0;
  static_assert(HALO_PT == 3, "three halo vectors per thread");
  uint4 wv0, wv1, wv2, wv3, wv4, wv5, wv6, wv7, wv8;
  static_assert(W_PT == 9, "nine weight vectors per thread");

#define MAU_HALO_LOAD(J, DST)                                                          \
  {                                                                                    \
    const int c = c0 + h_kv[J];                                                        \
    uint4 r = make_uint4(0, 0, 0, 0);                                                  \
    if (h_inb[J]) {                                                                    \
      if (c < p.C0 || (p.E == 0 && c < p.ldx)) {                                       \
        r = *reinterpret_cast<const uint4*>(xg + h_off[J] + c);                        \
      } else if (c < p.C0 + p.E) {                                                     \
        r = pack_emb<T>(p.emb + (size_t)n * p.E + (c - p.C0));                         \
      }                                                                                \
    }                                                                                  \
    DST = r;                                                                           \
  }
#define MAU_W_LOAD(J, DST) \
  DST = *reinterpret_cast<const uint4*>(wsrc + ((size_t)(J)*p.CoutPad + co0 + w_row) * KC + w_kv);
#define MAU_ISSUE(CHUNK)                                                 \
  {                                                                      \
    const int c0 = (CHUNK)*KC;                                           \
    MAU_HALO_LOAD(0, hv0) MAU_HALO_LOAD(1, hv1) MAU_HALO_LOAD(2, hv2)    \
    const T* wsrc = wg + (size_t)(CHUNK)*9 * p.CoutPad * KC;             \
    MAU_W_LOAD(0, wv0) MAU_W_LOAD(1, wv1) MAU_W_LOAD(2, wv2)             \
    MAU_W_LOAD(3, wv3) MAU_W_LOAD(4, wv4) MAU_W_LOAD(5, wv5)             \
    MAU_W_LOAD(6, wv6) MAU_W_LOAD(7, wv7) MAU_W_LOAD(8, wv8)             \
  }
#define MAU_W_PUT(J, SRC) lds_put<T>(lds_w + ((J)*BN + w_row) * KCP + w_kv, SRC);
#define MAU_COMMIT()                                              \
  {                                                               \
    if (h_dst[0] >= 0) lds_put<T>(lds_h + h_dst[0], hv0);         \
    if (h_dst[1] >= 0) lds_put<T>(lds_h + h_dst[1], hv1);         \
    if (h_dst[2] >= 0) lds_put<T>(lds_h + h_dst[2], hv2);         \
    MAU_W_PUT(0, wv0) MAU_W_PUT(1, wv1) MAU_W_PUT(2, wv2)         \
    MAU_W_PUT(3, wv3) MAU_W_PUT(4, wv4) MAU_W_PUT(5, wv5)         \
    MAU_W_PUT(6, wv6) MAU_W_PUT(7, wv7) MAU_W_PUT(8, wv8)         \
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = 0.f;
    acc1[r] = 0.f;
  }

  // wave (wm, wn): pixels rows [wm*4, wm*4+4) of the tile x output channels [wn*32, wn*32+32)
  const T* hb = lds_h + ((wm * 4 + (i32 >> 4)) * HW_ + (i32 & 15)) * KCP + h * KPL;
  const T* wb = lds_w + (wn * 32 + i32) * KCP + h * KPL;

  MAU_ISSUE(0)
  for (int chunk = 0; chunk < p.nChunks; ++chunk) {
    __syncthreads();
    MAU_COMMIT()
    __syncthreads();
    if (chunk + 1 < p.nChunks) MAU_ISSUE(chunk + 1)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = ((tap / 3) * HW_ + (tap % 3)) * KCP;
#pragma unroll
      for (int ks = 0; ks < KC / KSTEP; ++ks) {
        const auto a0 = lds_frag<T>(hb + toff + ks * KSTEP);
        const auto a1 = lds_frag<T>(hb + 2 * HW_ * KCP + toff + ks * KSTEP);
        const auto b = lds_frag<T>(wb + tap * BN * KCP + ks * KSTEP);
        acc0 = mfma32(a0, b, acc0);
        acc1 = mfma32(a1, b, acc1);
      }
    }
  }

  // ---- epilogue: bias, BatchNorm partial statistics, store ----
  const int co = co0 + wn * 32 + i32;
  const float bv = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
  const bool post = p.post_scale != nullptr;
  const float psc = (post && co < p.Cout) ? p.post_scale[co] : 0.f, psh = (post && co < p.Cout) ? p.post_shift[co] : 0.f;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int prow = acc_row(r, h);
      const int ty = wm * 4 + mt * 2 + (prow >> 4), tx = prow & 15;
      const int gy = ty0 + ty, gx = tx0 + tx;
      const bool valid = gy < p.H && gx < p.W;
      float v = (mt == 0 ? acc0[r] : acc1[r]) + bv;
      if (post) v = fmaxf(fmaf(v, psc, psh), 0.f);
      if (valid) {
        s += v;
        q += v * v;
        if (co < p.ldy) yg[((size_t)(n * p.H + gy) * p.W + gx) * p.ldy + co] = (T)v;
      }
    }
  }
  if (p.slab != nullptr) {
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    __syncthreads();                                  // everyone is done with the LDS tiles
    float* red = reinterpret_cast<float*>(smem);      // [2 wm][2 which][64]
    if (h == 0) {
      red[(wm * 2 + 0) * 64 + wn * 32 + i32] = s;
      red[(wm * 2 + 1) * 64 + wn * 32 + i32] = q;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      p.slab[((size_t)blockIdx.x * 2 + which) * p.CoutPad + co0 + c] = red[(0 * 2 + which) * 64 + c] + red[(1 * 2 + which) * 64 + c];
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
template <typename T>
struct WFrag;
template <>
struct WFrag<float> {
  // A[i][k]: lane (i = l&31, k = l>>5) -- one element, rows of the LDS image are pixels (k)
  static __device__ __forceinline__ float load(const float* img, int pix0, int col0, int lane) {
    return img[(pix0 + (lane >> 5)) * Cfg<float>::WP + col0 + (lane & 31)];
  }
  static constexpr int KSTEP = 2;
};

template <typename T>
__global__ __launch_bounds__(NT) void conv3x3_wgrad_kernel(WgradP p) {
  using C = Cfg<T>;
  constexpr int VEC = C::VEC, WP = C::WP;
  constexpr int VPR = 64 / VEC;                 // 16-byte vectors per 64-channel row
  constexpr int DY_V = TH * TW * VPR, X_V = HALO * VPR;
  constexpr int KSTEP = WFrag<T>::KSTEP;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* dyt = reinterpret_cast<T*>(smem);          // [128][WP]
  T* xh = dyt + TH * TW * WP;                   // [180][WP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave & 1, wci = wave >> 1;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int tile = blockIdx.x; tile < p.nTiles; tile += gridDim.x) {
    int tt = tile;
    const int txi = tt % p.tilesX;
    tt /= p.tilesX;
    const int tyi = tt % p.tilesY;
    const int n = tt / p.tilesY;
    const int ty0 = tyi * TH, tx0 = txi * TW;

    __syncthreads();
    for (int v = tid; v < DY_V; v += NT) {
      const int pix = v / VPR, kv = (v % VPR) * VEC;
      const int gy = ty0 + pix / TW, gx = tx0 + pix % TW;
      const int c = co0 + kv;
      uint4 u = make_uint4(0, 0, 0, 0);
      if (gy < p.H && gx < p.W && c < p.lddy)
        u = *reinterpret_cast<const uint4*>(dyg + ((size_t)(n * p.H + gy) * p.W + gx) * p.lddy + c);
      *reinterpret_cast<uint4*>(dyt + pix * WP + kv) = u;
    }
    for (int v = tid; v < X_V; v += NT) {
      const int hp = v / VPR, kv = (v % VPR) * VEC;
      const int gy = ty0 + hp / HW_ - 1, gx = tx0 + hp % HW_ - 1;
      const int c = ci0 + kv;
      uint4 r = make_uint4(0, 0, 0, 0);
      if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
        if (c < p.C0 || (p.E == 0 && c < p.ldx))
          r = *reinterpret_cast<const uint4*>(xg + ((size_t)(n * p.H + gy) * p.W + gx) * p.ldx + c);
        else if (c < p.C0 + p.E)
          r = pack_emb<T>(p.emb + (size_t)n * p.E + (c - p.C0));
      }
      *reinterpret_cast<uint4*>(xh + hp * WP + kv) = r;
    }
    __syncthreads();

#pragma unroll 1
    for (int trow = 0; trow < TH; ++trow) {
#pragma unroll
      for (int k0 = 0; k0 < TW; k0 += KSTEP) {
        const auto a = WFrag<T>::load(dyt, trow * TW + k0, wco * 32, lane);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const auto b = WFrag<T>::load(xh, (trow + tap / 3) * HW_ + k0 + tap % 3, wci * 32, lane);
          acc[tap] = mfma32(a, b, acc[tap]);
        }
      }
    }
  }

  // partial slab of this split: [tap][CoutPad][CinPad], 128 contiguous bytes per half-wave, plain stores
  const int h = lane >> 5;
  float* out = p.acc + (size_t)blockIdx.x * 9 * p.CoutPad * p.CinPad;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + acc_row(r, h);
      const int ci = ci0 + wci * 32 + (lane & 31);
      out[((size_t)tap * p.CoutPad + co) * p.CinPad + ci] = acc[tap][r];
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight packing / gradient unpacking
// ------------------------------------------------------------------------------------------
// bf16 packs: position q of every 64-row block of the "output" dimension holds channel 2*(q & 31) + (q >> 5), so that
// the two 32-column accumulator tiles of an MFMA lane carry ADJACENT channels (the bf16 kernel's epilogue converts and
// stores them as one pair); the fp32 parity kernels keep the natural order.
template <typename T>
__device__ __forceinline__ int pack_row_channel(int q) {
  return sizeof(T) == 2 ? (q & ~63) + 2 * (q & 31) + ((q >> 5) & 1) : q;
}

// One block = one tile of a pack: 64 row positions x one 16-channel K chunk x 9 taps, staged through LDS so that both the
// fp32 OIHW reads (runs of 144 / 576 contiguous floats) and the packed writes (2 KiB contiguous per tap) are coalesced.
//   forward tile  (blockIdx.x <  tilesF): rows = 64 output channels, k = 16 input channels, tap t  -> wf[ci/16][t][co'][ci%16]
//   dgrad tile    (blockIdx.x >= tilesF): rows = 64 input channels,  k = 16 output channels, tap t -> wd[co/16][8-t][ci'][co%16]
// 1024 threads per block: 9 loads per thread in flight at once (with 256 threads a block's 36 dependent load rounds
// put an 8.5 us floor under every launch).
template <typename T>
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout, int Cin,
                                          int tilesF, int tidx, float (*tile)[PackKC<T>::value * 9 + 1]) {
  constexpr int KC = PackKC<T>::value;                 // 16
  const int CoutPad = (Cout + 63) / 64 * 64, CinPad = (Cin + 63) / 64 * 64;
  const bool fwd = tidx < tilesF;
  const int t = fwd ? tidx : tidx - tilesF;
  const int rowBlocks = (fwd ? CoutPad : CinPad) / 64;
  const int rb = t % rowBlocks, chunk = t / rowBlocks;
  const int nRow = fwd ? Cout : Cin, nK = fwd ? Cin : Cout;
  // ---- load: tile[r][k*9 + tap] ----
  if (fwd) {
    for (int e = threadIdx.x; e < 64 * KC * 9; e += 1024) {
      const int r = e / (KC * 9), kt = e % (KC * 9);
      const int co = rb * 64 + r, ci = chunk * KC + kt / 9;
      tile[r][kt] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + chunk * KC) * 9 + kt] : 0.f;
    }
  } else {
    for (int e = threadIdx.x; e < KC * 64 * 9; e += 1024) {
      const int k = e / (64 * 9), rt = e % (64 * 9);
      const int r = rt / 9, tap = rt % 9;
      const int co = chunk * KC + k, ci = rb * 64 + r;
      tile[r][k * 9 + tap] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + rb * 64) * 9 + rt] : 0.f;
    }
  }
  __syncthreads();
  // ---- store: out[((chunk*9 + tapOut) * RowPad + rb*64 + pos) * KC + k], 8 elements (k .. k+7) per thread and step ----
  T* out = fwd ? wf : wd;
  const int RowPad = fwd ? CoutPad : CinPad;
  for (int e = threadIdx.x; e < 9 * 64 * (KC / 8); e += 1024) {
    const int k8 = e % (KC / 8), pos = (e / (KC / 8)) % 64, tap = e / (64 * (KC / 8));
    const int r = sizeof(T) == 2 ? 2 * (pos & 31) + (pos >> 5) : pos;          // channel held by position pos (pack_row_channel)
    const int tapOut = fwd ? tap : 8 - tap;
    F8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] = tile[r][(k8 * 8 + j) * 9 + tap];
    store8<T>(out + (((size_t)chunk * 9 + tapOut) * RowPad + rb * 64 + pos) * KC + k8 * 8, v);
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                                           int Cout, int Cin, int tilesF) {
  __shared__ float tile[64][PackKC<T>::value * 9 + 1];
  pack_tile<T>(w, wf, wd, Cout, Cin, tilesF, blockIdx.x, tile);
}

// Every conv layer of a network in ONE launch (the optimizer has just changed all of them): a device-resident table of
// (weights, packs, shape, first tile) rows, sorted by first tile; a workgroup finds its layer by a scan of the table
// (wave-uniform scalar loads; <= a few dozen rows) and packs one tile of it exactly as pack_weights_kernel does.
struct PackDesc {
  const float* w;
  void* wf;
  void* wd;
  int Cout, Cin, tilesF, tile0;
};
template <typename T>
__global__ __launch_bounds__(1024) void pack_weights_multi_kernel(const PackDesc* __restrict__ descs, int n) {
  __shared__ float tile[64][PackKC<T>::value * 9 + 1];
  int i = 0;
  while (i + 1 < n && (int)blockIdx.x >= descs[i + 1].tile0) ++i;
  const PackDesc d = descs[i];
  pack_tile<T>(d.w, (T*)d.wf, (T*)d.wd, d.Cout, d.Cin, d.tilesF, (int)blockIdx.x - d.tile0, tile);
}

// sum of the split-K partial slabs [nsplit][9][CoutPad][CinPad] (fixed order) -> OIHW (Cout,Cin,3,3).
// One thread per (co, ci): the nsplit x 9 reads are coalesced over ci, the 9 results are one
// contiguous 36-byte run of the output (consecutive threads -> consecutive runs).
// acc [nsplit][9][CoutPad][CinPad] -> dw (Cout, Cin, 3, 3) = sum over the splits, in a fixed order.
// One block = one output channel x 64 input channels: G wave-sized groups each add every G-th split slab
// (coalesced 256-byte reads), the partial sums meet in LDS, and the 64 x 9 results leave as ONE contiguous
// 2304-byte run of dw (the (ci, tap) order of the reference's weight tensor).
template <int G>
__global__ __launch_bounds__(64 * G) void unpack_wgrad_kernel(const float* __restrict__ acc, int nsplit, float* __restrict__ dw,
                                                              int Cout, int Cin, int CoutPad, int CinPad) {
  __shared__ float red[G][9][64];
  const int cib = blockIdx.x % (CinPad / 64), co = blockIdx.x / (CinPad / 64);
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t plane = (size_t)CoutPad * CinPad, slab = 9 * plane;
  const float* src = acc + (size_t)co * CinPad + cib * 64 + lane;
  float v[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) v[t] = 0.f;
  for (int sp = grp; sp < nsplit; sp += G) {
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] += src[(size_t)sp * slab + (size_t)t * plane];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[grp][t][lane] = v[t];
  __syncthreads();
  const int nci = min(64, Cin - cib * 64);
  for (int j = threadIdx.x; j < nci * 9; j += 64 * G) {
    const int cl = j / 9, t = j - cl * 9;
    float o = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) o += red[g][t][cl];
    dw[((size_t)co * Cin + cib * 64) * 9 + j] = o;
  }
}

template <typename T>
static size_t conv_lds_bytes() {
  return (size_t)(HALO + 9 * BN) * Cfg<T>::KCP * sizeof(T);
}
template <typename T>
static size_t wgrad_lds_bytes() {
  return (size_t)(TH * TW + HALO) * Cfg<T>::WP * sizeof(T);
}

template <typename T>
static int launch_conv(const ConvP& p, hipStream_t st) {
  const size_t lds = conv_lds_bytes<T>();
  MAU_LDS_ATTR(lds, &conv3x3_igemm_kernel<T>);
  dim3 grid(p.N * p.tilesX * p.tilesY, p.CoutPad / BN);
  MAU_LAUNCH(conv3x3_igemm_kernel<T>, grid, dim3(NT), lds, st, p);
  return check_launch("conv3x3_igemm_kernel");
}

// split-K of the parity-mode weight gradient: ~4 workgroups per CU, at most one per pixel tile, slabs under 256 MiB
static int wgrad_f32_splits(int N, int H, int W, int Cout, int Cin) {
  const int CoutPad = round_up(Cout, 64), CinPad = round_up(Cin, 64);
  const int tilesOut = (CoutPad / 64) * (CinPad / 64);
  const int nTiles = N * ceil_div(H, TH) * ceil_div(W, TW);
  int splits = (device_shape().cus * 4 + tilesOut - 1) / tilesOut;
  const size_t cap = ((size_t)256 << 20) / ((size_t)9 * CoutPad * CinPad * sizeof(float));
  if ((size_t)splits > cap) splits = (int)cap;
  if (splits > nTiles) splits = nTiles;
  return splits < 1 ? 1 : splits;
}

template <typename T>
static int launch_wgrad(const WgradP& p, hipStream_t st) {
  const size_t lds = wgrad_lds_bytes<T>();
  MAU_LDS_ATTR(lds, &conv3x3_wgrad_kernel<T>);
  const int splits = wgrad_f32_splits(p.N, p.H, p.W, p.Cout, p.Cin);
  dim3 grid(splits, p.CoutPad / 64, p.CinPad / 64);
  MAU_LAUNCH(conv3x3_wgrad_kernel<T>, grid, dim3(NT), lds, st, p);
  return check_launch("conv3x3_wgrad_kernel");
}

}  // namespace mau

using namespace mau;

template <typename T>
__global__ void cast_f32_to_lp_kernel(const float* __restrict__ src, T* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (T)src[i];
}
static void cast_emb(const float* emb, void* ws, int n, bool f16v, hipStream_t st) {
  if (f16v) MAU_LAUNCH(cast_f32_to_lp_kernel<f16>, dim3(ceil_div(n, 256)), dim3(256), 0, st, emb, (f16*)ws, n);
  else MAU_LAUNCH(cast_f32_to_lp_kernel<bf16>, dim3(ceil_div(n, 256)), dim3(256), 0, st, emb, (bf16*)ws, n);
}

extern "C" {

int mau_conv3x3_kc(int dtype) { return dtype == MAU_F32 ? PackKC<float>::value : PackKC<bf16>::value; }

size_t mau_conv3x3_packed_elems(int dtype, int nout, int nin) {
  const int kc = mau_conv3x3_kc(dtype);
  return (size_t)ceil_div(nin, kc) * 9 * round_up(nout, 64) * kc;
}

int mau_conv3x3_pack_weights(const float* w, void* wf, void* wd, int dtype, int Cout, int Cin,
                             mau_stream_t stream) {
  MAU_REQUIRE(w && (wf || wd) && Cout > 0 && Cin > 0, "pack_weights: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int kc = mau_conv3x3_kc(dtype);
  const int tilesF = wf ? (round_up(Cout, 64) / 64) * ceil_div(Cin, kc) : 0;
  const int tilesD = wd ? (round_up(Cin, 64) / 64) * ceil_div(Cout, kc) : 0;
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(pack_weights_kernel<T>, dim3(tilesF + tilesD), dim3(1024), 0, st, w, (T*)wf, (T*)wd, Cout, Cin, tilesF));
  return check_launch("pack_weights_kernel");
}

size_t mau_conv3x3_pack_desc_bytes(void) { return sizeof(PackDesc); }

int mau_conv3x3_pack_desc_fill(void* descs_host, int index, const float* w, void* wf, void* wd, int dtype, int Cout, int Cin,
                               int tile0, int* next_tile_host) {
  MAU_REQUIRE(descs_host && next_tile_host && index >= 0 && w && (wf || wd) && Cout > 0 && Cin > 0 && tile0 >= 0, "pack_desc_fill: bad arguments");
  MAU_REQUIRE(dtype == MAU_F32 || dtype == MAU_BF16 || dtype == MAU_F16, "bad dtype %d", dtype);
  const int kc = mau_conv3x3_kc(dtype);
  const int tilesF = wf ? (round_up(Cout, 64) / 64) * ceil_div(Cin, kc) : 0;
  const int tilesD = wd ? (round_up(Cin, 64) / 64) * ceil_div(Cout, kc) : 0;
  PackDesc* d = reinterpret_cast<PackDesc*>(descs_host) + index;
  d->w = w; d->wf = wf; d->wd = wd; d->Cout = Cout; d->Cin = Cin; d->tilesF = tilesF; d->tile0 = tile0;
  *next_tile_host = tile0 + tilesF + tilesD;
  return MAU_OK;
}

int mau_conv3x3_pack_weights_multi(const void* descs, int n, int total_tiles, int dtype, mau_stream_t stream) {
  MAU_REQUIRE(descs && n > 0 && total_tiles > 0, "pack_weights_multi: bad arguments");
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(pack_weights_multi_kernel<T>, dim3(total_tiles), dim3(1024), 0, (hipStream_t)stream,
                                       (const PackDesc*)descs, n));
  return check_launch("pack_weights_multi_kernel");
}

int mau_conv3x3_variant(int dtype, int N, int H, int W, int Cin, int Cout, int* tile_rows_host, int* waves_host, int* cout_block_host,
                        int* k_groups_host) {
  MAU_REQUIRE(tile_rows_host && waves_host && cout_block_host && k_groups_host && N > 0 && H > 0 && W > 0 && Cin >= 0 && Cout > 0,
              "conv3x3_variant: bad arguments");
  if (dtype == MAU_F32) {
    *tile_rows_host = TH;
    *waves_host = 4;
    *cout_block_host = 64;
    *k_groups_host = 1;
    return MAU_OK;
  }
  conv_bf16_v2_variant(N, H, W, Cin, Cout, tile_rows_host, waves_host, cout_block_host, k_groups_host);
  return MAU_OK;
}

int mau_conv3x3_num_pixel_tiles(int dtype, int N, int H, int W, int Cout) {
  return dtype != MAU_F32 ? conv_bf16_v2_num_pixel_tiles(N, H, W, Cout) : N * ceil_div(H, TH) * ceil_div(W, TW);
}

int mau_conv3x3_fwd2(const void* x, int ldx, int C0, const void* x1, int ldx1, int C1, const float* emb, void* emb_ws, int E,
                     const void* wpk, const float* bias, const float* post_scale, const float* post_shift, void* y, int ldy,
                     int Cout, float* slab, int dtype, int N, int H, int W, mau_stream_t stream) {
  MAU_REQUIRE(x && wpk && y, "conv3x3_fwd: null pointer");
  MAU_REQUIRE(N > 0 && H > 0 && W > 0 && C0 > 0 && Cout > 0, "conv3x3_fwd: bad shape");
  MAU_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C0 && ldy >= Cout, "conv3x3_fwd: ld must be a multiple of 8 and >= C");
  MAU_REQUIRE(C1 >= 0 && (C1 == 0 || (x1 && dtype != MAU_F32 && C0 % 16 == 0 && ldx1 % 8 == 0 && ldx1 >= C1 && ((uintptr_t)x1 % 16) == 0)),
              "conv3x3_fwd: a second tensor source needs a 16-bit dtype, C0 %% 16 == 0 and an aligned x1 with ldx1 %% 8 == 0");
  MAU_REQUIRE(E >= 0 && (E == 0 || (emb && E % 8 == 0 && (C0 + C1) % 8 == 0)), "conv3x3_fwd: broadcast source needs E%%8==0 and (C0+C1)%%8==0");
  MAU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)wpk % 16) == 0, "conv3x3_fwd: pointers must be 16-byte aligned");
  MAU_REQUIRE((post_scale == nullptr) == (post_shift == nullptr), "conv3x3_fwd: post_scale and post_shift come together");
  ConvP p;
  p.x = x; p.ldx = ldx; p.C0 = C0; p.x1 = C1 > 0 ? x1 : nullptr; p.ldx1 = C1 > 0 ? ldx1 : 0; p.C1 = C1;
  p.emb = emb; p.emb_lp = nullptr; p.E = E; p.w = wpk; p.bias = bias; p.post_scale = post_scale; p.post_shift = post_shift; p.y = y; p.ldy = ldy;
  p.Cout = Cout; p.CoutPad = round_up(Cout, 64); p.slab = slab; p.N = N; p.H = H; p.W = W;
  p.tilesX = ceil_div(W, TW); p.tilesY = ceil_div(H, TH);
  p.nChunks = ceil_div(C0 + C1 + E, mau_conv3x3_kc(dtype));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MAU_F32) return launch_conv<float>(p, st);
  if (dtype == MAU_BF16 || dtype == MAU_F16) {
    if (E > 0) {
      MAU_REQUIRE(emb_ws != nullptr && ((uintptr_t)emb_ws % 16) == 0, "conv3x3_fwd: 16-bit broadcast source needs the (N,E) workspace emb_ws");
      cast_emb(emb, emb_ws, N * E, dtype == MAU_F16, st);
      p.emb_lp = emb_ws;
    }
    return launch_conv_bf16_v2(p, dtype == MAU_F16, st);
  }
  set_error("bad dtype %d", dtype);
  return MAU_ERR_ARG;
}

int mau_conv3x3_fwd_pool(const void* x, int ldx, int C0, const void* wpk, const float* bias, const float* post_scale, const float* post_shift,
                         void* y, int ldy, int Cout, void* pooled, int ldpool, int dtype, int N, int H, int W, mau_stream_t stream) {
  MAU_REQUIRE(x && wpk && y && pooled && post_scale && post_shift, "conv3x3_fwd_pool: null pointer");
  MAU_REQUIRE(N > 0 && H >= 2 && W >= 2 && C0 > 0 && Cout > 0, "conv3x3_fwd_pool: bad shape");
  MAU_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldpool % 8 == 0 && ldx >= C0 && ldy >= Cout && ldpool >= round_up(Cout, 8),
              "conv3x3_fwd_pool: ld must be a multiple of 8 and >= C");
  MAU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)wpk % 16) == 0 && ((uintptr_t)pooled % 16) == 0,
              "conv3x3_fwd_pool: pointers must be 16-byte aligned");
  if (dtype == MAU_F32) {                              // parity mode: the two launches
    const int rc = mau_conv3x3_fwd2(x, ldx, C0, nullptr, 0, 0, nullptr, nullptr, 0, wpk, bias, post_scale, post_shift, y, ldy, Cout, nullptr, dtype, N, H, W, stream);
    return rc != MAU_OK ? rc : mau_maxpool2x2_fwd(y, ldy, pooled, ldpool, dtype, N, H, W, Cout, stream);
  }
  MAU_REQUIRE(dtype == MAU_BF16 || dtype == MAU_F16, "bad dtype %d", dtype);
  ConvP p;
  p.x = x; p.ldx = ldx; p.C0 = C0; p.x1 = nullptr; p.ldx1 = 0; p.C1 = 0;
  p.emb = nullptr; p.emb_lp = nullptr; p.E = 0; p.w = wpk; p.bias = bias; p.post_scale = post_scale; p.post_shift = post_shift; p.y = y; p.ldy = ldy;
  p.Cout = Cout; p.CoutPad = round_up(Cout, 64); p.slab = nullptr; p.N = N; p.H = H; p.W = W;
  p.tilesX = ceil_div(W, TW); p.tilesY = ceil_div(H, TH);
  p.nChunks = ceil_div(C0, mau_conv3x3_kc(dtype));
  p.pool = pooled; p.ldpool = ldpool;
  return launch_conv_bf16_v2(p, dtype == MAU_F16, (hipStream_t)stream);
}

int mau_conv3x3_fwd(const void* x, int ldx, int C0, const float* emb, void* emb_ws, int E, const void* wpk,
                    const float* bias, const float* post_scale, const float* post_shift, void* y, int ldy, int Cout,
                    float* slab, int dtype, int N, int H, int W, mau_stream_t stream) {
  return mau_conv3x3_fwd2(x, ldx, C0, nullptr, 0, 0, emb, emb_ws, E, wpk, bias, post_scale, post_shift, y, ldy, Cout, slab, dtype,
                          N, H, W, stream);
}

int mau_conv3x3_wgrad_splits(int dtype, int N, int H, int W, int Cout, int Cin) {
  return dtype != MAU_F32 ? wgrad_bf16_v2_splits(N, H, W, Cout, Cin) : wgrad_f32_splits(N, H, W, Cout, Cin);
}

size_t mau_conv3x3_wgrad_acc_elems(int dtype, int N, int H, int W, int Cout, int Cin) {
  return (size_t)mau_conv3x3_wgrad_splits(dtype, N, H, W, Cout, Cin) * 9 * round_up(Cout, 64) * round_up(Cin, 64);
}

int mau_conv3x3_wgrad2(const void* x, int ldx, int C0, const void* x1, int ldx1, int C1, const float* emb, void* emb_ws, int E,
                       const void* dy, int lddy, int Cout, float* acc, int dtype, int N, int H, int W, mau_stream_t stream) {
  MAU_REQUIRE(x && dy && acc, "conv3x3_wgrad: null pointer");
  MAU_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && ldx >= C0 && lddy >= Cout, "conv3x3_wgrad: bad ld");
  MAU_REQUIRE(C1 >= 0 && (C1 == 0 || (x1 && dtype != MAU_F32 && C0 % 8 == 0 && ldx1 % 8 == 0 && ldx1 >= C1 && ((uintptr_t)x1 % 16) == 0)),
              "conv3x3_wgrad: a second tensor source needs a 16-bit dtype, C0 %% 8 == 0 and an aligned x1 with ldx1 %% 8 == 0");
  MAU_REQUIRE(E >= 0 && (E == 0 || (emb && E % 8 == 0 && (C0 + C1) % 8 == 0)), "conv3x3_wgrad: broadcast source needs E%%8==0 and (C0+C1)%%8==0");
  MAU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "conv3x3_wgrad: pointers must be 16-byte aligned");
  WgradP p;
  p.x = x; p.ldx = ldx; p.C0 = C0; p.x1 = C1 > 0 ? x1 : nullptr; p.ldx1 = C1 > 0 ? ldx1 : 0; p.C1 = C1;
  p.emb = emb; p.emb_lp = nullptr; p.E = E; p.dy = dy; p.lddy = lddy; p.Cout = Cout;
  p.CoutPad = round_up(Cout, 64); p.Cin = C0 + C1 + E; p.CinPad = round_up(C0 + C1 + E, 64); p.acc = acc;
  p.N = N; p.H = H; p.W = W; p.tilesX = ceil_div(W, TW); p.tilesY = ceil_div(H, TH);
  p.nTiles = N * p.tilesX * p.tilesY;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MAU_BF16 || dtype == MAU_F16) {
    if (E > 0) {
      MAU_REQUIRE(emb_ws != nullptr && ((uintptr_t)emb_ws % 16) == 0, "conv3x3_wgrad: 16-bit broadcast source needs the (N,E) workspace emb_ws");
      cast_emb(emb, emb_ws, N * E, dtype == MAU_F16, st);
      p.emb_lp = emb_ws;
    }
    return launch_wgrad_bf16_v2(p, dtype == MAU_F16, st);        // split-K partial slabs, plain stores (no memset needed)
  }
  MAU_REQUIRE(dtype == MAU_F32, "bad dtype %d", dtype);
  return launch_wgrad<float>(p, st);                               // split-K partial slabs, plain stores (no memset needed)
}

int mau_conv3x3_wgrad(const void* x, int ldx, int C0, const float* emb, void* emb_ws, int E, const void* dy, int lddy,
                      int Cout, float* acc, int dtype, int N, int H, int W, mau_stream_t stream) {
  return mau_conv3x3_wgrad2(x, ldx, C0, nullptr, 0, 0, emb, emb_ws, E, dy, lddy, Cout, acc, dtype, N, H, W, stream);
}

int mau_conv3x3_unpack_wgrad(const float* acc, int nsplit, float* dw, int Cout, int Cin, mau_stream_t stream) {
  MAU_REQUIRE(acc && dw && Cout > 0 && Cin > 0 && nsplit >= 1, "unpack_wgrad: bad arguments");
  const int CoutPad = round_up(Cout, 64), CinPad = round_up(Cin, 64);
  const int grid = Cout * (CinPad / 64);
  if (nsplit >= 16) {
    MAU_LAUNCH(unpack_wgrad_kernel<16>, dim3(grid), dim3(1024), 0, (hipStream_t)stream, acc, nsplit, dw, Cout, Cin, CoutPad, CinPad);
  } else if (nsplit >= 4) {
    MAU_LAUNCH(unpack_wgrad_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, acc, nsplit, dw, Cout, Cin, CoutPad, CinPad);
  } else {
    MAU_LAUNCH(unpack_wgrad_kernel<1>, dim3(grid), dim3(64), 0, (hipStream_t)stream, acc, nsplit, dw, Cout, Cin, CoutPad, CinPad);
  }
  return check_launch("unpack_wgrad_kernel");
}

}  // extern "C"
