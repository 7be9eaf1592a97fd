// lstm.hip -- TemporalEncoder's recurrence (reference src/model.py:23-34: nn.LSTM(input_size=1, hidden_size=H,
// batch_first=True), last hidden state) as ONE persistent launch per direction.
//
// The reference's sequence is 828 monthly temperatures (conf/config.yaml:20) and the hidden size 96: 828 DEPENDENT steps of
// a 384x96 matrix-vector product per sample -- pure latency, 2.4 MFLOP per step.  A library LSTM launches kernels per
// step (measured: 26-30 ms forward + backward at B=32, more than the whole U-Net step).  Here:
//   forward  : one workgroup per sample, 4*HP threads, lane quad = hidden unit u; lane q multiplies the q-th QUARTER of h
//              with the matching quarter of all four gate rows of the unit (4 x HP/4 weights in registers for the whole
//              sequence; HP/4 values of h read from LDS per step instead of HP), quad butterflies (DPP) add the partials,
//              lane q applies gate q's nonlinearity, the activations are exchanged inside the quad: ONE barrier per step;
//              gate order i, f, g, o (torch)
//   backward : one workgroup per sample; the saved activations of 8 steps are staged into LDS a block ahead and the gradients
//              leave from LDS in bulk (no global memory inside the step loop); everything of a step that does not depend on
//              (dh, dc) is computed one step ahead; the W_hh^T product is laid out over 16-lane groups (4 units x 1/16 of the
//              4H gate gradients: a quarter of the LDS reads of a column per thread); ONE barrier per step.  The weight
//              gradient dW_hh = sum_t dpre_t (x) h_{t-1} is NOT part of the recurrence -- it is one [4H x B*T] x [B*T x H]
//              product computed afterwards by a tiled kernel (h_{t-1} = o_{t-1} * tanh(c_{t-1}) recomputed from the saved
//              gates, no h tensor).  All partial sums are added in a fixed order: bitwise reproducible, no atomics.
//   measured : B=32, T=828, H=96 (scripts/lstm_bench.py): forward 0.44 ms (533 ns/step), backward 0.92 ms; the step is a
//              dependency chain on ONE CU (LDS exchange, ~100 scalar FMAs per thread, exp + rcp, barrier: ~1250 cycles), not a
//              throughput problem -- two gate rows per thread measured 23 % slower, packed FMAs (also hipcc's SLP-packed ones:
//              built with -fno-slp-vectorize) slower than scalar.  In training the model runs it on a side stream.
// Arithmetic is fp32 throughout; gate nonlinearities on the hardware exp / rcp.
#include "mau_common.h"

namespace mau {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Gate nonlinearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): absolute error ~1e-7 per evaluation,
// which the contractive recurrence does not amplify (G11: 828 steps, embedding and gradients within 1e-4 of the reference);
// libm's expf / tanhf cost ~100 instructions and 50 branches per call on the critical path of every step.
#ifdef MAU_LSTM_LIBM
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
#else
// (v_rcp_f32 directly: __frcp_rn expands to a 12-instruction IEEE division)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
#endif

// value of lane (quad base + J) of every quad, for all four lanes of the quad (DPP quad_perm: no LDS, no barrier)
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), J * 0x55, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the lane quad, in every lane of the quad: butterfly with quad_perm [1,0,3,2] then [2,3,0,1]
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  return v;
}

// sum over each aligned group of 16 lanes, in all 16: the quad butterfly, then row rotations by 4 and 8 (DPP row_ror)
__device__ __forceinline__ float row16_sum(float v) {
  v = quad_sum(v);
  v += dpp_mov<0x124>(v);
  v += dpp_mov<0x128>(v);
  return v;
}

// x (B,T); w_ih (4H) [input_size 1]; w_hh (4H,H); b_ih, b_hh (4H); h_last (B,H);
// gates (B,T,4H) post-activation i,f,g,o and cells (B,T,H) are written when non-null (saved for backward).
// Thread 4u + q: the lane QUAD of unit u.  Lane q multiplies the q-th QUARTER of h with the matching quarter of ALL FOUR gate
// rows of the unit (4 x HP/4 weights in registers) -- it reads HP/4 values of h from LDS instead of HP (the broadcast reads
// of h were the step's longest pole: the LDS pipe serves 16 B x 16 lanes per cycle whether or not the addresses coincide)
// -- the four partial products of a gate are added inside the quad (DPP), lane q then applies gate q's nonlinearity, the
// activations are exchanged inside the quad and every lane carries the unit's cell state redundantly.  One barrier per step.
template <int HP>
__global__ __launch_bounds__(4 * HP) void lstm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                         const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                         const float* __restrict__ b_hh, float* __restrict__ h_last,
                                                         float* __restrict__ gates, float* __restrict__ cells, int T, int H) {
  constexpr int QK = HP / 4;                           // k values per lane
  static_assert(QK % 4 == 0, "quarter of h in 16-byte reads");
  __shared__ __attribute__((aligned(16))) float hs[2][HP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int u = tid >> 2, q = tid & 3;                // unit u; lane q: k quarter q, nonlinearity of gate q (i, f, g, o)
  const bool live = u < H;
  f32x2 w[4][QK / 2];                                  // w[g][kk]: row g*H + u, k = q*QK + kk (pairs: v_pk_fma_f32)
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kk = 0; kk < QK; ++kk) {
      const int k = q * QK + kk;
      w[g][kk >> 1][kk & 1] = (live && k < H) ? w_hh[(size_t)(g * H + u) * H + k] : 0.f;
    }
  const int row = q * H + u;                           // the gate row whose activation this lane produces
  const float wi = live ? w_ih[row] : 0.f;
  const float bias = live ? b_ih[row] + b_hh[row] : 0.f;
  const float gsc = q == 2 ? 2.f : 1.f;                // tanh(x) = 2 sigmoid(2x) - 1: one exp + rcp for every gate
  if (tid < HP) hs[0][tid] = 0.f;
  float c = 0.f, hval = 0.f;
  const float* xb = x + (size_t)b * T;
  __syncthreads();
  float xt = xb[0];
  for (int t = 0; t < T; ++t) {
    const float xn = t + 1 < T ? xb[t + 1] : 0.f;      // next input in flight while this step multiplies
    const float* hq = hs[t & 1] + q * QK;
    f32x2 p[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < QK; kk += 4) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(&hq[kk]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#ifdef MAU_LSTM_PACKED
        p[g] = __builtin_elementwise_fma(w[g][kk >> 1], hv.xy, p[g]);
        p[g] = __builtin_elementwise_fma(w[g][(kk >> 1) + 1], hv.zw, p[g]);
#else
        // scalar FMAs on purpose: a wave64 v_pk_fma_f32 costs more issue cycles than the two v_fma_f32 it replaces
        p[g][0] = fmaf(w[g][kk >> 1][0], hv[0], p[g][0]);
        p[g][1] = fmaf(w[g][kk >> 1][1], hv[1], p[g][1]);
        p[g][0] = fmaf(w[g][(kk >> 1) + 1][0], hv[2], p[g][0]);
        p[g][1] = fmaf(w[g][(kk >> 1) + 1][1], hv[3], p[g][1]);
#endif
      }
    }
    // gate g's pre-activation = sum over the quad of p[g]; lane q keeps gate q's
    const float s0 = quad_sum(p[0][0] + p[0][1]), s1 = quad_sum(p[1][0] + p[1][1]);
    const float s2 = quad_sum(p[2][0] + p[2][1]), s3 = quad_sum(p[3][0] + p[3][1]);
    const float dot = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
    const float pre = dot + fmaf(wi, xt, bias);
    const float sg = sigmoidf_(gsc * pre);
    const float av = q == 2 ? fmaf(2.f, sg, -1.f) : sg;
    if (gates != nullptr && live) gates[((size_t)b * T + t) * 4 * H + row] = av;
    const float ig = quad_bcast<0>(av), fg = quad_bcast<1>(av), gg = quad_bcast<2>(av), og = quad_bcast<3>(av);
    c = fmaf(fg, c, ig * gg);
    hval = og * tanhf_(c);
    if (q == 0) {
      hs[(t + 1) & 1][u] = hval;
      if (cells != nullptr && live) cells[((size_t)b * T + t) * H + u] = c;
    }
    __syncthreads();
    xt = xn;
  }
  if (q == 0 && live) h_last[(size_t)b * H + u] = hval;
}

// dh_last (B,H) -> dpre_all (B,T,4H) pre-activation gate gradients + per-sample partials dwih_p (B,4H), db_p (B,4H).
// Thread 4k + q owns column k of gate q: W_hh[q*H + j][k], j = 0..H.  Per step, lane q of a quad turns (dh[k], dc[k])
// -- both carried redundantly by the quad -- into the pre-activation gradient of ITS gate, publishes it in LDS, multiplies
// its column with the gate's gradient vector, and the four partial products of unit k are added inside the quad (DPP),
// which leaves dh_{t-1}[k] in the registers of the lanes that need it next.  One barrier per step.
// The step loop touches NO global memory: the saved activations of LSTM_BS steps are staged into LDS a whole block ahead
// (plain loads into registers at one block boundary, written to LDS at the next: their latency is a block old by then),
// and the block's gradients leave from LDS in bulk.  (With per-step loads and stores the compiler's s_waitcnt vmcnt(0)
// at the loop head waited out an HBM store round trip in every one of the 828 steps: 1.3 us per step.)
constexpr int LSTM_BS = 8;
template <int HP>
__global__ __launch_bounds__(4 * HP) void lstm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w_hh,
                                                         const float* __restrict__ gates, const float* __restrict__ cells,
                                                         const float* __restrict__ dh_last, float* __restrict__ dpre_all,
                                                         float* __restrict__ dwih_p, float* __restrict__ db_p, int T, int H) {
  constexpr int S = LSTM_BS, NT = 4 * HP, ROW = 5 * HP;            // staged row: 4*HP gates in [unit][gate] order | HP cells
  constexpr int GP = HP + 8, OROW = 4 * GP;                         // gradient row: [gate][GP]; the pitch puts the four gates'
                                                                    // 16-byte reads of a lane group on distinct banks (with
                                                                    // pitch HP = 96 gates 0/2 and 1/3 collide: every read 2x)
  constexpr int NLD = S * ROW / NT;                                 // staged values per thread and block (20)
  static_assert(S * ROW % NT == 0, "staging loop");
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  float* inb = lsm;                                                 // [2][S][ROW]
  float* xin = inb + 2 * S * ROW;                                   // [2][S]
  float* outb = xin + 2 * S;                                        // [2][S][OROW], [gate][unit] order (the dot's read layout)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int k = tid >> 2, q = tid & 3;
  const bool live = k < H;
  // product role: the 16 lanes that hold units 4G .. 4G+3 split the 4*HP gate gradients into 16 parts of PK; lane l holds
  // the weights of its part for all four units: wt[u][i] = W_hh[g*H + j0 + i][4G + u], g = l / 4, j0 = (l % 4) * PK
  constexpr int PK = HP / 4;
  static_assert(PK % 4 == 0, "part of the gradient vector in 16-byte reads");
  const int q16 = (tid >> 2) & 3;                                    // k - 4G
  const int l16 = tid & 15, G4 = (tid >> 4) * 4, pg = l16 >> 2, pj0 = (l16 & 3) * PK;
  float wt[4][PK];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int i = 0; i < PK; ++i)
      wt[u][i] = (G4 + u < H && pj0 + i < H) ? w_hh[(size_t)(pg * H + pj0 + i) * H + G4 + u] : 0.f;
  float dc = 0.f, dwi = 0.f, dbv = 0.f;
  float dh = live ? dh_last[(size_t)b * H + k] : 0.f;
  const float* xb = x + (size_t)b * T;
  const float* gb = gates + (size_t)b * T * 4 * H;
  const float* cb = cells + (size_t)b * T * H;
  float* dpb = dpre_all + (size_t)b * T * 4 * H;
  const int nblk = (T + S - 1) / S;

  // staging of block j (steps t0 = T-1-S*j, t0-1, ...): thread i-th value = element e = tid + i*NT of the padded [S][5*HP]
  // index space (compile-time divisors; coalesced for H == HP); pad units and steps before the sequence start read as zero
  float st[NLD], sx = 0.f;
  auto stage_load = [&](int j) {
    const int t0 = T - 1 - S * j;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * NT, sl = e / ROW, r = e % ROW, t = t0 - sl;
      const int g = r / HP, u = r % HP;                 // g < 4: gate g of unit u; g == 4: cell of unit u
      float v = 0.f;
      if (t >= 0 && u < H) v = g < 4 ? gb[(size_t)t * 4 * H + g * H + u] : cb[(size_t)t * H + u];
      st[i] = v;
    }
    sx = (tid < S && t0 - tid >= 0) ? xb[t0 - tid] : 0.f;
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * NT, sl = e / ROW, r = e % ROW;
      const int g = r / HP, u = r % HP;
      inb[(buf * S + sl) * ROW + (g < 4 ? 4 * u + g : 4 * HP + u)] = st[i];        // gate g, unit u -> 4u + g
    }
    if (tid < S) xin[buf * S + tid] = sx;
  };
  for (int i = tid; i < 2 * S * ROW; i += NT) inb[i] = 0.f;          // pad units stay zero
  for (int i = tid; i < 2 * S * OROW; i += NT) outb[i] = 0.f;
  __syncthreads();
  stage_load(0);
  stage_write(0);
  if (nblk > 1) stage_load(1);
#pragma unroll 1
  for (int j = 0; j < nblk; ++j) {
    const int buf = j & 1, t0 = T - 1 - S * j;
    __syncthreads();                                  // everyone is done with in-buffer buf^1 (block j-1) and out-buffer buf^1
    if (j + 1 < nblk) stage_write(buf ^ 1);           // block j+1 (loaded a block ago)
    if (j + 2 < nblk) stage_load(j + 2);
    if (j > 0) {                                      // block j-1's gradients leave in bulk (coalesced)
      const int tp = t0 + S;                          // first step of block j-1
      for (int e = tid; e < S * 4 * HP; e += NT) {
        const int sl = e / (4 * HP), r = e % (4 * HP), g = r / HP, u = r % HP;
        if (u < H) dpb[(size_t)(tp - sl) * 4 * H + g * H + u] = outb[((buf ^ 1) * S + sl) * OROW + g * GP + u];
      }
    }
    __syncthreads();                                  // in-buffer buf^1 written: step S-1 reads its slot 0 (c_{t-1})
    const int ns = min(S, t0 + 1);
    // Everything of a step that does NOT depend on (dh, dc) is computed one step ahead, in the shadow of the previous
    // step's product: the per-step dependency chain is  dh -> fma -> mul -> ds_write -> barrier -> product -> quad sum.
    //   dct = dh * A + dc,  dp = (q == 3 ? dh : dct) * Bq,  dc' = dct * fg
    //   A = o (1 - tanh(c)^2);  Bq = {g, c_prev, i, tanh(c)}[q] * (q == 2 ? 1 - a^2 : a (1 - a)),  a = this lane's gate
    auto coeffs = [&](int sl, float& A, float& Bq, float& fgv, float& xv) {
      const int t = t0 - sl;
      const float* row = inb + (buf * S + sl) * ROW;
      const float av = row[tid];
      const float ct = row[4 * HP + k];
      // c_{t-1}: next slot of this block, or slot 0 of the next block's buffer; 0 before the sequence start
      const float cp = t == 0 ? 0.f : (sl + 1 < S ? row[ROW + 4 * HP + k] : inb[((buf ^ 1) * S) * ROW + 4 * HP + k]);
      xv = xin[buf * S + sl];
      const float ig = quad_bcast<0>(av), gg = quad_bcast<2>(av), og = quad_bcast<3>(av);
      fgv = quad_bcast<1>(av);
      const float tc = tanhf_(ct);
      A = og * (1.f - tc * tc);
      const float up = q == 0 ? gg : (q == 1 ? cp : (q == 2 ? ig : tc));
      const float der = q == 2 ? 1.f - av * av : av * (1.f - av);
      Bq = live ? up * der : 0.f;
    };
    float A, Bq, fgv, xt;
    coeffs(0, A, Bq, fgv, xt);
#pragma unroll 1
    for (int sl = 0; sl < ns; ++sl) {
      const float dct = fmaf(dh, A, dc);
      const float dp = (q == 3 ? dh : dct) * Bq;
      dc = dct * fgv;
      float* dcur = outb + (buf * S + sl) * OROW;
      dcur[q * GP + k] = dp;
      dwi = fmaf(dp, xt, dwi);
      dbv += dp;
      float nA = 0.f, nB = 0.f, nf = 0.f, nx = 0.f;
      if (sl + 1 < ns) coeffs(sl + 1, nA, nB, nf, nx);      // (reads only the staged history: independent of this step)
      __syncthreads();
      float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < PK; i += 4) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dcur[pg * GP + pj0 + i]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int u = 0; u < 4; ++u) p[u] = fmaf(wt[u][i + e], d[e], p[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = row16_sum(p[u]);  // dh_{t-1}[4G + u], in all 16 lanes
      dh = (q16 & 2) ? ((q16 & 1) ? p[3] : p[2]) : ((q16 & 1) ? p[1] : p[0]);   // this lane's unit: k = 4G + l / 4
      A = nA, Bq = nB, fgv = nf, xt = nx;
    }
  }
  __syncthreads();
  {                                                   // the last block's gradients
    const int j = nblk - 1, buf = j & 1, t0 = T - 1 - S * j, ns = min(S, t0 + 1);
    for (int e = tid; e < ns * 4 * HP; e += NT) {
      const int sl = e / (4 * HP), r = e % (4 * HP), g = r / HP, u = r % HP;
      if (u < H) dpb[(size_t)(t0 - sl) * 4 * H + g * H + u] = outb[(buf * S + sl) * OROW + g * GP + u];
    }
  }
  if (live) {
    dwih_p[(size_t)b * 4 * H + q * H + k] = dwi;
    db_p[(size_t)b * 4 * H + q * H + k] = dbv;
  }
}

// dW_hh partial of one chunk of the (sample, step) axis:  part[chunk][r][c] = sum_{(b,t) in chunk} dpre[b][t][r] * h[b][t-1][c],
// h[b][t-1][c] = o[b][t-1][c] * tanh(cell[b][t-1][c]) (0 for t = 0).  Block = 64 rows x all (<= 128) columns for one chunk;
// 256 threads, thread (ty, tx) = 4 rows x CPT columns in registers; 16 (sample, step) pairs per LDS stage.
constexpr int LSTM_DW_CHUNK = 512, LSTM_DW_KS = 16;
template <int CPT>                                     // columns per thread: covers H <= 16 * CPT
__global__ __launch_bounds__(256) void lstm_dwhh_kernel(const float* __restrict__ dpre_all, const float* __restrict__ gates,
                                                        const float* __restrict__ cells, float* __restrict__ part, int BT, int T, int H) {
  __shared__ __attribute__((aligned(16))) float a[LSTM_DW_KS][64], hb[LSTM_DW_KS][16 * CPT];
  const int r0 = blockIdx.x * 64, chunk = blockIdx.y;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // rows r0 + 4*ty + i, columns tx + 16*j
  const int k0 = chunk * LSTM_DW_CHUNK, k1 = min(BT, k0 + LSTM_DW_CHUNK);
  float acc[4][CPT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[i][j] = 0.f;
  for (int kb = k0; kb < k1; kb += LSTM_DW_KS) {
    for (int e = threadIdx.x; e < LSTM_DW_KS * 64; e += 256) {
      const int kk = e >> 6, r = e & 63;
      a[kk][r] = (kb + kk < k1 && r0 + r < 4 * H) ? dpre_all[(size_t)(kb + kk) * 4 * H + r0 + r] : 0.f;
    }
    for (int e = threadIdx.x; e < LSTM_DW_KS * 16 * CPT; e += 256) {
      const int kk = e / (16 * CPT), c = e % (16 * CPT);
      const int g = kb + kk;
      float hv = 0.f;
      if (g < k1 && c < H && g % T != 0) hv = gates[(size_t)(g - 1) * 4 * H + 3 * H + c] * tanhf_(cells[(size_t)(g - 1) * H + c]);
      hb[kk][c] = hv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < LSTM_DW_KS; ++kk) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&a[kk][4 * ty]);
      float hv[CPT];
#pragma unroll
      for (int j = 0; j < CPT; ++j) hv[j] = hb[kk][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[i][j] = fmaf(av[i], hv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int r = r0 + 4 * ty + i, c = tx + 16 * j;
      if (r < 4 * H && c < H) part[((size_t)chunk * 4 * H + r) * H + c] = acc[i][j];
    }
}

// out[i] = sum over b of part[b][i], b ascending (fixed order)
__global__ void sum_over_batch_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ out2, int B, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += part[(size_t)b * n + i];
  out[i] = s;
  if (out2 != nullptr) out2[i] = s;
}

}  // namespace mau

using namespace mau;

extern "C" {

int mau_lstm_max_hidden(void) { return 128; }

int mau_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* h_last,
                 float* gates, float* cells, int B, int T, int H, mau_stream_t stream) {
  MAU_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && h_last && B > 0 && T > 0 && H > 0, "lstm_fwd: bad arguments");
  MAU_REQUIRE(H <= 128, "lstm_fwd: hidden size %d > 128 (one gate row per thread, 4*H threads per sample)", H);
  MAU_REQUIRE((gates == nullptr) == (cells == nullptr), "lstm_fwd: gates and cells come together");
  hipStream_t st = (hipStream_t)stream;
  if (H <= 32) MAU_LAUNCH(lstm_fwd_kernel<32>, dim3(B), dim3(128), 0, st, x, w_ih, w_hh, b_ih, b_hh, h_last, gates, cells, T, H);
  else if (H <= 64) MAU_LAUNCH(lstm_fwd_kernel<64>, dim3(B), dim3(256), 0, st, x, w_ih, w_hh, b_ih, b_hh, h_last, gates, cells, T, H);
  else if (H <= 96) MAU_LAUNCH(lstm_fwd_kernel<96>, dim3(B), dim3(384), 0, st, x, w_ih, w_hh, b_ih, b_hh, h_last, gates, cells, T, H);
  else MAU_LAUNCH(lstm_fwd_kernel<128>, dim3(B), dim3(512), 0, st, x, w_ih, w_hh, b_ih, b_hh, h_last, gates, cells, T, H);
  return check_launch("lstm_fwd_kernel");
}

static inline int lstm_dw_chunks(int B, int T) { return ceil_div((int64_t)B * T, LSTM_DW_CHUNK); }

size_t mau_lstm_bwd_ws_elems(int B, int T, int H) {
  return (size_t)B * T * 4 * H + (size_t)B * 8 * H + (size_t)lstm_dw_chunks(B, T) * 4 * H * H;
}

int mau_lstm_bwd(const float* x, const float* w_hh, const float* gates, const float* cells, const float* dh_last, float* dw_ih,
                 float* dw_hh, float* db_ih, float* db_hh, float* ws, int B, int T, int H, mau_stream_t stream) {
  MAU_REQUIRE(x && w_hh && gates && cells && dh_last && dw_ih && dw_hh && db_ih && db_hh && ws && B > 0 && T > 0 && H > 0, "lstm_bwd: bad arguments");
  MAU_REQUIRE(H <= 128, "lstm_bwd: hidden size %d > 128", H);
  hipStream_t st = (hipStream_t)stream;
  float* dpre_all = ws;
  float* dwih_p = ws + (size_t)B * T * 4 * H;
  float* db_p = dwih_p + (size_t)B * 4 * H;
  float* dwhh_p = db_p + (size_t)B * 4 * H;
  auto bwd_lds = [](int HP) { return (size_t)(2 * LSTM_BS * 5 * HP + 2 * LSTM_BS + 2 * LSTM_BS * 4 * (HP + 8)) * sizeof(float); };
#define MAU_LSTM_BWD(HP_)                                                                                                       \
  do {                                                                                                                          \
    MAU_LDS_ATTR(bwd_lds(HP_), &lstm_bwd_kernel<HP_>);                                                            \
    MAU_LAUNCH(lstm_bwd_kernel<HP_>, dim3(B), dim3(4 * HP_), bwd_lds(HP_), st, x, w_hh, gates, cells, dh_last, dpre_all, dwih_p,  \
               db_p, T, H);                                                                                                     \
  } while (0)
  if (H <= 32) MAU_LSTM_BWD(32);
  else if (H <= 64) MAU_LSTM_BWD(64);
  else if (H <= 96) MAU_LSTM_BWD(96);
  else MAU_LSTM_BWD(128);
#undef MAU_LSTM_BWD
  const int chunks = lstm_dw_chunks(B, T);
  if (H <= 32) MAU_LAUNCH(lstm_dwhh_kernel<2>, dim3(ceil_div(4 * H, 64), chunks), dim3(256), 0, st, (const float*)dpre_all, gates, cells, dwhh_p, B * T, T, H);
  else if (H <= 96) MAU_LAUNCH(lstm_dwhh_kernel<6>, dim3(ceil_div(4 * H, 64), chunks), dim3(256), 0, st, (const float*)dpre_all, gates, cells, dwhh_p, B * T, T, H);
  else MAU_LAUNCH(lstm_dwhh_kernel<8>, dim3(ceil_div(4 * H, 64), chunks), dim3(256), 0, st, (const float*)dpre_all, gates, cells, dwhh_p, B * T, T, H);
  float* const none = nullptr;
  MAU_LAUNCH(sum_over_batch_kernel, dim3(ceil_div(4 * H * H, 256)), dim3(256), 0, st, (const float*)dwhh_p, dw_hh, none, chunks, 4 * H * H);
  MAU_LAUNCH(sum_over_batch_kernel, dim3(ceil_div(4 * H, 256)), dim3(256), 0, st, (const float*)dwih_p, dw_ih, none, B, 4 * H);
  MAU_LAUNCH(sum_over_batch_kernel, dim3(ceil_div(4 * H, 256)), dim3(256), 0, st, (const float*)db_p, db_ih, db_hh, B, 4 * H);
  return check_launch("lstm_bwd_kernel");
}

}  // extern "C"
