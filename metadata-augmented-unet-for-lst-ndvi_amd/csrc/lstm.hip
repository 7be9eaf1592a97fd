// lstm.hip -- TemporalEncoder's recurrence (reference src/model.py:23-34: nn.LSTM(input_size=1, hidden_size=H,
// batch_first=True), last hidden state) as ONE persistent launch per direction.
//
// The reference's sequence is 828 monthly temperatures (conf/config.yaml:20) and the hidden size 96: 828 DEPENDENT steps of
// a 384x96 matrix-vector product per sample -- pure latency, 2.4 MFLOP per step.  A library LSTM launches kernels per
// step (measured: 26-30 ms forward + backward at B=32, more than the whole U-Net step).  Here:
//   forward  : one workgroup per sample, 4*HP threads; thread 4u + q owns gate row q*H + u of W_hh in registers for the
//              whole sequence; h_t in LDS (broadcast reads, double buffered); the four gates of a unit sit in one lane
//              quad and meet through DPP broadcasts: ONE barrier per step; gate order i, f, g, o (torch).
//   backward : one workgroup per sample, thread 4k + q owns column k of gate q's block of W_hh; per step each lane turns
//              the quad-redundant (dh, dc) into the pre-activation gradient of its gate (saved activations are read one
//              step ahead), publishes it in LDS (double buffered: ONE barrier per step), multiplies its column, and
//              the unit's four partial products are added inside the quad.  The pre-activation gradients are also
//              stored: the weight gradient dW_hh = sum_t dpre_t (x) h_{t-1} is NOT part of the recurrence -- it is
//              one [4H x B*T] x [B*T x H] product computed afterwards by a tiled kernel (h_{t-1} = o_{t-1} *
//              tanh(c_{t-1}) recomputed from the saved gates, no h tensor).  All partial sums are added in a fixed
//              order by a second stage: bitwise reproducible, no atomics.
//   measured : B=32, T=828, H=96 (scripts/lstm_bench.py): see DESIGN.md; the step is a latency chain (LDS broadcast of
//              h, 48 packed FMAs, exp + rcp, barrier), not a throughput problem -- two gate rows per thread (half the
//              LDS reads, twice the FMAs per thread) measured 23 % SLOWER.
// Arithmetic is fp32 throughout; dot products on packed fp32 FMAs, gate nonlinearities on the hardware exp / rcp.
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
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }
#endif

// value of lane (quad base + J) of every quad, for all four lanes of the quad (DPP quad_perm: no LDS, no barrier)
template <int J>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), J * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v) {
  return (quad_bcast<0>(v) + quad_bcast<1>(v)) + (quad_bcast<2>(v) + quad_bcast<3>(v));
}

// x (B,T); w_ih (4H) [input_size 1]; w_hh (4H,H); b_ih, b_hh (4H); h_last (B,H);
// gates (B,T,4H) post-activation i,f,g,o and cells (B,T,H) are written when non-null (saved for backward).
// Thread 4u + q owns gate row q*H + u: the four gates of a unit sit in one lane QUAD, meet through DPP broadcasts (no LDS
// round trip, no second barrier) and every lane of the quad carries the unit's cell state redundantly.  One barrier per
// step (h_t is double-buffered in LDS).
template <int HP>
__global__ __launch_bounds__(4 * HP) void lstm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                         const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                         const float* __restrict__ b_hh, float* __restrict__ h_last,
                                                         float* __restrict__ gates, float* __restrict__ cells, int T, int H) {
  __shared__ __attribute__((aligned(16))) float hs[2][HP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int u = tid >> 2, q = tid & 3;                // unit u, gate q (i, f, g, o)
  const bool live = u < H;
  const int row = q * H + u;
  f32x2 w[HP / 2];                                     // (pairs: the dot product runs on v_pk_fma_f32)
#pragma unroll
  for (int k = 0; k < HP; ++k) w[k >> 1][k & 1] = (live && k < H) ? w_hh[(size_t)row * H + k] : 0.f;
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
    const float* hcur = hs[t & 1];
    f32x2 a0 = {fmaf(wi, xt, bias), 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < HP; k += 4) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(&hcur[k]);
      a0 = __builtin_elementwise_fma(w[k >> 1], hv.xy, a0);
      a1 = __builtin_elementwise_fma(w[(k >> 1) + 1], hv.zw, a1);
    }
    const float pre = (a0[0] + a0[1]) + (a1[0] + a1[1]);
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
// -- both carried redundantly by the quad -- into the pre-activation gradient of ITS gate, publishes it in LDS (double
// buffered: one barrier per step), multiplies its column with the gate's gradient vector, and the four partial products
// of unit k are added inside the quad (DPP), which leaves dh_{t-1}[k] in the registers of the lanes that need it next.
template <int HP>
__global__ __launch_bounds__(4 * HP) void lstm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w_hh,
                                                         const float* __restrict__ gates, const float* __restrict__ cells,
                                                         const float* __restrict__ dh_last, float* __restrict__ dpre_all,
                                                         float* __restrict__ dwih_p, float* __restrict__ db_p, int T, int H) {
  __shared__ __attribute__((aligned(16))) float dpre[2][4 * HP];  // pre-activation gate gradients of a step, [gate][unit]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int k = tid >> 2, q = tid & 3;
  const bool live = k < H;
  f32x2 wt[HP / 2];
#pragma unroll
  for (int j = 0; j < HP; ++j) wt[j >> 1][j & 1] = (live && j < H) ? w_hh[(size_t)(q * H + j) * H + k] : 0.f;
  float dc = 0.f, dwi = 0.f, dbv = 0.f;
  float dh = live ? dh_last[(size_t)b * H + k] : 0.f;
  for (int i = tid; i < 2 * 4 * HP; i += 4 * HP) (&dpre[0][0])[i] = 0.f;
  __syncthreads();
  const float* xb = x + (size_t)b * T;
  const float* gb = gates + (size_t)b * T * 4 * H;
  const float* cb = cells + (size_t)b * T * H;
  // Saved activations are read ONE STEP AHEAD into registers (a dependent global load per step would cost its full
  // latency 828 times): this lane's gate of step t (the quad holds i, f, g, o), c_t and c_{t-1}.
  float av = 0.f, ct = 0.f, cp = 0.f, xt = xb[T - 1];
  if (live) {
    av = gb[(size_t)(T - 1) * 4 * H + q * H + k];
    ct = cb[(size_t)(T - 1) * H + k];
    if (T > 1) cp = cb[(size_t)(T - 2) * H + k];
  }
  float* dpb = dpre_all + (size_t)b * T * 4 * H;
  for (int t = T - 1; t >= 0; --t) {
    // ---- prefetch for step t-1 ----
    float nav = 0.f, ncp = 0.f, nxt = 0.f;
    if (t > 0) {
      nxt = xb[t - 1];
      if (live) {
        nav = gb[(size_t)(t - 1) * 4 * H + q * H + k];
        if (t > 1) ncp = cb[(size_t)(t - 2) * H + k];
      }
    }
    const float ig = quad_bcast<0>(av), fg = quad_bcast<1>(av), gg = quad_bcast<2>(av), og = quad_bcast<3>(av);
    const float tc = tanhf_(ct);
    const float dct = fmaf(dh * og, 1.f - tc * tc, dc);
    // pre-activation gradient of this lane's gate: i, f: d * s(1-s); g: d * (1-g^2); o: d * s(1-s)
    const float up = q == 0 ? dct * gg : (q == 1 ? dct * cp : (q == 2 ? dct * ig : dh * tc));
    const float der = q == 2 ? 1.f - av * av : av * (1.f - av);
    const float dp = live ? up * der : 0.f;
    dc = dct * fg;
    float* dcur = dpre[t & 1];
    dcur[q * HP + k] = dp;
    if (live) dpb[(size_t)t * 4 * H + q * H + k] = dp;
    dwi = fmaf(dp, xt, dwi);
    dbv += dp;
    __syncthreads();
    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HP; j += 4) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(&dcur[q * HP + j]);
      p0 = __builtin_elementwise_fma(wt[j >> 1], d.xy, p0);
      p1 = __builtin_elementwise_fma(wt[(j >> 1) + 1], d.zw, p1);
    }
    dh = quad_sum((p0[0] + p0[1]) + (p1[0] + p1[1]));       // dh_{t-1}[k], in every lane of the quad
    // ---- rotate: step t-1's cell state is this step's previous cell state ----
    ct = cp;
    av = nav, cp = ncp, xt = nxt;
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
  if (H <= 32) MAU_LAUNCH(lstm_bwd_kernel<32>, dim3(B), dim3(128), 0, st, x, w_hh, gates, cells, dh_last, dpre_all, dwih_p, db_p, T, H);
  else if (H <= 64) MAU_LAUNCH(lstm_bwd_kernel<64>, dim3(B), dim3(256), 0, st, x, w_hh, gates, cells, dh_last, dpre_all, dwih_p, db_p, T, H);
  else if (H <= 96) MAU_LAUNCH(lstm_bwd_kernel<96>, dim3(B), dim3(384), 0, st, x, w_hh, gates, cells, dh_last, dpre_all, dwih_p, db_p, T, H);
  else MAU_LAUNCH(lstm_bwd_kernel<128>, dim3(B), dim3(512), 0, st, x, w_hh, gates, cells, dh_last, dpre_all, dwih_p, db_p, T, H);
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
