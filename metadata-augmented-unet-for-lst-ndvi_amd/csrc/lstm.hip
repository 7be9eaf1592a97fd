// lstm.hip -- TemporalEncoder's recurrence (reference src/model.py:23-34: nn.LSTM(input_size=1, hidden_size=H,
// batch_first=True), last hidden state) as ONE persistent launch per direction.
//
// The reference's sequence is 828 monthly temperatures (conf/config.yaml:20) and the hidden size 96: 828 DEPENDENT steps of
// a 384x96 matrix-vector product per sample -- pure latency, 2.4 MFLOP per step.  A library LSTM launches kernels per
// step (measured: 26-30 ms forward + backward at B=32, more than the whole U-Net step).  Here:
//   forward  : one workgroup per sample, 4*HP threads; thread j owns gate row j: W_hh[j][0..H) lives in its registers for
//              the whole sequence, h_t in LDS (broadcast reads), two barriers per step; gate order i, f, g, o (torch).
//   backward : one workgroup per sample, thread (q, k) owns column k of gate q: W_hh[q*H + j][k] (j = 0..H) stays in
//              its registers over all 828 steps; per step the H "unit" threads turn (dh, dc) into the four
//              pre-activation gradients (saved activations are read one step ahead), everybody multiplies, the four
//              partial products per unit are summed through LDS (three barriers per step).  The pre-activation
//              gradients are also stored: the weight gradient dW_hh = sum_t dpre_t (x) h_{t-1} is NOT part of the
//              recurrence -- it is one [4H x B*T] x [B*T x H] product computed afterwards by a tiled kernel
//              (h_{t-1} = o_{t-1} * tanh(c_{t-1}) recomputed from the saved gates, no h tensor), which keeps 96 more
//              accumulators out of the sequential kernel's registers.  All partial sums are added in a fixed order
//              by a second stage: bitwise reproducible, no atomics.
// Arithmetic is fp32 throughout, with expf / tanhf as torch's CPU path uses them.
#include "mau_common.h"

namespace mau {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// x (B,T); w_ih (4H) [input_size 1]; w_hh (4H,H); b_ih, b_hh (4H); h_last (B,H);
// gates (B,T,4H) post-activation i,f,g,o and cells (B,T,H) are written when non-null (saved for backward).
template <int HP>
__global__ __launch_bounds__(4 * HP) void lstm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                         const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                         const float* __restrict__ b_hh, float* __restrict__ h_last,
                                                         float* __restrict__ gates, float* __restrict__ cells, int T, int H) {
  __shared__ __attribute__((aligned(16))) float hs[HP];
  __shared__ float act[4 * HP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int q = tid / HP, u = tid % HP;               // gate q, unit u
  const bool live = u < H;
  const int row = q * H + u;
  float w[HP];
#pragma unroll
  for (int k = 0; k < HP; ++k) w[k] = (live && k < H) ? w_hh[(size_t)row * H + k] : 0.f;
  const float wi = live ? w_ih[row] : 0.f;
  const float bias = live ? b_ih[row] + b_hh[row] : 0.f;
  if (tid < HP) hs[tid] = 0.f;
  float c = 0.f;
  const float* xb = x + (size_t)b * T;
  __syncthreads();
  float xt = xb[0];
  for (int t = 0; t < T; ++t) {
    const float xn = t + 1 < T ? xb[t + 1] : 0.f;      // next input in flight while this step multiplies
    float a0 = fmaf(wi, xt, bias), a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < HP; k += 4) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(&hs[k]);
      a0 = fmaf(w[k], hv[0], a0);
      a1 = fmaf(w[k + 1], hv[1], a1);
      a2 = fmaf(w[k + 2], hv[2], a2);
      a3 = fmaf(w[k + 3], hv[3], a3);
    }
    const float pre = (a0 + a1) + (a2 + a3);
    const float av = q == 2 ? tanhf(pre) : sigmoidf_(pre);
    act[tid] = av;
    if (gates != nullptr && live) gates[((size_t)b * T + t) * 4 * H + row] = av;
    __syncthreads();
    if (tid < H) {
      const float ig = act[tid], fg = act[HP + tid], gg = act[2 * HP + tid], og = act[3 * HP + tid];
      c = fmaf(fg, c, ig * gg);
      hs[tid] = og * tanhf(c);
      if (cells != nullptr) cells[((size_t)b * T + t) * H + tid] = c;
    }
    __syncthreads();
    xt = xn;
  }
  if (tid < H) h_last[(size_t)b * H + tid] = hs[tid];
}

// dh_last (B,H) -> dpre_all (B,T,4H) pre-activation gate gradients + per-sample partials dwih_p (B,4H), db_p (B,4H)
template <int HP>
__global__ __launch_bounds__(4 * HP) void lstm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w_hh,
                                                         const float* __restrict__ gates, const float* __restrict__ cells,
                                                         const float* __restrict__ dh_last, float* __restrict__ dpre_all,
                                                         float* __restrict__ dwih_p, float* __restrict__ db_p, int T, int H) {
  __shared__ __attribute__((aligned(16))) float dpre[4 * HP];     // pre-activation gate gradients of the step, [gate][unit]
  __shared__ float part[4 * HP];                                  // partial W^T dpre per (gate, unit)
  __shared__ float dhs[HP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int q = tid / HP, k = tid % HP;
  const bool live = k < H;
  // column k of gate q's block of W_hh
  float wt[HP];
#pragma unroll
  for (int j = 0; j < HP; ++j) wt[j] = (live && j < H) ? w_hh[(size_t)(q * H + j) * H + k] : 0.f;
  float dc = 0.f;                                      // unit threads (tid < H): running d loss / d c_t
  float dwi[4] = {0.f, 0.f, 0.f, 0.f}, dbv[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < HP) dhs[tid] = tid < H ? dh_last[(size_t)b * H + tid] : 0.f;
  for (int i = tid; i < 4 * HP; i += 4 * HP) dpre[i] = 0.f;
  __syncthreads();
  const float* xb = x + (size_t)b * T;
  const float* gb = gates + (size_t)b * T * 4 * H;
  const float* cb = cells + (size_t)b * T * H;
  const bool unit = tid < H;
  // Saved activations are read ONE STEP AHEAD into registers (a dependent global load per step would cost its full
  // latency 828 times): (ig, fg, gg, og, ct) of step t and c_{t-1} for the unit threads.
  float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, ct = 0.f, xt = 0.f, cp = 0.f;
  if (unit) {
    const float* g = gb + (size_t)(T - 1) * 4 * H;
    ig = g[tid], fg = g[H + tid], gg = g[2 * H + tid], og = g[3 * H + tid];
    ct = cb[(size_t)(T - 1) * H + tid];
    xt = xb[T - 1];
    if (T > 1) cp = cb[(size_t)(T - 2) * H + tid];
  }
  float* dpb = dpre_all + (size_t)b * T * 4 * H;
  for (int t = T - 1; t >= 0; --t) {
    // ---- prefetch for step t-1 ----
    float nig = 0.f, nfg = 0.f, ngg = 0.f, nog = 0.f, nxt = 0.f, ncp = 0.f;
    if (unit && t > 0) {
      const float* g = gb + (size_t)(t - 1) * 4 * H;
      nig = g[tid], nfg = g[H + tid], ngg = g[2 * H + tid], nog = g[3 * H + tid];
      nxt = xb[t - 1];
      if (t > 1) ncp = cb[(size_t)(t - 2) * H + tid];
    }
    if (unit) {
      const float cprev = t > 0 ? cp : 0.f;            // (unit threads: k == tid)
      const float tc = tanhf(ct);
      const float dh = dhs[tid];
      const float dct = fmaf(dh * og, 1.f - tc * tc, dc);
      const float pi = dct * gg * ig * (1.f - ig);
      const float pf = dct * cprev * fg * (1.f - fg);
      const float pg = dct * ig * (1.f - gg * gg);
      const float po = dh * tc * og * (1.f - og);
      dc = dct * fg;
      dpre[tid] = pi;
      dpre[HP + tid] = pf;
      dpre[2 * HP + tid] = pg;
      dpre[3 * HP + tid] = po;
      float* dp = dpb + (size_t)t * 4 * H;
      dp[tid] = pi, dp[H + tid] = pf, dp[2 * H + tid] = pg, dp[3 * H + tid] = po;
      dwi[0] = fmaf(pi, xt, dwi[0]);
      dwi[1] = fmaf(pf, xt, dwi[1]);
      dwi[2] = fmaf(pg, xt, dwi[2]);
      dwi[3] = fmaf(po, xt, dwi[3]);
      dbv[0] += pi;
      dbv[1] += pf;
      dbv[2] += pg;
      dbv[3] += po;
    }
    __syncthreads();
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
    for (int j = 0; j < HP; j += 4) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(&dpre[q * HP + j]);
      p0 = fmaf(wt[j], d[0], p0);
      p1 = fmaf(wt[j + 1], d[1], p1);
      p2 = fmaf(wt[j + 2], d[2], p2);
      p3 = fmaf(wt[j + 3], d[3], p3);
    }
    part[tid] = (p0 + p1) + (p2 + p3);
    __syncthreads();
    if (tid < HP) dhs[tid] = (part[tid] + part[HP + tid]) + (part[2 * HP + tid] + part[3 * HP + tid]);
    __syncthreads();
    // ---- rotate: step t-1's cell state is this step's previous cell state ----
    ct = cp;
    ig = nig, fg = nfg, gg = ngg, og = nog, xt = nxt, cp = ncp;
  }
  if (tid < H) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      dwih_p[(size_t)b * 4 * H + gq * H + tid] = dwi[gq];
      db_p[(size_t)b * 4 * H + gq * H + tid] = dbv[gq];
    }
  }
}

// dW_hh partial of one chunk of the (sample, step) axis:  part[chunk][r][c] = sum_{(b,t) in chunk} dpre[b][t][r] * h[b][t-1][c],
// h[b][t-1][c] = o[b][t-1][c] * tanh(cell[b][t-1][c]) (0 for t = 0).  Block = 32 rows x 32 columns, 256 threads x 4 outputs.
constexpr int LSTM_DW_CHUNK = 1024;
__global__ __launch_bounds__(256) void lstm_dwhh_kernel(const float* __restrict__ dpre_all, const float* __restrict__ gates,
                                                        const float* __restrict__ cells, float* __restrict__ part, int BT, int T, int H) {
  __shared__ float a[32][33], hb[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, chunk = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // outputs (r0 + ty + 8*i, c0 + tx), i = 0..3
  const int k0 = chunk * LSTM_DW_CHUNK, k1 = min(BT, k0 + LSTM_DW_CHUNK);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kb = k0; kb < k1; kb += 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = kb + ty + 8 * i;                             // (sample, step) index of this tile row
      float av = 0.f, hv = 0.f;
      if (kk < k1) {
        if (r0 + tx < 4 * H) av = dpre_all[(size_t)kk * 4 * H + r0 + tx];
        if (kk % T != 0 && c0 + tx < H) hv = gates[(size_t)(kk - 1) * 4 * H + 3 * H + c0 + tx] * tanhf(cells[(size_t)(kk - 1) * H + c0 + tx]);
      }
      a[ty + 8 * i][tx] = av;
      hb[ty + 8 * i][tx] = hv;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const float hv = hb[kk][tx];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(a[kk][ty + 8 * i], hv, acc[i]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < 4 * H && c < H) part[((size_t)chunk * 4 * H + r) * H + c] = acc[i];
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
  MAU_LAUNCH(lstm_dwhh_kernel, dim3(ceil_div(4 * H, 32), ceil_div(H, 32), chunks), dim3(256), 0, st, (const float*)dpre_all, gates, cells,
             dwhh_p, B * T, T, H);
  float* const none = nullptr;
  MAU_LAUNCH(sum_over_batch_kernel, dim3(ceil_div(4 * H * H, 256)), dim3(256), 0, st, (const float*)dwhh_p, dw_hh, none, chunks, 4 * H * H);
  MAU_LAUNCH(sum_over_batch_kernel, dim3(ceil_div(4 * H, 256)), dim3(256), 0, st, (const float*)dwih_p, dw_ih, none, B, 4 * H);
  MAU_LAUNCH(sum_over_batch_kernel, dim3(ceil_div(4 * H, 256)), dim3(256), 0, st, (const float*)db_p, db_ih, db_hh, B, 4 * H);
  return check_launch("lstm_bwd_kernel");
}

}  // extern "C"
