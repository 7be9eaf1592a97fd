// embfold.hip -- the broadcast embedding of a convolution as a rank-one term.
//
// The reference concatenates a spatially CONSTANT E-channel map (the per-image embedding broadcast over the pixels,
// src/model.py:248-259 / :111-121) behind the tensors a 3x3 convolution reads: E of its input channels (128 in every decoder node of
// the U-Net++) carry one value per image.  Their contribution to an output pixel is
//     sum over the taps that lie inside the image of  T[n][co][tap],     T[n][co][tap] = sum_e W[co][Ct + e][tap] * emb[n][e]
// which is exactly what the SAME convolution computes over Ep = roundup(N, 16) "indicator" channels (channel i is 1 in image i, 0
// elsewhere -- the zero padding of the convolution takes care of the borders) with the weights T in place of W[:, Ct:, :].  So:
//     W_eff = [ W[:, :Ct, :] | T ]  (Cout x (Ct + Ep) x 9),   embedding := the N x Ep identity matrix,
// and the loader, the multiply loop, the data gradient and the weight gradient all run unchanged on Ct + Ep instead of Ct + E
// channels (U-Net++ node x^{0,1} at B=16: 208 instead of 320).  The backward of the fold maps dW_eff back:
//     dW[:, :Ct] = dW_eff[:, :Ct],   dW[co][Ct + e][tap] = sum_i dW_eff[co][Ct + i][tap] * emb[i][e],
//     demb[i][e] = sum_{co, tap} dW_eff[co][Ct + i][tap] * W[co][Ct + e][tap].
// All in fp32, fixed summation order (deterministic).  functional.EmbFold is the autograd node around these two entry points.
#include "mau_common.h"

namespace mau {

// One workgroup per output channel co: W[co][Ct:][.] (E x 9 floats, contiguous) and emb (N x E) are staged in LDS, thread (i, tap)
// then sums its 128 products from LDS -- as one thread per element reading global memory the sum was a chain of 128 dependent load
// round trips (60 us per layer, more than the fold saved below the top resolution).  The tensor part of the row is a coalesced copy.
__global__ __launch_bounds__(256) void emb_fold_fwd_kernel(const float* __restrict__ w, const float* __restrict__ emb, float* __restrict__ weff,
                                                           int Ct, int E, int N, int Ep) {
  extern __shared__ float lds[];
  float* we = lds;                 // [E][9]
  float* se = lds + E * 9;         // [N][E]
  const int Ce = Ct + Ep, Cf = Ct + E, co = blockIdx.x;
  const float* wrow = w + (size_t)co * Cf * 9;
  float* orow = weff + (size_t)co * Ce * 9;
  for (int t = threadIdx.x; t < E * 9; t += 256) we[t] = wrow[(size_t)Ct * 9 + t];
  for (int t = threadIdx.x; t < N * E; t += 256) se[t] = emb[t];
  for (int t = threadIdx.x; t < Ct * 9; t += 256) orow[t] = wrow[t];
  __syncthreads();
  for (int o = threadIdx.x; o < Ep * 9; o += 256) {
    const int i = o / 9, tap = o - i * 9;
    float v = 0.f;
    if (i < N)
      for (int k = 0; k < E; ++k) v = fmaf(we[k * 9 + tap], se[i * E + k], v);
    orow[(size_t)Ct * 9 + o] = v;
  }
}

// One workgroup per output channel: dW_eff[co][Ct:][.] (Ep x 9) and emb in LDS; dW[co][Ct + e][tap] = sum_i dT[i][tap] * emb[i][e]
__global__ __launch_bounds__(256) void emb_fold_bwd_w_kernel(const float* __restrict__ emb, const float* __restrict__ dweff, float* __restrict__ dw,
                                                             int Ct, int E, int N, int Ep) {
  extern __shared__ float lds[];
  float* dt = lds;                 // [N][9]
  float* se = lds + N * 9;         // [N][E]
  const int Ce = Ct + Ep, Cf = Ct + E, co = blockIdx.x;
  const float* grow = dweff + (size_t)co * Ce * 9;
  float* orow = dw + (size_t)co * Cf * 9;
  for (int t = threadIdx.x; t < N * 9; t += 256) dt[t] = grow[(size_t)Ct * 9 + t];
  for (int t = threadIdx.x; t < N * E; t += 256) se[t] = emb[t];
  for (int t = threadIdx.x; t < Ct * 9; t += 256) orow[t] = grow[t];
  __syncthreads();
  for (int o = threadIdx.x; o < E * 9; o += 256) {
    const int e = o / 9, tap = o - e * 9;
    float v = 0.f;
    for (int i = 0; i < N; ++i) v = fmaf(dt[i * 9 + tap], se[i * E + e], v);
    orow[(size_t)Ct * 9 + o] = v;
  }
}

// demb[i][e], level 1: one workgroup per (64 embedding channels, image i, chunk z of the output channels); its 4 waves split the
// chunk, fixed-order join.  Level 2 adds the chunks in order.  (One workgroup per (i, 64 e) alone -- 32 workgroups walking up to
// 1024 output channels each -- took 0.1-0.5 ms per layer: a latency chain on an empty chip.)
constexpr int FOLD_CO_CHUNK = 32;
__global__ __launch_bounds__(256) void emb_fold_bwd_e_kernel(const float* __restrict__ w, const float* __restrict__ dweff, float* __restrict__ part,
                                                             int Cout, int Ct, int E, int Ep, int N) {
  __shared__ float red[4][64];
  const int Ce = Ct + Ep, Cf = Ct + E;
  const int i = blockIdx.y, z = blockIdx.z, el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int co1 = min(Cout, (z + 1) * FOLD_CO_CHUNK);
  float s = 0.f;
  if (e < E) {
    for (int co = z * FOLD_CO_CHUNK + q; co < co1; co += 4) {
      const float* g = dweff + ((size_t)co * Ce + Ct + i) * 9;
      const float* ww = w + ((size_t)co * Cf + Ct + e) * 9;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) s = fmaf(g[tap], ww[tap], s);
    }
  }
  red[q][el] = s;
  __syncthreads();
  if (q == 0 && e < E) part[((size_t)z * N + i) * E + e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

__global__ void emb_fold_bwd_e_final_kernel(const float* __restrict__ part, float* __restrict__ demb, int n, int chunks) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  float s = 0.f;
  for (int z = 0; z < chunks; ++z) s += part[(size_t)z * n + idx];
  demb[idx] = s;
}

}  // namespace mau

using namespace mau;

extern "C" {

int mau_emb_fold_fwd(const float* w, const float* emb, float* weff, int Cout, int Ct, int E, int N, int Ep, mau_stream_t stream) {
  MAU_REQUIRE(w && emb && weff && Cout > 0 && Ct >= 0 && E > 0 && N > 0 && Ep >= N, "emb_fold_fwd: bad arguments (Ep >= N)");
  const size_t lds = ((size_t)E * 9 + (size_t)N * E) * sizeof(float);
  MAU_REQUIRE(lds <= 64 * 1024, "emb_fold_fwd: E * 9 + N * E floats must fit 64 KB of LDS");
  MAU_LAUNCH(emb_fold_fwd_kernel, dim3(Cout), dim3(256), lds, (hipStream_t)stream, w, emb, weff, Ct, E, N, Ep);
  return check_launch("emb_fold_fwd_kernel");
}

size_t mau_emb_fold_ws_elems(int Cout, int N, int E) { return (size_t)ceil_div(Cout, FOLD_CO_CHUNK) * N * E; }

int mau_emb_fold_bwd(const float* w, const float* emb, const float* dweff, float* dw, float* demb, float* ws, int Cout, int Ct, int E,
                     int N, int Ep, mau_stream_t stream) {
  MAU_REQUIRE(w && emb && dweff && (dw || demb) && (!demb || ws) && Cout > 0 && Ct >= 0 && E > 0 && N > 0 && Ep >= N,
              "emb_fold_bwd: bad arguments (Ep >= N; demb needs the workspace of mau_emb_fold_ws_elems floats)");
  hipStream_t st = (hipStream_t)stream;
  if (dw) {
    const size_t lds = ((size_t)N * 9 + (size_t)N * E) * sizeof(float);
    MAU_REQUIRE(lds <= 64 * 1024, "emb_fold_bwd: N * 9 + N * E floats must fit 64 KB of LDS");
    MAU_LAUNCH(emb_fold_bwd_w_kernel, dim3(Cout), dim3(256), lds, st, emb, dweff, dw, Ct, E, N, Ep);
  }
  if (demb) {
    const int chunks = ceil_div(Cout, FOLD_CO_CHUNK);
    MAU_REQUIRE(N <= 65535 && chunks <= 65535, "emb_fold_bwd: batch / channel chunks must fit a grid dimension");
    MAU_LAUNCH(emb_fold_bwd_e_kernel, dim3(ceil_div(E, 64), N, chunks), dim3(256), 0, st, w, dweff, ws, Cout, Ct, E, Ep, N);
    MAU_LAUNCH(emb_fold_bwd_e_final_kernel, dim3(ceil_div(N * E, 256)), dim3(256), 0, st, (const float*)ws, demb, N * E, chunks);
  }
  return check_launch("emb_fold_bwd_kernel");
}

}  // extern "C"
