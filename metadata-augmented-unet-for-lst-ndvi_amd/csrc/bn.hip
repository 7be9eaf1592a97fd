// bn.hip -- train/eval BatchNorm2d + ReLU on NHWC-ld activations (reference src/model.py:13,15,16).
//
// Forward, train mode:  conv epilogue -> per-tile partial sums (slab)  -> reduce_rows (fp64)
//                       -> [RCCL all-reduce when data parallel] -> bn_finalize_train -> bn_relu_apply
// Backward:             bn_relu_bwd_reduce (partials) -> reduce_rows -> [all-reduce] -> bn_relu_bwd_apply
// All reductions are slab + fixed-order second stage: bitwise reproducible, no float atomics.
// The two levels of a slab reduction are ONE launch when the caller passes a ticket buffer: every workgroup of a column
// block publishes its fp64 partial and takes a ticket; the workgroup that draws the last ticket adds the partials in the
// SAME fixed order as the two-launch form (bit-identical), finalizes, and resets the ticket (launch tail: 36 launches fewer
// per train step of the U-Net).  Hand-off protocol: cdna_hip_programming.md Guideline 16 (agent-scope release / acquire).
// All of these kernels are HBM-streaming: 16-byte (bf16) / 32-byte (f32) vectors per lane.
#include <stdlib.h>
#include "mau_common.h"

namespace mau {

// ---- the fixed-order two-accumulator sum of the reductions below, with a group's loads in flight at once ----
//     s0 += v[first], v[first + 8], ...;   s1 += v[first + 4], v[first + 12], ...      (indices < end, ascending)
// is the arithmetic of the plain loop "for (k = first; k + 4 < end; k += 8) { s0 += v[k]; s1 += v[k + 4]; } if (k < end) s0 += v[k];"
// -- same operands, same order, same bits.  As that loop the level was a chain of dependent round trips (each iteration's
// loads were issued after the previous iteration's additions: 16 trips of ~0.65 us through L2 for 128 chunks; the finalize
// launches of the 256^2 / 128^2 layers took 14 us against 5.5 us for the layers with <= 16 chunks).  Here whole groups of 16,
// then of 4 iterations issue their loads back to back and add in order; what is left (< 4 iterations) runs as the plain loop,
// so the short reductions of the deep layers execute exactly what they did before.  (A single 16-wide group with clamped
// indices and predicated additions for every length cost the short ones +3.5 us of instruction issue: 4 waves, one per SIMD.)
template <int G, typename Load>
__device__ __forceinline__ void ordered_group(int& k, int end, Load load, double& s0, double& s1) {
  for (; k + 8 * (G - 1) + 4 < end; k += 8 * G) {          // every index of the group is valid
    double a[G], b[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      a[i] = load(k + 8 * i);
      b[i] = load(k + 8 * i + 4);
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
      s0 += a[i];
      s1 += b[i];
    }
  }
}
template <typename Load>
__device__ __forceinline__ void ordered_pair_sum(int first, int end, Load load, double& s0, double& s1) {
  int k = first;
#ifndef MAU_REDUCE_SERIAL           // A/B (scripts/build_variants.sh): the plain loop alone
  ordered_group<16>(k, end, load, s0, s1);
  ordered_group<4>(k, end, load, s0, s1);
#endif
  for (; k + 4 < end; k += 8) {
    s0 += load(k);
    s1 += load(k + 4);
  }
  if (k < end) s0 += load(k);
}

// ---- column sums of a row-major slab [rows][ldrow] -> out[M], accumulated in fp64 ----
// Two levels, both deterministic: level 1 cuts the rows into gridDim.y chunks and writes fp64
// partials [chunk][M]; level 2 (same kernel, one chunk) adds the partials in fixed order.
// block = 64 columns x 4 row-lanes: each wave reads 64 consecutive columns of one row.
template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const InT* __restrict__ slab, int rows, int M, int ldrow,
                                                          OutT* __restrict__ out, int ldout, float* __restrict__ out32) {
  __shared__ double part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s0 = 0.0, s1 = 0.0;
  if (col < M) {
    if (r0 + rl < r1) ordered_pair_sum(r0 + rl, r1, [&](int r) { return (double)slab[(size_t)r * ldrow + col]; }, s0, s1);
  }
  part[rl][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (rl == 0 && col < M) {
    const double v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    out[(size_t)blockIdx.y * ldout + col] = (OutT)v;
    if (out32 != nullptr) out32[col] = (float)v;         // (final level only) the same sums rounded to fp32
  }
}

constexpr int REDUCE_MAX_CHUNKS = 128;
static int reduce_chunks(int rows) {
  int c = rows / 32;
  if (c > REDUCE_MAX_CHUNKS) c = REDUCE_MAX_CHUNKS;
  return c < 1 ? 1 : c;
}

template <typename OutT>
static int reduce_rows_launch(const float* slab, int rows, int M, int ldrow, OutT* out, double* ws, hipStream_t st,
                              float* out32 = nullptr) {
  const int chunks = reduce_chunks(rows);
  float* const none = nullptr;
  if (chunks == 1 || ws == nullptr) {
    MAU_LAUNCH((reduce_rows_kernel<float, OutT>), dim3(ceil_div(M, 64), 1), dim3(256), 0, st, slab, rows, M, ldrow, out, M, out32);
  } else {
    MAU_LAUNCH((reduce_rows_kernel<float, double>), dim3(ceil_div(M, 64), chunks), dim3(256), 0, st, slab, rows, M, ldrow, ws, M, none);
    MAU_LAUNCH((reduce_rows_kernel<double, OutT>), dim3(ceil_div(M, 64), 1), dim3(256), 0, st, (const double*)ws, chunks, M, M, out, M, out32);
  }
  return check_launch("reduce_rows_kernel");
}

// ---- "the last workgroup of a column block runs the second level" ----
// Every thread of the block has issued its partial stores.  Returns true in exactly one block per ticket: the one that
// arrives last; by then the partials of all n blocks are visible to it.  Producer side: each storing wave drains its
// stores (vmcnt(0)), workgroup barrier, one lane releases at agent scope and takes the ticket.  Consumer side (the last
// arriver): agent-scope acquire (invalidates this CU's L1: a line of the partial buffer may be resident from an earlier
// launch), its completion waited for, barrier, then plain loads.  The ticket is reset for the next launch on the stream.
__device__ __forceinline__ bool last_block_of(unsigned* ticket, unsigned n) {
  __shared__ unsigned s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = t == n - 1 ? 1u : 0u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

// Single-launch form of reduce_rows_launch: grid (column blocks, chunks).  Output column j reads slab column
// j (j < split) or j - split + gap (j >= split): the BatchNorm statistics slab keeps sum and sum of squares in two
// 64-padded halves, the outputs are [C | C] dense.  Arithmetic and order are those of the two-launch form.
template <typename OutT>
__global__ __launch_bounds__(256) void reduce_rows_fused_kernel(const float* __restrict__ slab, int rows, int M, int ldrow, int split,
                                                                int gap, double* __restrict__ part, unsigned* __restrict__ tickets,
                                                                OutT* __restrict__ out, float* __restrict__ out32, double append) {
  __shared__ double ps[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cl;
  const int col = j < split ? j : j - split + gap;
  const int chunks = gridDim.y;
  const int per = (rows + chunks - 1) / chunks;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s0 = 0.0, s1 = 0.0;
  if (j < M) {
    if (r0 + rl < r1) ordered_pair_sum(r0 + rl, r1, [&](int r) { return (double)slab[(size_t)r * ldrow + col]; }, s0, s1);
  }
  ps[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && j < M) part[(size_t)blockIdx.y * M + j] = ps[0][cl] + ps[1][cl] + ps[2][cl] + ps[3][cl];
  if (!last_block_of(tickets + blockIdx.x, (unsigned)chunks)) return;
  // level 2: the chunks' partials in a fixed order (the order of reduce_rows_kernel<double, OutT> on one chunk)
  s0 = s1 = 0.0;
  if (j < M) {
    if (rl < chunks) ordered_pair_sum(rl, chunks, [&](int k) { return part[(size_t)k * M + j]; }, s0, s1);
  }
  ps[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && j < M) {
    const double v = ps[0][cl] + ps[1][cl] + ps[2][cl] + ps[3][cl];
    out[j] = (OutT)v;
    if (out32 != nullptr) out32[j] = (float)v;
  }
  if (append > 0.0 && blockIdx.x == 0 && threadIdx.x == 0) out[M] = (OutT)append;    // the local pixel count travels with the sums
}

template <typename OutT>
static int reduce_rows_fused_launch(const float* slab, int rows, int M, int ldrow, int split, int gap, OutT* out, double* ws,
                                    unsigned* tickets, hipStream_t st, float* out32 = nullptr, double append = 0.0) {
  const int chunks = reduce_chunks(rows);
  MAU_LAUNCH((reduce_rows_fused_kernel<OutT>), dim3(ceil_div(M, 64), chunks), dim3(256), 0, st, slab, rows, M, ldrow, split, gap, ws,
             tickets, out, out32, append);
  return check_launch("reduce_rows_fused_kernel");
}

__global__ void bn_finalize_train_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                         int64_t* nbt, float momentum, float eps, float* __restrict__ scale,
                                         float* __restrict__ shift, float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                         int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  if (count <= 0.0) count = sums[2 * C];              // data parallel: the all-reduced pixel count travels with the sums
  const double mean = sums[c] / count;
  double var = sums[C + c] / count - mean * mean;     // biased variance (normalisation)
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_o[c] = (float)mean;
  invstd_o[c] = invstd;
  if (rmean) {
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
}

// Fused second level of the statistics reduce + finalize (single-GPU training): partial fp64 sums
// [chunks][2*ldp] (column c: sum, column ldp + c: sum of squares) -> the same outputs as
// bn_finalize_train_kernel; chunks added in a fixed order (bitwise reproducible).
// block = 64 channels x 4 chunk-lanes: each lane adds every 4th chunk (two independent chains), LDS joins the four.
__global__ __launch_bounds__(256) void bn_finalize_from_partials_kernel(const double* __restrict__ part, int chunks, int ldp, double count,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* __restrict__ rmean, float* __restrict__ rvar, int64_t* nbt, float momentum,
                                                 float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                 float* __restrict__ mean_o, float* __restrict__ invstd_o, int C) {
  __shared__ double ps[4][64], pq[4][64];
  const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
  if (c < C && kl < chunks) {
    ordered_pair_sum(kl, chunks, [&](int k) { return part[(size_t)k * 2 * ldp + c]; }, s0, s1);
    ordered_pair_sum(kl, chunks, [&](int k) { return part[(size_t)k * 2 * ldp + ldp + c]; }, q0, q1);
  }
  ps[kl][cl] = s0 + s1;
  pq[kl][cl] = q0 + q1;
  __syncthreads();
  if (kl != 0 || c >= C) return;
  const double s = (ps[0][cl] + ps[1][cl]) + (ps[2][cl] + ps[3][cl]);
  const double q = (pq[0][cl] + pq[1][cl]) + (pq[2][cl] + pq[3][cl]);
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_o[c] = (float)mean;
  invstd_o[c] = invstd;
  if (rmean) {
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
}

// Single-launch statistics reduce + finalize (single-GPU training): grid (cpad / 64, chunks); level 1 = the conv epilogue's
// slab [rows][2 * cpad] -> fp64 partials [chunk][2 * cpad]; the last workgroup of a channel block adds the chunks in the
// order of bn_finalize_from_partials_kernel and finalizes its 64 channels (bit-identical to the two-launch form).
__global__ __launch_bounds__(256) void bn_stats_finalize_fused_kernel(const float* __restrict__ slab, int rows, int cpad,
                                                                      double* __restrict__ part, unsigned* __restrict__ tickets,
                                                                      double count, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, float* __restrict__ rmean,
                                                                      float* __restrict__ rvar, int64_t* nbt, float momentum,
                                                                      float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                                      float* __restrict__ mean_o, float* __restrict__ invstd_o, int C) {
  __shared__ double ps[4][64], pq[4][64];
  const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;                          // (c < cpad always: the slab's pad columns are zeros)
  const int chunks = gridDim.y, ld = 2 * cpad;
  const int per = (rows + chunks - 1) / chunks;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
  if (r0 + kl < r1) {
    ordered_pair_sum(r0 + kl, r1, [&](int r) { return (double)slab[(size_t)r * ld + c]; }, s0, s1);
    ordered_pair_sum(r0 + kl, r1, [&](int r) { return (double)slab[(size_t)r * ld + cpad + c]; }, q0, q1);
  }
  ps[kl][cl] = s0 + s1;
  pq[kl][cl] = q0 + q1;
  __syncthreads();
  if (kl == 0) {
    part[(size_t)blockIdx.y * ld + c] = ps[0][cl] + ps[1][cl] + ps[2][cl] + ps[3][cl];
    part[(size_t)blockIdx.y * ld + cpad + c] = pq[0][cl] + pq[1][cl] + pq[2][cl] + pq[3][cl];
  }
  if (!last_block_of(tickets + blockIdx.x, (unsigned)chunks)) return;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  s0 = q0 = s1 = q1 = 0.0;
  if (c < C && kl < chunks) {
    ordered_pair_sum(kl, chunks, [&](int k) { return part[(size_t)k * ld + c]; }, s0, s1);
    ordered_pair_sum(kl, chunks, [&](int k) { return part[(size_t)k * ld + cpad + c]; }, q0, q1);
  }
  ps[kl][cl] = s0 + s1;
  pq[kl][cl] = q0 + q1;
  __syncthreads();
  if (kl != 0 || c >= C) return;
  const double s = (ps[0][cl] + ps[1][cl]) + (ps[2][cl] + ps[3][cl]);
  const double q = (pq[0][cl] + pq[1][cl]) + (pq[2][cl] + pq[3][cl]);
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_o[c] = (float)mean;
  invstd_o[c] = invstd;
  if (rmean) {
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_coeffs_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o,
                                      float* __restrict__ invstd_o, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rvar[c] + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - rmean[c] * sc;
  if (mean_o) mean_o[c] = rmean[c];
  if (invstd_o) invstd_o[c] = is;
}

// Thread mapping of the streaming kernels: a thread owns ONE 8-channel vector (per-channel
// coefficients live in registers) and walks over pixels; consecutive lanes take consecutive
// channel vectors of a pixel, then the next pixel: fully coalesced, no per-element div/mod.
struct PixVec {
  int nvl, PS, v, ps;
  __device__ __forceinline__ PixVec(int nv) {
    nvl = nv < 256 ? nv : 256;
    PS = 256 / nvl;
    v = threadIdx.x % nvl;
    ps = threadIdx.x / nvl;
  }
};
// pixels per workgroup: 8..32 pixels per thread (the per-channel coefficient set-up is amortised over
// them) while keeping at least ~2048 workgroups in flight for the chip
static inline int pixvec_pixels_per_block(int nv, int64_t npix, int max_per_thread, int min_blocks = 2048) {
  const int nvl = nv < 256 ? nv : 256;
  const int ps = 256 / nvl;
  int per_thread = max_per_thread;
  while (per_thread > 8 && npix / ((int64_t)ps * per_thread) < min_blocks) per_thread >>= 1;
  return ps * per_thread;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ a, int lda,
                                                            int64_t npix, int C, int C8, int pixb) {
  const int nv = C8 >> 3;
  const PixVec m(nv);
  if (m.ps >= m.PS) return;
  const int64_t p0 = (int64_t)blockIdx.x * pixb;
  const int64_t p1 = p0 + pixb < npix ? p0 + pixb : npix;
  for (int vv = m.v; vv < nv; vv += m.nvl) {
    const int c = vv * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = coef(scale, c + j, C);
      sh[j] = coef(shift, c + j, C);
    }
    int64_t pix = p0 + m.ps;
    for (; pix + 3 * m.PS < p1; pix += 4 * m.PS) {          // 4 independent 16-byte loads in flight per lane
      F8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = load8<T>(y + (pix + u * m.PS) * ldy + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = fmaxf(fmaf(v[u].v[j], sc[j], sh[j]), 0.f);
        store8<T>(a + (pix + u * m.PS) * lda + c, o);
      }
    }
    for (; pix < p1; pix += m.PS) {
      const F8 v = load8<T>(y + pix * ldy + c);
      F8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = fmaxf(fmaf(v.v[j], sc[j], sh[j]), 0.f);
      store8<T>(a + pix * lda + c, o);
    }
  }
}

// a = relu(scale*y + shift) AND p = maxpool2x2(a) in one pass over y (an encoder block's output feeds the next level
// through nn.MaxPool2d(2,2) and the decoder through the skip connection: reference src/model.py:268-271).
// One thread = one 2x2 window x 8 channels; grid = (x-chunks of the window row, window rows incl. a last odd row, images).
// argidx (optional) [N][H/2][W/2][C8/8] uint16: for each pooled window and 8-channel vector, 2 bits per channel = position
// (2*dy + dx) of the window's FIRST maximum (scan order and strict '>' of ATen's max_pool2d) -- what the backward needs to route
// the pooled gradient, so that it never re-derives it from the activations (csrc/bn_fused.hip).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_pool_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, T* __restrict__ a, int lda,
                                                                 T* __restrict__ pl, int ldp, unsigned short* __restrict__ argidx, int H, int W,
                                                                 int C, int C8) {
  const int nv = C8 >> 3, Wc = (W + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wc * nv) return;
  const int xo = idx / nv, c = (idx - xo * nv) * 8;
  const int yo = blockIdx.y, n = blockIdx.z;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = coef(scale, c + j, C);
    sh[j] = coef(shift, c + j, C);
  }
  const int y0 = 2 * yo, x0 = 2 * xo;
  const bool hasx = x0 + 1 < W, hasy = y0 + 1 < H;
  const size_t row0 = ((size_t)n * H + y0) * W, row1 = row0 + W;
  F8 v[4];
  v[0] = load8<T>(y + (row0 + x0) * ldy + c);
  v[1] = hasx ? load8<T>(y + (row0 + x0 + 1) * ldy + c) : zero8();
  v[2] = hasy ? load8<T>(y + (row1 + x0) * ldy + c) : zero8();
  v[3] = (hasx && hasy) ? load8<T>(y + (row1 + x0 + 1) * ldy + c) : zero8();
  F8 o[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) o[u].v[j] = fmaxf(fmaf(v[u].v[j], sc[j], sh[j]), 0.f);
  store8<T>(a + (row0 + x0) * lda + c, o[0]);
  if (hasx) store8<T>(a + (row0 + x0 + 1) * lda + c, o[1]);
  if (hasy) store8<T>(a + (row1 + x0) * lda + c, o[2]);
  if (hasx && hasy) store8<T>(a + (row1 + x0 + 1) * lda + c, o[3]);
  if (yo < Ho && xo < Wo) {
    // the pooled value is the max of the ROUNDED activations (what a separate pool pass would read back)
    F8 m;
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a0 = opaque((float)(T)opaque(o[0].v[j])), a1 = opaque((float)(T)opaque(o[1].v[j]));
      const float a2 = opaque((float)(T)opaque(o[2].v[j])), a3 = opaque((float)(T)opaque(o[3].v[j]));
      unsigned arg = 0;
      float mx = a0;
      arg = a1 > mx ? 1u : arg;
      mx = fmaxf(mx, a1);
      arg = a2 > mx ? 2u : arg;
      mx = fmaxf(mx, a2);
      arg = a3 > mx ? 3u : arg;
      mx = fmaxf(mx, a3);
      m.v[j] = mx;
      bits |= arg << (2 * j);
    }
    store8<T>(pl + (((size_t)n * Ho + yo) * Wo + xo) * ldp + c, m);
    if (argidx != nullptr) argidx[(((size_t)n * Ho + yo) * Wo + xo) * nv + (c >> 3)] = (unsigned short)bits;
  }
}

// Backward pass 1: block (64 channels x 32 pixel slots over BWD_PIX_PER_BLOCK pixels); partial sums of dz and dz*xhat.

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const T* __restrict__ da, int ldda, const T* __restrict__ y,
                                                                 int ldy, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* __restrict__ slab,
                                                                 int ldslab, int64_t npix, int C) {
  // threads: 8 channel-vectors (64 channels) x 32 pixel slots
  __shared__ float red[2][32][64 + 1];
  const int cv = threadIdx.x & 7, ps = threadIdx.x >> 3;
  const int c0 = blockIdx.y * 64 + cv * 8;
  const int64_t p0 = (int64_t)blockIdx.x * BWD_PIX_PER_BLOCK;
  const int64_t p1 = p0 + BWD_PIX_PER_BLOCK < npix ? p0 + BWD_PIX_PER_BLOCK : npix;
  float s1[8], s2[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = s2[j] = 0.f;
    const int cc = c0 + j;
    sc[j] = coef(scale, cc, C);
    sh[j] = coef(shift, cc, C);
    mu[j] = coef(mean, cc, C);
    is[j] = coef(invstd, cc, C);
  }
  if (c0 < C) {
    int64_t p = p0 + ps;
    for (; p + 32 < p1; p += 64) {                           // 4 independent 16-byte loads in flight per lane
      F8 g[2], v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        g[u] = load8<T>(da + (p + 32 * u) * ldda + c0);
        v[u] = load8<T>(y + (p + 32 * u) * ldy + c0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float act = fmaf(v[u].v[j], sc[j], sh[j]);
          const float dz = act > 0.f ? g[u].v[j] : 0.f;
          s1[j] += dz;
          s2[j] = fmaf(dz, (v[u].v[j] - mu[j]) * is[j], s2[j]);      // (explicit: the fused forms in bn_fused.hip must round alike)
        }
    }
    for (; p < p1; p += 32) {
      const F8 g = load8<T>(da + p * ldda + c0);
      const F8 v = load8<T>(y + p * ldy + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float act = fmaf(v.v[j], sc[j], sh[j]);
        const float dz = act > 0.f ? g.v[j] : 0.f;
        s1[j] += dz;
        s2[j] = fmaf(dz, (v.v[j] - mu[j]) * is[j], s2[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][ps][cv * 8 + j] = s1[j];
    red[1][ps][cv * 8 + j] = s2[j];
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) s += red[which][r][c];
    const int cc = blockIdx.y * 64 + c;
    if (cc < ldslab) slab[((size_t)blockIdx.x * 2 + which) * ldslab + cc] = s;
  }
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const T* __restrict__ da, int ldda, const T* __restrict__ y,
                                                                int ldy, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const double* __restrict__ sums,
                                                                double inv_count, T* __restrict__ dy, int lddy, int64_t npix,
                                                                int C, int C8, int pixb) {
  const int nv = C8 >> 3;
  const PixVec m(nv);
  if (m.ps >= m.PS) return;
  if (inv_count <= 0.0) inv_count = 1.0 / sums[2 * C];   // data parallel: all-reduced pixel count appended to the sums
  const int64_t p0 = (int64_t)blockIdx.x * pixb;
  const int64_t p1 = p0 + pixb < npix ? p0 + pixb : npix;
  for (int vv = m.v; vv < nv; vv += m.nvl) {
    const int c = vv * 8;
    // dy = sc*(dz - m1 - xhat*m2), xhat = (y - mu)*is   ==>   dy = sc*dz - k0 - k1*y
    float sc[8], sh[8], k0[8], k1[8];
    double d1[8], d2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                        // all 48 loads first, no wait between them (mau_common.h: coef)
      sc[j] = coef(scale, c + j, C);
      sh[j] = coef(shift, c + j, C);
      k0[j] = coef(mean, c + j, C);
      k1[j] = coef(invstd, c + j, C);
      d1[j] = coef(sums, c + j, C);
      d2[j] = coef(sums + C, c + j, C);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mu = k0[j], is = k1[j];
      const float m1 = (float)(d1[j] * inv_count), m2 = (float)(d2[j] * inv_count);
      // (every product rounded on its own, the difference one explicit FMA: the generic and the fused apply kernels must agree
      //  to the bit -- left to the compiler, "a*b - c*d" contracts differently from kernel to kernel, visible in fp16 outputs)
      k1[j] = __fmul_rn(__fmul_rn(sc[j], m2), is);
      k0[j] = fmaf(sc[j], m1, -__fmul_rn(k1[j], mu));
    }
    int64_t pix = p0 + m.ps;
    for (; pix + 3 * m.PS < p1; pix += 4 * m.PS) {          // 8 independent 16-byte loads in flight per lane
      F8 g[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = NT ? load8_nt<T>(da + (pix + u * m.PS) * ldda + c) : load8<T>(da + (pix + u * m.PS) * ldda + c);
        v[u] = NT ? load8_nt<T>(y + (pix + u * m.PS) * ldy + c) : load8<T>(y + (pix + u * m.PS) * ldy + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float act = fmaf(v[u].v[j], sc[j], sh[j]);
          const float dz = act > 0.f ? g[u].v[j] : 0.f;
          o.v[j] = fmaf(sc[j], dz, -fmaf(k1[j], v[u].v[j], k0[j]));
        }
        if (NT) store8_nt<T>(dy + (pix + u * m.PS) * lddy + c, o); else store8<T>(dy + (pix + u * m.PS) * lddy + c, o);
      }
    }
    for (; pix < p1; pix += m.PS) {
      const F8 g = load8<T>(da + pix * ldda + c);
      const F8 v = load8<T>(y + pix * ldy + c);
      F8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float act = fmaf(v.v[j], sc[j], sh[j]);
        const float dz = act > 0.f ? g.v[j] : 0.f;
        o.v[j] = fmaf(sc[j], dz, -fmaf(k1[j], v.v[j], k0[j]));
      }
      store8<T>(dy + pix * lddy + c, o);
    }
  }
}

}  // namespace mau

using namespace mau;

extern "C" {

size_t mau_reduce_rows_ws_elems(int rows, int M) { return (size_t)reduce_chunks(rows) * M; }

int mau_reduce_tickets_elems(void) { return 64; }            // one ticket per 64-column block: M <= 4096 columns

#define MAU_REQUIRE_TICKETS(M) MAU_REQUIRE(tickets == nullptr || ceil_div((M), 64) <= mau_reduce_tickets_elems(), "reduce_rows: more than 4096 columns")

int mau_reduce_rows_f64(const float* slab, int rows, int M, int ldrow, double* sums, double* ws, unsigned* tickets,
                        mau_stream_t stream) {
  MAU_REQUIRE(slab && sums && rows > 0 && M > 0 && ldrow >= M, "reduce_rows: bad arguments");
  MAU_REQUIRE_TICKETS(M);
  if (tickets && ws) return reduce_rows_fused_launch<double>(slab, rows, M, ldrow, M, 0, sums, ws, tickets, (hipStream_t)stream);
  return reduce_rows_launch<double>(slab, rows, M, ldrow, sums, ws, (hipStream_t)stream);
}

int mau_reduce_rows_f64_f32(const float* slab, int rows, int M, int ldrow, double* sums, float* sums32, double* ws,
                            unsigned* tickets, double append, mau_stream_t stream) {
  MAU_REQUIRE(slab && sums && sums32 && rows > 0 && M > 0 && ldrow >= M, "reduce_rows: bad arguments");
  MAU_REQUIRE_TICKETS(M);
  MAU_REQUIRE(append <= 0.0 || (tickets && ws), "reduce_rows: the appended count needs the single-launch form (ws and tickets)");
  if (tickets && ws)
    return reduce_rows_fused_launch<double>(slab, rows, M, ldrow, M, 0, sums, ws, tickets, (hipStream_t)stream, sums32, append);
  return reduce_rows_launch<double>(slab, rows, M, ldrow, sums, ws, (hipStream_t)stream, sums32);
}

int mau_reduce_rows_f32(const float* slab, int rows, int M, int ldrow, float* out, double* ws, unsigned* tickets,
                        mau_stream_t stream) {
  MAU_REQUIRE(slab && out && rows > 0 && M > 0 && ldrow >= M, "reduce_rows: bad arguments");
  MAU_REQUIRE_TICKETS(M);
  if (tickets && ws) return reduce_rows_fused_launch<float>(slab, rows, M, ldrow, M, 0, out, ws, tickets, (hipStream_t)stream);
  return reduce_rows_launch<float>(slab, rows, M, ldrow, out, ws, (hipStream_t)stream);
}

// [sum | sum of squares] of a conv-epilogue statistics slab [rows][2 * cpad] -> sums[0..C) | sums[C..2C), one launch
// (the data-parallel forward: these sums are all-reduced before mau_bn_finalize_train)
int mau_bn_stats_sums_f64(const float* slab, int rows, int C, double* sums, double* ws, unsigned* tickets, double append,
                          mau_stream_t stream) {
  MAU_REQUIRE(slab && sums && ws && tickets && rows > 0 && C > 0, "bn_stats_sums: bad arguments");
  MAU_REQUIRE_TICKETS(2 * C);
  const int cpad = round_up(C, 64);
  return reduce_rows_fused_launch<double>(slab, rows, 2 * C, 2 * cpad, C, cpad, sums, ws, tickets, (hipStream_t)stream, nullptr, append);
}

int mau_bn_finalize_train(const double* sums, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                          float* scale, float* shift, float* mean, float* invstd, int C, mau_stream_t stream) {
  MAU_REQUIRE(sums && gamma && beta && scale && shift && mean && invstd && C > 0 && count >= 0, "bn_finalize_train: bad arguments");
  MAU_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_train: running_mean/var must come together");
  MAU_LAUNCH(bn_finalize_train_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, count, gamma,
                     beta, running_mean, running_var, nbt, momentum, eps, scale, shift, mean, invstd, C);
  return check_launch("bn_finalize_train_kernel");
}

size_t mau_bn_stats_ws_elems(int rows, int C) { return (size_t)reduce_chunks(rows) * 2 * round_up(C, 64); }

int mau_bn_stats_finalize_train(const float* slab, int rows, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* invstd, double* ws, unsigned* tickets, int C,
                                mau_stream_t stream) {
  MAU_REQUIRE(slab && ws && gamma && beta && scale && shift && mean && invstd && rows > 0 && C > 0 && count > 0, "bn_stats_finalize_train: bad arguments");
  MAU_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_finalize_train: running_mean/var must come together");
  hipStream_t st = (hipStream_t)stream;
  const int cpad = round_up(C, 64), M = 2 * cpad, chunks = reduce_chunks(rows);
  MAU_REQUIRE_TICKETS(cpad);
  if (tickets) {                                              // one launch: the last workgroup of a channel block finalizes
    MAU_LAUNCH(bn_stats_finalize_fused_kernel, dim3(cpad / 64, chunks), dim3(256), 0, st, slab, rows, cpad, ws, tickets, count, gamma,
               beta, running_mean, running_var, nbt, momentum, eps, scale, shift, mean, invstd, C);
    return check_launch("bn_stats_finalize_fused_kernel");
  }
  // level 1: the conv epilogue's slab [rows][2*cpad] -> fp64 partials [chunks][2*cpad]
  MAU_LAUNCH((reduce_rows_kernel<float, double>), dim3(ceil_div(M, 64), chunks), dim3(256), 0, st, slab, rows, M, M, ws, M, (float*)nullptr);
  // level 2 + finalize
  MAU_LAUNCH(bn_finalize_from_partials_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, (const double*)ws, chunks, cpad, count, gamma,
             beta, running_mean, running_var, nbt, momentum, eps, scale, shift, mean, invstd, C);
  return check_launch("bn_stats_finalize_train");
}

int mau_bn_coeffs_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, float* mean, float* invstd, int C, mau_stream_t stream) {
  MAU_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_coeffs_eval: bad arguments");
  MAU_LAUNCH(bn_coeffs_eval_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, scale, shift, mean, invstd, C);
  return check_launch("bn_coeffs_eval_kernel");
}

int mau_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift, void* a, int lda, int dtype,
                      int64_t npix, int C, mau_stream_t stream) {
  MAU_REQUIRE(y && a && scale && shift && npix > 0 && C > 0, "bn_relu_apply: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && lda % 8 == 0 && ldy >= C8 && lda >= C8, "bn_relu_apply: bad ld");
  // measured (scripts/elementwise_bench.py): the 2M-pixel level-0 tensors stream best with many short workgroups,
  // the smaller ones with few long ones (the per-channel coefficient set-up is paid once per thread)
  const bool big = npix >= (int64_t)1 << 20;
  const int pixb = pixvec_pixels_per_block(C8 / 8, npix, big ? 8 : 32, big ? 2048 : 256);
  // (measured, round 4: walking the tensor from its END right behind the convolution that wrote it -- hoping for Infinity-Cache hits
  //  on the most recently written 256 MB -- changes nothing: 86 us either way for 2 x 268 MB = 6.2 TB/s, scripts/stream_order_probe.py)
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(bn_relu_apply_kernel<T>, dim3(ceil_div(npix, pixb)), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)y, ldy, scale, shift, (T*)a, lda, npix, C, C8, pixb));
  return check_launch("bn_relu_apply_kernel");
}

int mau_bn_relu_apply_pool(const void* y, int ldy, const float* scale, const float* shift, void* a, int lda, void* pooled,
                           int ldp, unsigned short* argidx, int dtype, int N, int H, int W, int C, mau_stream_t stream) {
  MAU_REQUIRE(y && a && pooled && scale && shift && N > 0 && H >= 2 && W >= 2 && C > 0, "bn_relu_apply_pool: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldy % 8 == 0 && lda % 8 == 0 && ldp % 8 == 0 && ldy >= C8 && lda >= C8 && ldp >= C8, "bn_relu_apply_pool: bad ld");
  MAU_REQUIRE((H + 1) / 2 <= 65535 && N <= 65535, "bn_relu_apply_pool: H/2 and N must fit a grid dimension");
  dim3 grid(ceil_div(((W + 1) / 2) * (C8 / 8), 256), (H + 1) / 2, N);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(bn_relu_apply_pool_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)y, ldy, scale,
                                       shift, (T*)a, lda, (T*)pooled, ldp, argidx, H, W, C, C8));
  return check_launch("bn_relu_apply_pool_kernel");
}

int mau_bn_bwd_rows(int64_t npix) { return ceil_div(npix, BWD_PIX_PER_BLOCK); }

int mau_bn_relu_bwd_reduce(const void* da, int ldda, const void* y, int ldy, const float* scale, const float* shift,
                           const float* mean, const float* invstd, float* slab, int ldslab, int dtype, int64_t npix,
                           int C, mau_stream_t stream) {
  MAU_REQUIRE(da && y && scale && shift && mean && invstd && slab && npix > 0 && C > 0, "bn_relu_bwd_reduce: bad arguments");
  MAU_REQUIRE(ldda % 8 == 0 && ldy % 8 == 0 && ldslab >= C, "bn_relu_bwd_reduce: bad ld");
  dim3 grid(mau_bn_bwd_rows(npix), ceil_div(C, 64));
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(bn_relu_bwd_reduce_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                               (const T*)da, ldda, (const T*)y, ldy, scale, shift, mean, invstd, slab,
                                               ldslab, npix, C));
  return check_launch("bn_relu_bwd_reduce_kernel");
}

int mau_bn_relu_bwd_apply(const void* da, int ldda, const void* y, int ldy, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const double* sums, double count, void* dy, int lddy,
                          int dtype, int64_t npix, int C, mau_stream_t stream) {
  MAU_REQUIRE(da && y && dy && sums && npix > 0 && C > 0 && count >= 0, "bn_relu_bwd_apply: bad arguments");
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(ldda % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && lddy >= C8, "bn_relu_bwd_apply: bad ld");
  // up to 64 pixels per thread and as few as 256 workgroups: this kernel's per-channel set-up (6 coefficient vectors,
  // fp64 means) is what the mid-size layers were paying for (C=256: 61 -> 36 us, scripts/elementwise_bench.py)
  const int pixb = pixvec_pixels_per_block(C8 / 8, npix, 64, 256);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH((bn_relu_bwd_apply_kernel<T, false>), dim3(ceil_div(npix, pixb)), dim3(256), 0, (hipStream_t)stream,
                                               (const T*)da, ldda, (const T*)y, ldy, scale, shift, mean, invstd, sums,
                                               count > 0 ? 1.0 / count : 0.0, (T*)dy, lddy, npix, C, C8, pixb));
  return check_launch("bn_relu_bwd_apply_kernel");
}

}  // extern "C"
