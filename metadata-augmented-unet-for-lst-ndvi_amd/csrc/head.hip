// head.hip -- output head (1x1 conv + tanh on channel 0, reference src/model.py:241,284-292 and
// :187-193), MetadataEncoder MLP (:38-48) and the MSE criterion (src/utils/losses.py:27-39).
// The head reads NHWC-ld activations and writes the module's NCHW fp32 output directly.
#include "mau_common.h"

namespace mau {

// (HEAD_MAX_CO, HEAD_PIX_PER_BLOCK: mau_common.h)

// 8 lanes per pixel: lane (pixel slot = tid/8, vector = tid%8) loads ONE 16-byte vector, so a wave reads
// 8 pixels x 128 contiguous bytes per instruction; each lane keeps the weights of its 8 channels in
// registers, three shuffle steps sum the 8 lanes, lane o writes output channel o (NCHW fp32).
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, int tanh0, int HW,
                                                       int C, int Co, int64_t npix) {
  const int sub = threadIdx.x & 7;
  const int nv = (C + 7) >> 3;
  float wr[HEAD_MAX_CO][8];                 // weights of vector `sub` (valid when nv <= 8: the usual 64-channel head)
#pragma unroll
  for (int o = 0; o < HEAD_MAX_CO; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[o][j] = (o < Co && sub * 8 + j < C) ? w[o * C + sub * 8 + j] : 0.f;
  const int64_t stride = (int64_t)gridDim.x * 32;
  for (int64_t pix0 = (int64_t)blockIdx.x * 32; pix0 < npix; pix0 += stride) {      // block-uniform trip count
    const int64_t pix = pix0 + (threadIdx.x >> 3);
    const bool live = pix < npix;
    float acc[HEAD_MAX_CO];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o) acc[o] = 0.f;
    if (live) {
      if (sub < nv) {
        const F8 x = load8<T>(a + pix * lda + sub * 8);
#pragma unroll
        for (int o = 0; o < HEAD_MAX_CO; ++o)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[o] = fmaf(x.v[j], wr[o][j], acc[o]);
      }
      for (int v = sub + 8; v < nv; v += 8) {       // heads wider than 64 channels: weights from memory
        const F8 x = load8<T>(a + pix * lda + v * 8);
#pragma unroll
        for (int o = 0; o < HEAD_MAX_CO; ++o)
          if (o < Co) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (v * 8 + j < C) acc[o] = fmaf(x.v[j], w[o * C + v * 8 + j], acc[o]);
          }
      }
    }
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o) {
      acc[o] += __shfl_xor(acc[o], 1);
      acc[o] += __shfl_xor(acc[o], 2);
      acc[o] += __shfl_xor(acc[o], 4);
    }
    if (live && sub < Co) {
      float r = 0.f;
#pragma unroll
      for (int o = 0; o < HEAD_MAX_CO; ++o)
        if (o == sub) r = acc[o];
      r += b[sub];
      if (tanh0 && sub == 0) r = tanhf(r);
      const int64_t n = pix / HW, q = pix - n * HW;
      out[((size_t)n * Co + sub) * HW + q] = r;
    }
  }
}

// da[p][c] = sum_o dz[o][p] * w[o][c];  partial dW[o][c] = sum_p dz[o][p]*a[p][c], db[o] = sum_p dz[o][p]
// block: 8 channel-vector lanes x 32 pixel slots over HEAD_PIX_PER_BLOCK pixels; slab row = [Co][C8 + 8]
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w,
                                                       const float* __restrict__ out, const float* __restrict__ dout,
                                                       T* __restrict__ da, int ldda, float* __restrict__ slab, int tanh0,
                                                       int HW, int C, int C8, int Co, int64_t npix) {
  extern __shared__ float red[];            // [32][Co*(C8+8)] would be too big: reduce per 64-channel group instead
  const int cv = threadIdx.x & 7, ps = threadIdx.x >> 3;
  const int64_t p0 = (int64_t)blockIdx.x * HEAD_PIX_PER_BLOCK;
  const int64_t p1 = p0 + HEAD_PIX_PER_BLOCK < npix ? p0 + HEAD_PIX_PER_BLOCK : npix;
  const int rowlen = Co * (C8 + 8);
  float* row = slab + (size_t)blockIdx.x * rowlen;
  for (int cg = 0; cg < C8; cg += 64) {
    const int c0 = cg + cv * 8;
    float dwp[HEAD_MAX_CO][8], dbp[HEAD_MAX_CO];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o) {
      dbp[o] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dwp[o][j] = 0.f;
    }
    float wr[HEAD_MAX_CO][8];
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o)
#pragma unroll
      for (int j = 0; j < 8; ++j) wr[o][j] = (o < Co && c0 + j < C) ? w[o * C + c0 + j] : 0.f;
    if (c0 < C8) {
      int64_t n = (p0 + ps) / HW;
      int q = (int)((p0 + ps) - n * HW);
      for (int64_t p = p0 + ps; p < p1; p += 32, q += 32) {
        while (q >= HW) {                       // advance (image, pixel-in-image) without a 64-bit division per pixel
          q -= HW;
          ++n;
        }
        float dz[HEAD_MAX_CO];
#pragma unroll
        for (int o = 0; o < HEAD_MAX_CO; ++o) {
          dz[o] = 0.f;
          if (o < Co) {
            const size_t oi = ((size_t)n * Co + o) * HW + q;
            float g = dout[oi];
            if (tanh0 && o == 0) {
              const float t = out[oi];
              g *= fmaf(-t, t, 1.f);              // (explicit: bn_fused.hip's head_dz must round alike)
            }
            dz[o] = g;
            dbp[o] += g;
          }
        }
        const F8 x = load8<T>(a + p * lda + c0);
        F8 g8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c0 + j;
          float s = 0.f;
#pragma unroll
          for (int o = 0; o < HEAD_MAX_CO; ++o)
            if (o < Co) {
              s = fmaf(dz[o], wr[o][j], s);
              dwp[o][j] = fmaf(dz[o], x.v[j], dwp[o][j]);
            }
          g8.v[j] = s;
        }
        store8<T>(da + p * ldda + c0, g8);
      }
    }
    // reduce the 32 pixel slots through LDS: red[ps][o][64]
    __syncthreads();
#pragma unroll
    for (int o = 0; o < HEAD_MAX_CO; ++o)
      if (o < Co) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(ps * HEAD_MAX_CO + o) * 65 + cv * 8 + j] = dwp[o][j];
      }
    __syncthreads();
    for (int t = threadIdx.x; t < Co * 64; t += 256) {
      const int o = t >> 6, c = t & 63;
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += red[(r * HEAD_MAX_CO + o) * 65 + c];
      if (cg + c < C8) row[o * (C8 + 8) + cg + c] = s;
    }
    if (cg == 0) {
      __syncthreads();
      // bias partials: only the cv == 0 lanes carry the complete per-slot sums (every cv lane of a
      // slot saw the same pixels, so take one of them)
      if (cv == 0) {
#pragma unroll
        for (int o = 0; o < HEAD_MAX_CO; ++o)
          if (o < Co) red[ps * HEAD_MAX_CO + o] = dbp[o];
      }
      __syncthreads();
      if (threadIdx.x < Co) {
        float s = 0.f;
        for (int r = 0; r < 32; ++r) s += red[r * HEAD_MAX_CO + threadIdx.x];
        row[threadIdx.x * (C8 + 8) + C8] = s;
      }
    }
  }
}

// ---- MetadataEncoder: tiny; one block, fp32 ------------------------------------------------------
__global__ void meta_mlp_fwd_kernel(const float* __restrict__ md, const float* __restrict__ w0, const float* __restrict__ b0,
                                    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ hidden,
                                    float* __restrict__ emb, int N, int F, int Hd, int D) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * Hd; i += gridDim.x * blockDim.x) {
    const int n = i / Hd, j = i % Hd;
    float s = b0[j];
    for (int f = 0; f < F; ++f) s = fmaf(md[n * F + f], w0[j * F + f], s);
    hidden[i] = fmaxf(s, 0.f);
  }
}
__global__ void meta_mlp_fwd2_kernel(const float* __restrict__ hidden, const float* __restrict__ w2, const float* __restrict__ b2,
                                     float* __restrict__ emb, int N, int Hd, int D) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * D; i += gridDim.x * blockDim.x) {
    const int n = i / D, d = i % D;
    float s = b2[d];
    for (int j = 0; j < Hd; ++j) s = fmaf(hidden[n * Hd + j], w2[d * Hd + j], s);
    emb[i] = s;
  }
}
// one block; N is a batch (<= a few hundred), Hd = 32, D <= 256: everything fits one workgroup's loops
// Backward in two grid-wide phases (a single 256-thread block looping over everything took 88 us of pure latency):
// phase 1: dW2, db2 and dhidden (through the ReLU);  phase 2: dW0, db0 from dhidden.  Sums over the batch in index order
// (loops unrolled x8: eight iterations' loads in flight, the fma chain keeps its order -- the kernels are chains of L2 round trips).
__global__ void meta_mlp_bwd1_kernel(const float* __restrict__ w2, const float* __restrict__ hidden, const float* __restrict__ demb,
                                     float* __restrict__ dw2, float* __restrict__ db2, float* __restrict__ dhid_ws, int N, int Hd, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D * Hd) {                            // dW2[d][j] = sum_n demb[n][d] * hidden[n][j]
    const int d = i / Hd, j = i % Hd;
    float s = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) s = fmaf(demb[n * D + d], hidden[n * Hd + j], s);
    dw2[i] = s;
  }
  if (i < D) {
    float s = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) s += demb[n * D + i];
    db2[i] = s;
  }
  if (i < N * Hd) {                            // dhidden (through ReLU)
    const int n = i / Hd, j = i % Hd;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) s = fmaf(demb[n * D + d], w2[d * Hd + j], s);
    dhid_ws[i] = hidden[i] > 0.f ? s : 0.f;
  }
}
__global__ void meta_mlp_bwd2_kernel(const float* __restrict__ md, const float* __restrict__ dhid_ws, float* __restrict__ dw0,
                                     float* __restrict__ db0, int N, int F, int Hd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Hd * F) {
    const int j = i / F, f = i % F;
    float s = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) s = fmaf(dhid_ws[n * Hd + j], md[n * F + f], s);
    dw0[i] = s;
  }
  if (i < Hd) {
    float s = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) s += dhid_ws[n * Hd + i];
    db0[i] = s;
  }
}

// ---- MSE: per-block fp64 partial of (out-tgt)^2 and dout = 2*(out-tgt)/n -----------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                  double* __restrict__ partial, float* __restrict__ dout, int64_t n, float k) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(out + i), b = *reinterpret_cast<const f32x4*>(tgt + i);
      f32x4 d = a - b;
      s += (double)(d[0] * d[0]) + (double)(d[1] * d[1]) + (double)(d[2] * d[2]) + (double)(d[3] * d[3]);
      if (dout) *reinterpret_cast<f32x4*>(dout + i) = d * k;
    } else {
      for (int64_t j = i; j < n; ++j) {
        const float d = out[j] - tgt[j];
        s += (double)(d * d);
        if (dout) dout[j] = d * k;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_final_kernel(const double* __restrict__ partial, int nblocks, double inv_n, float* __restrict__ loss) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += partial[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) * inv_n);
}

// ---- L1 and gradient (finite-difference) losses of src/utils/losses.py:5-25,68 on (B,C,H,W) fp32 ------------
// One thread per element.  partial[block] = {sum |o-t|, sum | |dy o| - |dy t| |, sum | |dx o| - |dx t| |} (fp64);
// dout (optional) = k_l1 * sign(o-t) + k_gy * d/do(sum_y) + k_gx * d/do(sum_x), where the coefficients fold the
// 1/count means and the lambda weights.  Each element sees the two vertical and the two horizontal differences
// it takes part in (gather form: no atomics).
__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) - (v < 0.f); }

__global__ __launch_bounds__(256) void l1_gradient_loss_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                               double* __restrict__ partial, float* __restrict__ dout, int H,
                                                               int W, int64_t n, float k_l1, float k_gy, float k_gx) {
  __shared__ double red[3][4];
  double s_l1 = 0.0, s_gy = 0.0, s_gx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float o = out[i], t = tgt[i];
    float g = k_l1 * sgnf(o - t);
    s_l1 += fabsf(o - t);
    if (y + 1 < H) {            // difference (y+1) - y : this element is the minuend's partner "p[y]"
      const float a = out[i + W] - o, b = tgt[i + W] - t;
      const float d = fabsf(a) - fabsf(b);
      s_gy += fabsf(d);
      g -= k_gy * sgnf(d) * sgnf(a);
    }
    if (y > 0) {                // difference y - (y-1) : this element is "p[y+1]" of the row above
      const float a = o - out[i - W], b = t - tgt[i - W];
      g += k_gy * sgnf(fabsf(a) - fabsf(b)) * sgnf(a);
    }
    if (x + 1 < W) {
      const float a = out[i + 1] - o, b = tgt[i + 1] - t;
      const float d = fabsf(a) - fabsf(b);
      s_gx += fabsf(d);
      g -= k_gx * sgnf(d) * sgnf(a);
    }
    if (x > 0) {
      const float a = o - out[i - 1], b = t - tgt[i - 1];
      g += k_gx * sgnf(fabsf(a) - fabsf(b)) * sgnf(a);
    }
    if (dout) dout[i] = g;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s_l1 += __shfl_xor(s_l1, o);
    s_gy += __shfl_xor(s_gy, o);
    s_gx += __shfl_xor(s_gx, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s_l1;
    red[1][threadIdx.x >> 6] = s_gy;
    red[2][threadIdx.x >> 6] = s_gx;
  }
  __syncthreads();
  if (threadIdx.x < 3) partial[(size_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
// terms[0..2] = {mean |o-t|, mean_y term, mean_x term}
__global__ void l1_gradient_final_kernel(const double* __restrict__ partial, int nblocks, double inv_n, double inv_ny, double inv_nx,
                                         float* __restrict__ terms) {
  __shared__ double red[3][4];
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x)
    for (int k = 0; k < 3; ++k) s[k] += partial[(size_t)i * 3 + k];
  for (int k = 0; k < 3; ++k) {
    for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    terms[threadIdx.x] = (float)(v * (threadIdx.x == 0 ? inv_n : threadIdx.x == 1 ? inv_ny : inv_nx));
  }
}

// ---- nn.Linear (TemporalEncoder.fc, reference src/model.py:27,34): out (N,D) = x (N,F) w^T (D,F) + b ------------------
__global__ void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                  float* __restrict__ out, int N, int F, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D) return;
  const int n = i / D, d = i % D;
  float s = b ? b[d] : 0.f;
  for (int f = 0; f < F; ++f) s = fmaf(x[n * F + f], w[d * F + f], s);
  out[i] = s;
}
// dx (N,F) = dout w;  dw (D,F) = dout^T x;  db (D) = sum_n dout  (fixed summation order over n: reproducible)
__global__ void linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dout,
                                  float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int N, int F, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D * F) {
    const int d = i / F, f = i % F;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(dout[n * D + d], x[n * F + f], s);
    dw[i] = s;
  }
  if (i < D) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dout[n * D + i];
    db[i] = s;
  }
  if (dx != nullptr && i < N * F) {
    const int n = i / F, f = i % F;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(dout[n * D + d], w[d * F + f], s);
    dx[i] = s;
  }
}

}  // namespace mau

using namespace mau;

extern "C" {

int mau_head_fwd(const void* a, int lda, const float* w, const float* b, float* out, int tanh0, int dtype, int N, int HW,
                 int C, int Co, mau_stream_t stream) {
  MAU_REQUIRE(a && w && b && out && N > 0 && HW > 0 && C > 0, "head_fwd: bad arguments");
  MAU_REQUIRE(Co >= 1 && Co <= HEAD_MAX_CO, "head_fwd: out_channels must be in [1,%d]", HEAD_MAX_CO);
  MAU_REQUIRE(lda % 8 == 0 && lda >= round_up(C, 8), "head_fwd: bad ld");
  const int64_t npix = (int64_t)N * HW;
  // heads of at most 64 channels (every model of the reference): four pixels per thread and iteration, no per-pixel division
  // (bn_fused.hip: head_bn_fwd_kernel<..., BN = false>, the same arithmetic in the same order)
  if (C <= mau_head_bn_max_channels() && HW >= 32 && npix < ((int64_t)1 << 31))
    return head_fwd_fast(a, lda, w, b, out, tanh0, dtype, npix, HW, C, Co, (hipStream_t)stream);
  const int grid = stream_grid(npix * 8, 256);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(head_fwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, w, b, out, tanh0, HW, C, Co, npix));
  return check_launch("head_fwd_kernel");
}

int mau_head_bwd_rows(int N, int HW) { return ceil_div((int64_t)N * HW, HEAD_PIX_PER_BLOCK); }
int mau_head_bwd_rowlen(int C, int Co) { return Co * (round_up(C, 8) + 8); }

int mau_head_bwd(const void* a, int lda, const float* w, const float* out, const float* dout, void* da, int ldda,
                 float* slab, int tanh0, int dtype, int N, int HW, int C, int Co, mau_stream_t stream) {
  MAU_REQUIRE(a && w && out && dout && da && slab && N > 0 && HW > 0 && C > 0, "head_bwd: bad arguments");
  MAU_REQUIRE(Co >= 1 && Co <= HEAD_MAX_CO, "head_bwd: out_channels must be in [1,%d]", HEAD_MAX_CO);
  const int C8 = round_up(C, 8);
  MAU_REQUIRE(lda % 8 == 0 && ldda % 8 == 0 && lda >= C8 && ldda >= C8, "head_bwd: bad ld");
  const int64_t npix = (int64_t)N * HW;
  const size_t lds = (size_t)32 * HEAD_MAX_CO * 65 * sizeof(float);
  MAU_DISPATCH_DTYPE(dtype, MAU_LAUNCH(head_bwd_kernel<T>, dim3(mau_head_bwd_rows(N, HW)), dim3(256), lds, (hipStream_t)stream, (const T*)a, lda, w, out, dout, (T*)da, ldda, slab, tanh0, HW, C, C8, Co, npix));
  return check_launch("head_bwd_kernel");
}

int mau_meta_mlp_fwd(const float* md, const float* w0, const float* b0, const float* w2, const float* b2, float* hidden,
                     float* emb, int N, int F, int Hd, int D, mau_stream_t stream) {
  MAU_REQUIRE(md && w0 && b0 && w2 && b2 && hidden && emb && N > 0 && F > 0 && Hd > 0 && D > 0, "meta_mlp_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  MAU_LAUNCH(meta_mlp_fwd_kernel, dim3(ceil_div(N * Hd, 256)), dim3(256), 0, st, md, w0, b0, w2, b2, hidden, emb, N, F, Hd, D);
  MAU_LAUNCH(meta_mlp_fwd2_kernel, dim3(ceil_div(N * D, 256)), dim3(256), 0, st, hidden, w2, b2, emb, N, Hd, D);
  return check_launch("meta_mlp_fwd_kernel");
}

int mau_meta_mlp_bwd(const float* md, const float* w0, const float* w2, const float* hidden, const float* demb, float* dw0,
                     float* db0, float* dw2, float* db2, float* dhidden_ws, int N, int F, int Hd, int D, mau_stream_t stream) {
  MAU_REQUIRE(md && w0 && w2 && hidden && demb && dw0 && db0 && dw2 && db2 && dhidden_ws, "meta_mlp_bwd: null pointer");
  const int work1 = D * Hd > N * Hd ? D * Hd : N * Hd, work2 = Hd * F > Hd ? Hd * F : Hd;
  MAU_LAUNCH(meta_mlp_bwd1_kernel, dim3(ceil_div(work1, 128)), dim3(128), 0, (hipStream_t)stream, w2, hidden, demb, dw2, db2, dhidden_ws, N, Hd, D);
  MAU_LAUNCH(meta_mlp_bwd2_kernel, dim3(ceil_div(work2, 128)), dim3(128), 0, (hipStream_t)stream, md, (const float*)dhidden_ws, dw0, db0, N, F, Hd);
  return check_launch("meta_mlp_bwd_kernel");
}

int mau_mse_blocks(int64_t n) { return stream_grid((n + 3) / 4, 256); }

int mau_mse_fwd_bwd(const float* out, const float* tgt, double* partial, float* loss, float* dout, int64_t n,
                    mau_stream_t stream) {
  MAU_REQUIRE(out && tgt && partial && loss && n > 0, "mse_fwd_bwd: bad arguments");
  MAU_REQUIRE(((uintptr_t)out % 16) == 0 && ((uintptr_t)tgt % 16) == 0 && (!dout || ((uintptr_t)dout % 16) == 0), "mse_fwd_bwd: 16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = mau_mse_blocks(n);
  MAU_LAUNCH(mse_kernel, dim3(blocks), dim3(256), 0, st, out, tgt, partial, dout, n, 2.0f / (float)n);
  MAU_LAUNCH(mse_final_kernel, dim3(1), dim3(256), 0, st, partial, blocks, 1.0 / (double)n, loss);
  return check_launch("mse_kernel");
}

int mau_l1_gradient_blocks(int64_t n) { return stream_grid(n, 256); }

int mau_l1_gradient_loss(const float* out, const float* tgt, double* partial, float* terms, float* dout, float w_l1,
                         float w_grad, int B, int C, int H, int W, mau_stream_t stream) {
  MAU_REQUIRE(out && tgt && partial && terms && B > 0 && C > 0 && H > 0 && W > 0, "l1_gradient_loss: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * C * H * W, ny = (int64_t)B * C * (H - 1) * W, nx = (int64_t)B * C * H * (W - 1);
  const int blocks = mau_l1_gradient_blocks(n);
  const float k_l1 = w_l1 / (float)n, k_gy = ny > 0 ? w_grad / (float)ny : 0.f, k_gx = nx > 0 ? w_grad / (float)nx : 0.f;
  MAU_LAUNCH(l1_gradient_loss_kernel, dim3(blocks), dim3(256), 0, st, out, tgt, partial, dout, H, W, n, k_l1, k_gy, k_gx);
  MAU_LAUNCH(l1_gradient_final_kernel, dim3(1), dim3(256), 0, st, partial, blocks, 1.0 / (double)n, ny > 0 ? 1.0 / (double)ny : 0.0,
             nx > 0 ? 1.0 / (double)nx : 0.0, terms);
  return check_launch("l1_gradient_loss_kernel");
}

int mau_linear_fwd(const float* x, const float* w, const float* b, float* out, int N, int F, int D, mau_stream_t stream) {
  MAU_REQUIRE(x && w && out && N > 0 && F > 0 && D > 0, "linear_fwd: bad arguments");
  MAU_LAUNCH(linear_fwd_kernel, dim3(ceil_div(N * D, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, out, N, F, D);
  return check_launch("linear_fwd_kernel");
}

int mau_linear_bwd(const float* x, const float* w, const float* dout, float* dx, float* dw, float* db, int N, int F, int D,
                   mau_stream_t stream) {
  MAU_REQUIRE(x && w && dout && dw && db && N > 0 && F > 0 && D > 0, "linear_bwd: bad arguments");
  const int work = D * F > N * F ? D * F : N * F;
  MAU_LAUNCH(linear_bwd_kernel, dim3(ceil_div(work, 256)), dim3(256), 0, (hipStream_t)stream, x, w, dout, dx, dw, db, N, F, D);
  return check_launch("linear_bwd_kernel");
}

}  // extern "C"
