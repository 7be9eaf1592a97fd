// ssim.hip -- the SSIM term of compute_loss_l1_grad_ssim (reference src/utils/losses.py:70-97: piq.ssim(outputs_ssim,
// targets_ssim, data_range=1.0, reduction='none') per sample, then 1 - mean) as one fused HIP reduction.
//
// piq is not available in the build environment; the kernel follows piq.ssim's published defaults (11x11 Gaussian window,
// sigma 1.5, k1 0.01, k2 0.03, "valid" windows, average-pool downsampling by max(1, round(min(H, W) / 256))) -- the same
// formula the torch-op version in mau_amd/losses.py::ssim_value_torch spells out.  PARITY UNPINNED (no fixture from the
// reference exists for this scalar); it carries no gradient in the reference (torch.Tensor(ssim_vals) at :96 detaches it).
//
// The reference's channel preparation (:72-84) is fused into the tile load: channel 0 -> (v + 1) / 2, channel 1 ->
// clamp(v, 0, 1).  One block = one 16x16 tile of SSIM-map pixels of one (image, channel): 26x26 input window in LDS,
// separable Gaussian (horizontal pass to LDS, vertical pass in registers) over x, y, x^2, y^2, xy, block sum in fp64.
#include <math.h>
#include "mau_common.h"

namespace mau {

constexpr int SS_K = 11, SS_T = 16, SS_IN = SS_T + SS_K - 1;   // 26

struct SsimW {
  float g[SS_K];
};

__global__ __launch_bounds__(256) void ssim_tiles_kernel(const float* __restrict__ out, const float* __restrict__ tgt, double* __restrict__ part,
                                                         SsimW w, int C, int H, int W, int f, int Ho, int Wo, int tilesX, int tilesY,
                                                         int prep) {
  __shared__ float xs[SS_IN][SS_IN + 1], ys[SS_IN][SS_IN + 1];
  __shared__ float hz[5][SS_IN][SS_T + 1];
  __shared__ double red[256];
  const int bc = blockIdx.z, c = bc % C;
  const int ty0 = blockIdx.y * SS_T, tx0 = blockIdx.x * SS_T;
  const float* ob = out + (size_t)bc * H * W;
  const float* tb = tgt + (size_t)bc * H * W;
  const int Hd = H / f, Wd = W / f;
  const float inv = 1.f / (float)(f * f);
  for (int i = threadIdx.x; i < SS_IN * SS_IN; i += 256) {
    const int r = i / SS_IN, cc = i % SS_IN;
    const int y = ty0 + r, x = tx0 + cc;
    float xv = 0.f, yv = 0.f;
    if (y < Hd && x < Wd) {
      for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
          float a = ob[(size_t)(y * f + dy) * W + x * f + dx], b = tb[(size_t)(y * f + dy) * W + x * f + dx];
          if (prep) {
            if (c == 0) {
              a = (a + 1.f) * 0.5f;
              b = (b + 1.f) * 0.5f;
            } else if (c == 1) {
              a = fminf(fmaxf(a, 0.f), 1.f);
              b = fminf(fmaxf(b, 0.f), 1.f);
            }
          }
          xv += a;
          yv += b;
        }
      xv *= inv;
      yv *= inv;
    }
    xs[r][cc] = xv;
    ys[r][cc] = yv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SS_IN * SS_T; i += 256) {
    const int r = i / SS_T, cc = i % SS_T;
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < SS_K; ++k) {
      const float a = xs[r][cc + k], b = ys[r][cc + k], g = w.g[k];
      s[0] = fmaf(g, a, s[0]);
      s[1] = fmaf(g, b, s[1]);
      s[2] = fmaf(g, a * a, s[2]);
      s[3] = fmaf(g, b * b, s[3]);
      s[4] = fmaf(g, a * b, s[4]);
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) hz[m][r][cc] = s[m];
  }
  __syncthreads();
  const int r = threadIdx.x / SS_T, cc = threadIdx.x % SS_T;
  double v = 0.0;
  if (ty0 + r < Ho && tx0 + cc < Wo) {
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < SS_K; ++k) {
      const float g = w.g[k];
#pragma unroll
      for (int m = 0; m < 5; ++m) s[m] = fmaf(g, hz[m][r + k][cc], s[m]);
    }
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    const float mx = s[0], my = s[1];
    const float sxx = s[2] - mx * mx, syy = s[3] - my * my, sxy = s[4] - mx * my;
    const float cs = (2.f * sxy + c2) / (sxx + syy + c2);
    v = (double)((2.f * mx * my + c1) / (mx * mx + my * my + c1) * cs);
  }
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)bc * tilesX * tilesY + blockIdx.y * tilesX + blockIdx.x] = red[0];
}

// per_image[b] = mean over channels of (sum of the tiles of (b, c) / (Ho*Wo)); loss[0] = 1 - mean over images
__global__ void ssim_finalize_kernel(const double* __restrict__ part, float* __restrict__ per_image, float* __restrict__ loss, int B, int C,
                                     int ntiles, double inv_pix) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double tot = 0.0;
  for (int b = 0; b < B; ++b) {
    double sb = 0.0;
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int t = 0; t < ntiles; ++t) s += part[((size_t)b * C + c) * ntiles + t];
      sb += s * inv_pix;
    }
    sb /= (double)C;
    per_image[b] = (float)sb;
    tot += sb;
  }
  loss[0] = (float)(1.0 - tot / (double)B);
}

}  // namespace mau

using namespace mau;

extern "C" {

static void ssim_geometry(int H, int W, int* f, int* Ho, int* Wo, int* tx, int* ty) {
  const int m = H < W ? H : W;
  int ff = (int)nearbyint(m / 256.0);    // round half to even, as Python's round() in piq's downsampling factor
  if (ff < 1) ff = 1;
  *f = ff;
  *Ho = H / ff - (SS_K - 1);
  *Wo = W / ff - (SS_K - 1);
  *tx = *Wo > 0 ? ceil_div(*Wo, SS_T) : 0;
  *ty = *Ho > 0 ? ceil_div(*Ho, SS_T) : 0;
}

size_t mau_ssim_ws_elems(int B, int C, int H, int W) {
  int f, Ho, Wo, tx, ty;
  ssim_geometry(H, W, &f, &Ho, &Wo, &tx, &ty);
  return (size_t)B * C * tx * ty;
}

int mau_ssim_loss(const float* out, const float* tgt, double* ws, float* per_image, float* loss, int prep, int B, int C, int H, int W,
                  mau_stream_t stream) {
  MAU_REQUIRE(out && tgt && ws && per_image && loss && B > 0 && C > 0, "ssim_loss: bad arguments");
  int f, Ho, Wo, tx, ty;
  ssim_geometry(H, W, &f, &Ho, &Wo, &tx, &ty);
  MAU_REQUIRE(Ho > 0 && Wo > 0, "ssim_loss: image %dx%d is smaller than the 11x11 window after downsampling by %d", H, W, f);
  MAU_REQUIRE((int64_t)B * C <= 65535 && ty <= 65535, "ssim_loss: B*C and tile rows must fit a grid dimension");
  SsimW w;
  double sum = 0.0, g[SS_K];
  for (int i = 0; i < SS_K; ++i) {
    const double d = i - (SS_K - 1) / 2.0;
    g[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5));
    sum += g[i];
  }
  for (int i = 0; i < SS_K; ++i) w.g[i] = (float)(g[i] / sum);
  hipStream_t st = (hipStream_t)stream;
  MAU_LAUNCH(ssim_tiles_kernel, dim3(tx, ty, B * C), dim3(256), 0, st, out, tgt, ws, w, C, H, W, f, Ho, Wo, tx, ty, prep);
  MAU_LAUNCH(ssim_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, per_image, loss, B, C, tx * ty, 1.0 / ((double)Ho * Wo));
  return check_launch("ssim_tiles_kernel");
}

}  // extern "C"
