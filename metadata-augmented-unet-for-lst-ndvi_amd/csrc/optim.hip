// optim.hip -- the AdamW update of every 3x3 convolution weight of a network AND the re-pack of the updated weights, one launch.
//
// An optimizer step of the reference (torch.optim.AdamW, src/train.py:213-214,255) is followed, on this path, by re-packing all
// convolution weights into the matrix-core layouts (forward pack and data-gradient pack, conv3x3.hip).  As separate passes that is
//     AdamW:  read p, g, m, v; write p, m, v            (7 x 4 bytes per parameter)
//     pack :  read p (twice: the two packs are tiled differently); write 2 x 2 bytes
// Here one workgroup owns a 64 (output channel) x 64 (input channel) x 9 (tap) block of ONE layer: it streams the block's p, g, m, v
// once, applies AdamW, writes p, m, v back and keeps the new weights in LDS (147 KB), from which BOTH packs of the block -- four
// 16-channel chunks each, rows permuted for the 16-bit kernels exactly as pack_weights_kernel does -- are written.
//     fused:  read p, g, m, v; write p, m, v, 2 x 2 bytes        (32 bytes per parameter instead of 40, one launch instead of ~5)
// The update is torch's AdamW (decoupled weight decay, bias correction, eps outside the root):
//     p <- p (1 - lr wd);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2;  p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with the step count t read from DEVICE memory (the step is capturable into a hipGraph).
#include "mau_common.h"

namespace mau {

struct AdamWPackDesc {
  float* w;
  const float* g;
  float* m;
  float* v;
  void* wf;
  void* wd;
  int Cout, Cin, nCoB, tile0;
};

constexpr int OPT_ROW = 64 * 9 + 1;      // LDS row: 64 input channels x 9 taps (+1: bank spread)

template <typename T>
__global__ __launch_bounds__(1024) void adamw_pack_kernel(const AdamWPackDesc* __restrict__ descs, int n, const float* __restrict__ step_ptr,
                                                          float lr, float beta1, float beta2, float eps, float wd) {
  extern __shared__ float tile[];                       // [64][OPT_ROW]: tile[co_local][ci_local * 9 + tap]
  int i = 0;
  while (i + 1 < n && (int)blockIdx.x >= descs[i + 1].tile0) ++i;
  const AdamWPackDesc d = descs[i];
  const int t = (int)blockIdx.x - d.tile0;
  const int cob = t % d.nCoB, cib = t / d.nCoB;
  const int co0 = cob * 64, ci0 = cib * 64;
  const int Cout = d.Cout, Cin = d.Cin;
  const int CoutPad = (Cout + 63) / 64 * 64, CinPad = (Cin + 63) / 64 * 64;
  const float step = *step_ptr;
  const float bc1 = 1.f - powf(beta1, step), bc2s = sqrtf(1.f - powf(beta2, step));
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  // ---- AdamW on the block: row r = output channel co0 + r, 576 contiguous floats (64 input channels x 9 taps) of the OIHW tensor ----
  const int ncol = (Cin - ci0 < 64 ? Cin - ci0 : 64) * 9;          // valid floats of a row
  for (int e = threadIdx.x; e < 64 * 576; e += 1024) {
    const int r = e / 576, k = e - r * 576;
    float pn = 0.f;
    if (co0 + r < Cout && k < ncol) {
      const size_t idx = ((size_t)(co0 + r) * Cin + ci0) * 9 + k;
      const float g = d.g[idx];
      float p = d.w[idx], m = d.m[idx], v = d.v[idx];
      p *= decay;
      m = fmaf(beta1, m, (1.f - beta1) * g);                 // (lerp(m, g, 1 - b1))
      v = fmaf(beta2, v, (1.f - beta2) * g * g);
      const float denom = sqrtf(v) / bc2s + eps;
      p = p - step_size * (m / denom);
      d.w[idx] = p;
      d.m[idx] = m;
      d.v[idx] = v;
      pn = p;
    }
    tile[r * OPT_ROW + k] = pn;
  }
  __syncthreads();
  if (d.wf == nullptr && d.wd == nullptr) return;
  // ---- the block's share of both packs (layouts of pack_tile, conv3x3.hip): four 16-channel chunks each ----
  // forward  wf[((chunk * 9 + tap) * CoutPad + co0 + pos) * 16 + k] = W[co0 + ch(pos)][chunk * 16 + k][tap],       chunk = ci0 / 16 + cc
  // dgrad    wd[((chunk * 9 + 8 - tap) * CinPad + ci0 + pos) * 16 + k] = W[chunk * 16 + k][ci0 + ch(pos)][tap],    chunk = co0 / 16 + cc
  // (ch(pos) = 2 (pos & 31) + (pos >> 5) for the 16-bit packs: the two accumulator tiles of an MFMA lane carry adjacent channels)
  const int nchF = (Cin + 15) / 16, nchD = (Cout + 15) / 16;
  T* wf = (T*)d.wf;
  T* wdp = (T*)d.wd;
  for (int e = threadIdx.x; e < 2 * 4 * 9 * 64 * 2; e += 1024) {
    const int k8 = e & 1, pos = (e >> 1) & 63, tap = (e >> 7) % 9, cc = ((e >> 7) / 9) & 3, which = (e >> 7) / 36;
    const int ch = sizeof(T) == 2 ? 2 * (pos & 31) + (pos >> 5) : pos;
    F8 val;
    if (which == 0) {
      const int chunk = ci0 / 16 + cc;
      if (wf == nullptr || chunk >= nchF) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) val.v[j] = tile[ch * OPT_ROW + (cc * 16 + k8 * 8 + j) * 9 + tap];
      store8<T>(wf + (((size_t)chunk * 9 + tap) * CoutPad + co0 + pos) * 16 + k8 * 8, val);
    } else {
      const int chunk = co0 / 16 + cc;
      if (wdp == nullptr || chunk >= nchD) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) val.v[j] = tile[(cc * 16 + k8 * 8 + j) * OPT_ROW + ch * 9 + tap];
      store8<T>(wdp + (((size_t)chunk * 9 + 8 - tap) * CinPad + ci0 + pos) * 16 + k8 * 8, val);
    }
  }
}

}  // namespace mau

using namespace mau;

extern "C" {

size_t mau_adamw_pack_desc_bytes(void) { return sizeof(AdamWPackDesc); }

int mau_adamw_pack_desc_fill(void* descs_host, int index, float* w, const float* grad, float* exp_avg, float* exp_avg_sq, void* wf,
                             void* wd, int Cout, int Cin, int tile0, int* next_tile_host) {
  MAU_REQUIRE(descs_host && next_tile_host && index >= 0 && w && grad && exp_avg && exp_avg_sq && Cout > 0 && Cin > 0 && tile0 >= 0,
              "adamw_pack_desc_fill: bad arguments");
  AdamWPackDesc* d = reinterpret_cast<AdamWPackDesc*>(descs_host) + index;
  d->w = w; d->g = grad; d->m = exp_avg; d->v = exp_avg_sq; d->wf = wf; d->wd = wd; d->Cout = Cout; d->Cin = Cin;
  d->nCoB = round_up(Cout, 64) / 64; d->tile0 = tile0;
  *next_tile_host = tile0 + d->nCoB * (round_up(Cin, 64) / 64);
  return MAU_OK;
}

int mau_adamw_pack_step(const void* descs, int n, int total_tiles, int dtype, const float* step, float lr, float beta1, float beta2,
                        float eps, float weight_decay, mau_stream_t stream) {
  MAU_REQUIRE(descs && step && n > 0 && total_tiles > 0 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f,
              "adamw_pack_step: bad arguments");
  const size_t lds = (size_t)64 * OPT_ROW * sizeof(float);
  MAU_DISPATCH_DTYPE(dtype, {
    MAU_LDS_ATTR(lds, &adamw_pack_kernel<T>);
    MAU_LAUNCH(adamw_pack_kernel<T>, dim3(total_tiles), dim3(1024), lds, (hipStream_t)stream, (const AdamWPackDesc*)descs, n, step, lr, beta1,
               beta2, eps, weight_decay);
  });
  return check_launch("adamw_pack_kernel");
}

}  // extern "C"
