/* Host-side sanitizer check of the C ABI (run by `make -C .../csrc asan` with AddressSanitizer on both sides).
 * Exercises only code that runs on the HOST: size helpers and the argument validation of every entry-point family
 * (each call below must be REFUSED with MAU_ERR_ARG before any launch); no GPU is needed. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mau_hip.h"

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }
#define REFUSED(expr) do { int rc_ = (expr); if (rc_ != MAU_ERR_ARG) { fprintf(stderr, "%s -> %d (expected MAU_ERR_ARG)\n", #expr, rc_); return 3; } \
                           if (strlen(p_mau_last_error()) == 0) { fprintf(stderr, "%s: empty error message\n", #expr); return 4; } } while (0)

int main(int argc, char** argv) {
  void* h = dlopen(argc > 1 ? argv[1] : "libmau_hip_asan.so", RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  SYM(mau_abi_version) SYM(mau_last_error) SYM(mau_conv3x3_kc) SYM(mau_conv3x3_packed_elems) SYM(mau_conv3x3_num_pixel_tiles)
  SYM(mau_conv3x3_wgrad_splits) SYM(mau_conv3x3_wgrad_acc_elems) SYM(mau_reduce_rows_ws_elems) SYM(mau_bn_stats_ws_elems)
  SYM(mau_bn_bwd_rows) SYM(mau_bcast_bwd_ws_elems) SYM(mau_head_bwd_rows) SYM(mau_head_bwd_rowlen) SYM(mau_mse_blocks)
  SYM(mau_l1_gradient_blocks) SYM(mau_lstm_bwd_ws_elems) SYM(mau_ssim_ws_elems) SYM(mau_lstm_max_hidden)
  SYM(mau_conv3x3_fwd) SYM(mau_conv3x3_fwd2) SYM(mau_conv3x3_wgrad2) SYM(mau_bn_relu_apply) SYM(mau_maxpool2x2_fwd)
  SYM(mau_linear_fwd) SYM(mau_resize_bilinear_fwd) SYM(mau_head_fwd) SYM(mau_lstm_fwd) SYM(mau_ssim_loss) SYM(mau_nchw_to_nhwc) SYM(mau_copy_channels) SYM(mau_emb_fold_fwd)
  SYM(mau_conv3x3_first_max_channels) SYM(mau_conv3x3_first_rows) SYM(mau_conv3x3_first_fwd) SYM(mau_conv3x3_first_wgrad_ws_elems) SYM(mau_conv3x3_first_wgrad)
  if (p_mau_abi_version() != 5) return 5;
  size_t acc = 0;
  for (int dt = MAU_F32; dt <= MAU_F16; ++dt) {
    if (p_mau_conv3x3_kc(dt) != 16) return 6;
    for (int c = 1; c <= 1536; c = c * 3 + 1) {
      acc += p_mau_conv3x3_packed_elems(dt, c, 2 * c + 1);
      acc += (size_t)p_mau_conv3x3_num_pixel_tiles(dt, 3, 250, 131, c);
      acc += (size_t)p_mau_conv3x3_wgrad_splits(dt, 32, 256, 256, c, 64);
      acc += p_mau_conv3x3_wgrad_acc_elems(dt, 2, 31, 17, c, 2 * c + 1);
    }
  }
  acc += p_mau_reduce_rows_ws_elems(70000, 2048) + p_mau_bn_stats_ws_elems(70000, 1000) + (size_t)p_mau_bn_bwd_rows(1 << 22);
  acc += p_mau_bcast_bwd_ws_elems(16, 65536, 128) + (size_t)p_mau_head_bwd_rows(32, 65536) + (size_t)p_mau_head_bwd_rowlen(64, 2);
  acc += (size_t)p_mau_mse_blocks(1 << 24) + (size_t)p_mau_l1_gradient_blocks(1 << 24) + p_mau_lstm_bwd_ws_elems(32, 828, 96);
  acc += p_mau_ssim_ws_elems(32, 2, 250, 250) + (size_t)p_mau_lstm_max_hidden();
  if (p_mau_conv3x3_first_max_channels() != 8) return 7;
  acc += (size_t)p_mau_conv3x3_first_rows(32, 256, 256) + (size_t)p_mau_conv3x3_first_rows(1, 31, 47) + p_mau_conv3x3_first_wgrad_ws_elems(32, 256, 256, 64)
         + p_mau_conv3x3_first_wgrad_ws_elems(3, 31, 47, 8) + (size_t)p_mau_conv3x3_first_rows(0, 4, 4);
  float dummy[64];
  REFUSED(p_mau_conv3x3_fwd(dummy, 12, 12, NULL, NULL, 0, dummy, NULL, NULL, NULL, dummy, 8, 8, NULL, MAU_BF16, 1, 4, 4, NULL));   /* ld % 8 */
  REFUSED(p_mau_conv3x3_fwd(NULL, 8, 8, NULL, NULL, 0, dummy, NULL, NULL, NULL, dummy, 8, 8, NULL, MAU_BF16, 1, 4, 4, NULL));      /* null x */
  REFUSED(p_mau_conv3x3_fwd2(dummy, 8, 8, dummy, 8, 8, NULL, NULL, 0, dummy, NULL, NULL, NULL, dummy, 8, 8, NULL, MAU_BF16, 1, 4, 4, NULL));  /* C0 % 16 */
  REFUSED(p_mau_conv3x3_wgrad2(dummy, 8, 8, NULL, 0, 0, dummy, NULL, 12, dummy, 8, 8, dummy, MAU_BF16, 1, 4, 4, NULL));           /* E % 8 */
  REFUSED(p_mau_bn_relu_apply(dummy, 8, NULL, NULL, dummy, 8, MAU_BF16, 16, 8, NULL));
  REFUSED(p_mau_maxpool2x2_fwd(dummy, 8, dummy, 8, MAU_BF16, 1, 1, 1, 8, NULL));                                                /* H < 2 */
  REFUSED(p_mau_resize_bilinear_fwd(dummy, 8, 2, 2, dummy, 8, 4, MAU_BF16, 1, 4, 4, 8, NULL));                                  /* choff % 8 */
  REFUSED(p_mau_head_fwd(NULL, 8, dummy, dummy, dummy, 1, MAU_BF16, 1, 16, 8, 2, NULL));
  REFUSED(p_mau_linear_fwd(dummy, dummy, dummy, NULL, 4, 8, 8, NULL));
  REFUSED(p_mau_lstm_fwd(dummy, dummy, dummy, dummy, dummy, dummy, NULL, NULL, 1, 4, 200, NULL));                               /* H > 128 */
  REFUSED(p_mau_ssim_loss(dummy, dummy, (double*)dummy, dummy, dummy, 1, 1, 2, 8, 8, NULL));                                    /* smaller than the window */
  REFUSED(p_mau_nchw_to_nhwc(dummy, dummy, MAU_BF16, 1, 9, 2, 2, 8, NULL));                                                     /* ld < C */
  REFUSED(p_mau_emb_fold_fwd((const float*)dummy, (const float*)dummy, (float*)dummy, 8, 16, 128, 32, 16, NULL));                             /* Ep < N */
  REFUSED(p_mau_copy_channels(dummy, 8, dummy, 8, 4, 0, MAU_BF16, 4, 8, NULL));                                                 /* choff + C > ld */
  REFUSED(p_mau_conv3x3_first_fwd(dummy, 9, dummy, NULL, NULL, NULL, dummy, 64, 64, NULL, NULL, MAU_BF16, 1, 4, 4, NULL));        /* > 8 input channels */
  REFUSED(p_mau_conv3x3_first_fwd(dummy, 6, dummy, NULL, NULL, NULL, dummy, 64, 64, NULL, NULL, MAU_F32, 1, 4, 4, NULL));         /* fp32 = parity mode */
  REFUSED(p_mau_conv3x3_first_fwd(dummy, 6, dummy, NULL, dummy, NULL, dummy, 64, 64, NULL, NULL, MAU_BF16, 1, 4, 4, NULL));       /* scale without shift */
  REFUSED(p_mau_conv3x3_first_fwd(dummy, 6, dummy, NULL, NULL, NULL, dummy, 12, 12, NULL, NULL, MAU_BF16, 1, 4, 4, NULL));        /* ldy % 8 */
  REFUSED(p_mau_conv3x3_first_wgrad(dummy, dummy, 64, dummy, dummy, 9, 64, MAU_BF16, 1, 4, 4, NULL));                            /* > 8 input channels */
  REFUSED(p_mau_conv3x3_first_wgrad(dummy, dummy, 60, dummy, dummy, 6, 64, MAU_BF16, 1, 4, 4, NULL));                            /* lddz % 8, < Cout */
  REFUSED(p_mau_conv3x3_first_wgrad(dummy, dummy, 64, dummy, NULL, 6, 64, MAU_F16, 1, 4, 4, NULL));                              /* no workspace */
  printf("asan host check OK (%zu)\n", acc);
  return 0;
}
