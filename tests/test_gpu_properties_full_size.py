"""Size-independent properties of the convolution kernels AT THE SIZES THE BENCH RUNS (BASELINE configs[1]: B = 32, 256 x 256, the
U-Net's eighteen layer shapes) -- no CPU reference is affordable there, and none is needed:

  * **scaling** (exact): multiplying the input -- or the output gradient -- by 2 is exact in bf16 and commutes with every fp32
    partial sum and every rounding, so conv(2x) == 2 conv(x), dgrad(2 dy) == 2 dgrad(dy), wgrad(x, 2 dy) == 2 wgrad(x, dy) and the
    BatchNorm partial sums scale by 2 and 4 BIT FOR BIT, whatever the tiling, the split-K partition or the MFMA lane maps do;
  * **translation** (exact): the result at a pixel does not depend on where the pixel sits in a workgroup tile, a wave strip or a halo --
    every output element accumulates its 9 * Cin products in the same order -- so with a zero frame around the data (nothing wraps)
    conv(shift(x)) == shift(conv(x)) bit for bit, for shifts that move every pixel to another tile position / another tile.  A wrong
    tap, a dropped halo column or an edge mask in ONE big-tile variant -- which the full-size oracle comparison's bf16 tolerance would
    hide -- breaks this at every tile boundary;
  * **translation of the weight gradient** (exact on small-integer data, where every partial sum is an integer): the split-K sum over a
    shifted pixel set equals the unshifted one.

The shapes select the production variants (<64,4,4>, <128,4,8>, <128,2,8>, <64,2,4>, wgrad16 <64,2> / <128,1>, wgrad_bf16): asserted by
name for the two top levels.  Runs through the C ABI; about a minute on the GPU.
"""
import os
import zlib

import pytest
import torch

pytestmark = pytest.mark.gpu

B = int(os.environ.get("MAU_TEST_FULL_B", "32"))
S = 256
LAYERS = [("conv0_0.conv2", 64, 64, S), ("conv1_0.conv1", 64, 128, S // 2), ("conv1_0.conv2", 128, 128, S // 2),
          ("conv2_0.conv1", 128, 256, S // 4), ("conv2_0.conv2", 256, 256, S // 4), ("conv3_0.conv1", 256, 512, S // 8),
          ("conv3_0.conv2", 512, 512, S // 8), ("conv4_0.conv1", 576, 1024, S // 16), ("conv4_0.conv2", 1024, 1024, S // 16),
          ("conv3_1.conv1", 1536, 512, S // 8), ("conv2_1.conv1", 768, 256, S // 4), ("conv1_1.conv1", 384, 128, S // 2),
          ("conv0_1.conv1", 192, 64, S)]


@pytest.fixture(scope="module")
def env():
    import mau_amd
    from mau_amd import functional as F_
    from mau_amd import _lib
    assert torch.cuda.is_available()
    _lib.check(_lib.lib.mau_device_check(), "mau_device_check")
    if B >= 8:
        assert _lib.conv3x3_variant(_lib.MAU_BF16, B, 256, 256, 64)[:3] == (32, 4, 64) and _lib.conv3x3_variant(_lib.MAU_BF16, B, 128, 128, 128)[:3] == (32, 8, 128)
    return F_, _lib


def _framed(t, f=3):
    """zero frame of f pixels: a shift by < f pixels moves nothing across the image border"""
    t[:, :f] = 0
    t[:, -f:] = 0
    t[:, :, :f] = 0
    t[:, :, -f:] = 0
    return t


def _conv(F_, _lib, x, C0, wpk, Cout, slab=None):
    N, H, W, _ = x.shape
    y = torch.empty((N, H, W, F_.pad8(Cout)), dtype=x.dtype, device="cuda")
    _lib.call("mau_conv3x3_fwd", x.data_ptr(), x.shape[-1], C0, None, None, 0, wpk.data_ptr(), None, None, None, y.data_ptr(), y.shape[-1], Cout,
              slab.data_ptr() if slab is not None else None, _lib.MAU_BF16, N, H, W, torch.cuda.current_stream().cuda_stream)
    return y


@pytest.mark.parametrize("name,cin,cout,h", LAYERS)
def test_forward_and_data_gradient_scale_and_translate_exactly(env, name, cin, cout, h):
    F_, _lib = env
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()) % 1000)
    dt = torch.bfloat16
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
    wf, wd = F_.pack_conv_weights(w, _lib.MAU_BF16, forward=True, dgrad=True)
    for what, C_in, C_out, pack in (("forward", cin, cout, wf), ("data gradient", cout, cin, wd)):
        x = _framed(torch.randn(B, h, h, C_in, device="cuda", generator=g).to(dt))
        rows = _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, B, h, h, C_out)
        cpad = (C_out + 63) // 64 * 64
        slab1 = torch.zeros(rows, 2 * cpad, device="cuda") if what == "forward" else None
        slab2 = torch.zeros(rows, 2 * cpad, device="cuda") if what == "forward" else None
        y1 = _conv(F_, _lib, x, C_in, pack, C_out, slab1)
        # scaling by 2: exact
        y2 = _conv(F_, _lib, x * 2, C_in, pack, C_out, slab2)
        assert torch.isfinite(y1.float()).all()
        assert torch.equal(y2, y1 * 2), (name, what, "conv(2x) != 2 conv(x)")
        if slab1 is not None:
            assert torch.equal(slab2[:, :cpad], slab1[:, :cpad] * 2) and torch.equal(slab2[:, cpad:], slab1[:, cpad:] * 4), (name, "statistics")
        # translation: every pixel lands on another position of its tile, most on another wave strip or tile
        for dy_, dx_ in ((1, 1), (2, 0)) if h >= 32 else ((1, 1),):
            ys = _conv(F_, _lib, torch.roll(x, (dy_, dx_), dims=(1, 2)), C_in, pack, C_out)
            assert torch.equal(ys, torch.roll(y1, (dy_, dx_), dims=(1, 2))), (name, what, "translation", dy_, dx_)
        del x, y1, y2, ys
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name,cin,cout,h", LAYERS)
def test_weight_gradient_scales_and_translates_exactly(env, name, cin, cout, h):
    F_, _lib = env
    lib, call = _lib.lib, _lib.call
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()) % 1000 + 1)
    dt, code = torch.bfloat16, _lib.MAU_BF16
    st = torch.cuda.current_stream().cuda_stream
    n = min(B, 8) if h >= 128 else B                      # (integer partial sums must stay below 2^24: 8 x 256 x 256 x 4 = 2^21)

    def wgrad(x, dy):
        acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, x.shape[0], h, h, cout, cin), dtype=torch.float32, device="cuda")
        call("mau_conv3x3_wgrad", x.data_ptr(), x.shape[-1], cin, None, None, 0, dy.data_ptr(), dy.shape[-1], cout, acc.data_ptr(), code, x.shape[0], h, h, st)
        dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device="cuda")
        call("mau_conv3x3_unpack_wgrad", acc.data_ptr(), lib.mau_conv3x3_wgrad_splits(code, x.shape[0], h, h, cout, cin), dw.data_ptr(), cout, cin, st)
        return dw

    # real-valued data at the bench's batch: scaling is exact whatever the split-K partition
    x = torch.randn(B, h, h, cin, device="cuda", generator=g).to(dt)
    dy = torch.randn(B, h, h, cout, device="cuda", generator=g).to(dt)
    d1 = wgrad(x, dy)
    assert torch.isfinite(d1).all() and float(d1.abs().max()) > 0
    assert torch.equal(wgrad(x, dy * 2), d1 * 2) and torch.equal(wgrad(x * 2, dy), d1 * 2), (name, "scaling")
    del x, dy, d1
    # small integers inside a zero frame: every partial sum is an integer, so the sum over a shifted pixel set is the same number
    xi = _framed(torch.randint(-2, 3, (n, h, h, cin), device="cuda", generator=g).to(dt))
    dyi = _framed(torch.randint(-2, 3, (n, h, h, cout), device="cuda", generator=g).to(dt))
    di = wgrad(xi, dyi)
    assert float(di.abs().max()) < 2 ** 24
    for dy_, dx_ in ((1, 1), (2, 0)) if h >= 32 else ((1, 1),):
        ds = wgrad(torch.roll(xi, (dy_, dx_), dims=(1, 2)), torch.roll(dyi, (dy_, dx_), dims=(1, 2)))
        assert torch.equal(ds, di), (name, "translation", dy_, dx_)
    torch.cuda.empty_cache()
