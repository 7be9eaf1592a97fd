"""Run the weight-gradient kernel of one U-Net layer REPS times (for rocprofv3 --pmc): LAYER=conv3_1.conv1 MAU_WGRAD16=0|1 python scripts/wg_one.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16
L = {"conv3_1.conv1": (1536, 512, 32), "conv2_0.conv2": (256, 256, 64), "conv0_1.conv2": (64, 64, 256), "conv1_1.conv1": (384, 128, 128), "conv0_1.conv1": (192, 64, 256)}
cin, cout, h = L[os.environ.get("LAYER", "conv3_1.conv1")]
N = 32; st = torch.cuda.current_stream().cuda_stream
x = torch.randn(N, h, h, cin, device="cuda").bfloat16(); dy = torch.randn(N, h, h, cout, device="cuda").bfloat16()
acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(MAU_BF16, N, h, h, cout, cin), device="cuda")
for _ in range(int(os.environ.get("REPS", 20))):
    call("mau_conv3x3_wgrad", x.data_ptr(), cin, cin, None, None, 0, dy.data_ptr(), cout, cout, acc.data_ptr(), MAU_BF16, N, h, h, st)
torch.cuda.synchronize()
