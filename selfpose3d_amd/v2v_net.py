"""V2V 3D encoder-decoder as a plain PyTorch-ROCm module (MIOpen convolutions).

Architecture and ``state_dict`` key names follow the reference's
/root/reference/lib/models/v2v_net.py:10-144 so its checkpoints load unchanged
(``front_layers.0.block.0.weight``, ``encoder_decoder.encoder_res1.res_branch.0.weight`` ...):
7^3 conv(Cin->16) -> Res(16->32) -> [pool2, Res(32->64)] -> [pool2, Res(64->128)] -> Res(128)
-> Res(128) -> ConvT2(128->64) (+skip Res(64)) -> Res(64) -> ConvT2(64->32) (+skip Res(32))
-> 1^3 conv(32->Cout).  north_star keeps this stack on the framework's conv kernels; the
hand-written HIP work of this repo is the unprojection feeding it.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _bn(c):
    return nn.BatchNorm3d(c)


class PadCinConv3d(nn.Conv3d):
    """Conv3d whose input-channel count is rounded up to a multiple of 4 AT RUN TIME (zero input
    channels x zero weight slices: same math, parameters and state_dict unchanged).  MIOpen's
    7^3 kernels for Cin = 15 run 3.2x slower than for Cin = 16 on gfx950 (4.84 ms vs 1.51 ms at
    (4,15,80,80,20), tools/bench_cin.py); callers may also hand in an already padded tensor."""

    def forward(self, x):
        cin = self.in_channels
        have = x.shape[1]
        target = have if have > cin else ((cin + 3) // 4 * 4 if cin >= 3 else cin)
        if target == cin:
            return super().forward(x)
        w = F.pad(self.weight, (0, 0, 0, 0, 0, 0, 0, target - cin))
        if have < target:
            x = F.pad(x, (0, 0, 0, 0, 0, 0, 0, target - have))
        return F.conv3d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)


class ConvBnRelu3d(nn.Module):
    """`.block` = Conv3d -> BN -> ReLU (v2v_net.py:10-20)"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.block = nn.Sequential(PadCinConv3d(cin, cout, k, stride=1, padding=(k - 1) // 2), _bn(cout), nn.ReLU(True))

    def forward(self, x):
        return self.block(x)


class Residual3d(nn.Module):
    """`.res_branch` (conv-bn-relu-conv-bn) + `.skip_con` (identity or 1^3 conv-bn) (v2v_net.py:23-45)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.res_branch = nn.Sequential(
            nn.Conv3d(cin, cout, 3, 1, 1), _bn(cout), nn.ReLU(True), nn.Conv3d(cout, cout, 3, 1, 1), _bn(cout))
        self.skip_con = nn.Sequential() if cin == cout else nn.Sequential(nn.Conv3d(cin, cout, 1, 1, 0), _bn(cout))

    def forward(self, x):
        return F.relu(self.res_branch(x) + self.skip_con(x), True)


class Up2x3d(nn.Module):
    """`.block` = ConvTranspose3d(k2,s2) -> BN -> ReLU (v2v_net.py:57-69)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(nn.ConvTranspose3d(cin, cout, 2, stride=2, padding=0, output_padding=0), _bn(cout),
                                   nn.ReLU(True))

    def forward(self, x):
        return self.block(x)


class _EncDec(nn.Module):
    """two-level U-shaped core (v2v_net.py:72-110); attribute names = checkpoint keys"""

    def __init__(self):
        super().__init__()
        self.encoder_res1 = Residual3d(32, 64)
        self.encoder_res2 = Residual3d(64, 128)
        self.mid_res = Residual3d(128, 128)
        self.decoder_res2 = Residual3d(128, 128)
        self.decoder_upsample2 = Up2x3d(128, 64)
        self.decoder_res1 = Residual3d(64, 64)
        self.decoder_upsample1 = Up2x3d(64, 32)
        self.skip_res1 = Residual3d(32, 32)
        self.skip_res2 = Residual3d(64, 64)

    def forward(self, x):
        s1 = self.skip_res1(x)
        x = self.encoder_res1(F.max_pool3d(x, 2, 2))
        s2 = self.skip_res2(x)
        x = self.encoder_res2(F.max_pool3d(x, 2, 2))
        x = self.decoder_res2(self.mid_res(x))
        x = self.decoder_upsample2(x) + s2
        x = self.decoder_upsample1(self.decoder_res1(x)) + s1
        return x


class V2VNet(nn.Module):
    def __init__(self, input_channels: int, output_channels: int):
        super().__init__()
        self.front_layers = nn.Sequential(ConvBnRelu3d(input_channels, 16, 7), Residual3d(16, 32))
        self.encoder_decoder = _EncDec()
        self.output_layer = nn.Conv3d(32, output_channels, 1, 1, 0)
        self.reset_parameters()

    def reset_parameters(self):
        # N(0, 1e-3) weights, zero bias for every (transposed) conv (v2v_net.py:135-144)
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.normal_(m.weight, 0.0, 0.001)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.output_layer(self.encoder_decoder(self.front_layers(x)))
