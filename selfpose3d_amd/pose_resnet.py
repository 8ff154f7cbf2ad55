"""2D heat-map backbone (ResNet + 3 transposed-conv upsamplers + 1x1 head, stride 4) as a plain
PyTorch-ROCm module.  Architecture and ``state_dict`` keys follow the reference's
/root/reference/lib/models/pose_resnet.py:96-262 (``conv1 bn1 layer1..4 deconv_layers.{0,1,3,4,6,7}
final_layer``) so ``pose_resnet50_panoptic.pth.tar`` loads unchanged.  north_star keeps this network
on the framework's conv kernels; what is MI355X-specific here is ``forward_views``: all V camera
views go through the network as ONE (V*B) batch in channels_last.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

_BN_MOM = 0.1


def _bn2(c):
    return nn.BatchNorm2d(c, momentum=_BN_MOM)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = _bn2(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = _bn2(planes)
        self.downsample = downsample

    def forward(self, x):
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = _bn2(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = _bn2(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn2(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


_SPEC = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
         101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}


class PoseResNet(nn.Module):
    def __init__(self, cfg, num_layers=None):
        super().__init__()
        block, depths = _SPEC[int(cfg.POSE_RESNET.NUM_LAYERS if num_layers is None else num_layers)]
        self._cin = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = _bn2(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, depths[0], 1)
        self.layer2 = self._stage(block, 128, depths[1], 2)
        self.layer3 = self._stage(block, 256, depths[2], 2)
        self.layer4 = self._stage(block, 512, depths[3], 2)
        ups = []
        with_bias = bool(cfg.POSE_RESNET.DECONV_WITH_BIAS)
        for planes, k in zip(cfg.POSE_RESNET.NUM_DECONV_FILTERS, cfg.POSE_RESNET.NUM_DECONV_KERNELS):
            pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[int(k)]
            ups += [nn.ConvTranspose2d(self._cin, int(planes), int(k), 2, pad, opad, bias=with_bias), _bn2(int(planes)),
                    nn.ReLU(inplace=True)]
            self._cin = int(planes)
        self.deconv_layers = nn.Sequential(*ups)
        fk = int(cfg.POSE_RESNET.FINAL_CONV_KERNEL)
        self.final_layer = nn.Conv2d(self._cin, int(cfg.NETWORK.NUM_JOINTS), fk, 1, 1 if fk == 3 else 0)
        self.reset_parameters()

    def _stage(self, block, planes, n, stride):
        down = None
        if stride != 1 or self._cin != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self._cin, planes * block.expansion, 1, stride, bias=False),
                                 _bn2(planes * block.expansion))
        blocks = [block(self._cin, planes, stride, down)]
        self._cin = planes * block.expansion
        blocks += [block(self._cin, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def reset_parameters(self):
        # N(0, 1e-3) for every conv / transposed conv, BN = identity (pose_resnet.py:247-260)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if isinstance(m, nn.ConvTranspose2d) and m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x, attn: bool = False, head: bool = True):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        feat = self.deconv_layers(x)
        out = self.final_layer(feat) if head else None
        return (out, feat) if attn else out

    def forward_views(self, views: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """list[V] of (B,3,H,W) -> list[V] of (B,J,h,w): one (V*B)-image pass instead of V passes.
        Identical to the per-view loop whenever BatchNorm is in eval mode (running statistics).

        On the GPU the 1x1 head runs with its filter bank zero-padded to ceil4(J) outputs in channels_last, so its
        result IS the (V,B,h,w,Jp) buffer the unprojection kernel gathers from; the returned per-view tensors are
        (B,J,h,w) views of it (``project_layer.nhwc_heatmap_views``) and the re-tiling pass disappears."""
        if self.training and any(isinstance(m, nn.BatchNorm2d) and m.training for m in self.modules()):
            return [self.forward(v) for v in views]
        V, B = len(views), views[0].shape[0]
        x = torch.cat(list(views), 0).contiguous(memory_format=torch.channels_last)
        fl = self.final_layer
        J = fl.out_channels
        if not x.is_cuda or J > 16:
            y = self.forward(x).contiguous()
            return list(y.view(V, B, *y.shape[1:]).unbind(0))
        from .project_layer import ProjectLayer, nhwc_heatmap_views
        _, feat = self.forward(x, attn=True, head=False)
        jp = ProjectLayer.jp_for(J)
        wgt, bias = fl.weight, fl.bias
        if jp != J:
            wgt = torch.cat([wgt, wgt.new_zeros((jp - J,) + tuple(wgt.shape[1:]))], 0)
            bias = None if bias is None else torch.cat([bias, bias.new_zeros(jp - J)], 0)
        y = F.conv2d(feat, wgt, bias, fl.stride, fl.padding).contiguous(memory_format=torch.channels_last)
        packed = y.permute(0, 2, 3, 1).view(V, B, y.shape[2], y.shape[3], jp)
        return nhwc_heatmap_views(packed, J)


class PoseResAttnNet(nn.Module):
    """attention head of the SSL model: a PoseResNet followed by a sigmoid (pose_resnet.py:287-300; state_dict
    prefix ``backbone.``)"""

    def __init__(self, cfg, num_layers):
        super().__init__()
        self.backbone = PoseResNet(cfg, num_layers)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return self.sigmoid(self.backbone(x))


def _load_pretrained(net, cfg, is_train):
    import os
    path = str(cfg.NETWORK.get("PRETRAINED", "") or "")
    if is_train and path and os.path.isfile(path):
        sd = torch.load(path, map_location="cpu")
        own = net.state_dict()
        sd = {k: v for k, v in sd.items() if k in own and v.shape == own[k].shape}
        net.load_state_dict(sd, strict=False)
    return net


def get_pose_attn_net(cfg, is_train: bool = True, **_):
    """factory of the attention net (pose_resnet.py:323-333): ResNet-``ATTN_NUM_LAYERS`` (default 18)"""
    net = PoseResAttnNet(cfg, int(cfg.get("ATTN_NUM_LAYERS", 18)))
    _load_pretrained(net.backbone, cfg, is_train)
    return net


def get_pose_net(cfg, is_train: bool = True, **_):
    """factory with the reference's name/signature (pose_resnet.py:274-284); loads
    cfg.NETWORK.PRETRAINED when the file exists (ImageNet/COCO initialisation)."""
    return _load_pretrained(PoseResNet(cfg), cfg, is_train)
