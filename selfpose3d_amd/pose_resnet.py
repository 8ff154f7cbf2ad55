"""2D heat-map backbone (ResNet + 3 transposed-conv upsamplers + 1x1 head, stride 4) as a plain
PyTorch-ROCm module.  Architecture and ``state_dict`` keys follow the reference's
/root/reference/lib/models/pose_resnet.py:96-262 (``conv1 bn1 layer1..4 deconv_layers.{0,1,3,4,6,7}
final_layer``) so ``pose_resnet50_panoptic.pth.tar`` loads unchanged.  north_star keeps this network
on the framework's conv kernels; what is MI355X-specific here is ``forward_views``: all V camera
views go through the network as ONE (V*B) batch - in channels_last in eval mode, and in train mode with BatchNorm
statistics kept per view (``ViewBatchNorm2d``), so the result is the reference's per-view loop's.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

_BN_MOM = 0.1


_COEF = {}


def _view_coefficients(V: int, m: float, dtype, device) -> torch.Tensor:
    """weights of V sequential momentum-m updates of a running statistic (view 0 first), built once per device: a fresh
    torch.tensor(..., device=cuda) is a synchronous host-to-device copy in every BatchNorm layer of every step"""
    key = (V, m, dtype, str(device))
    if key not in _COEF:
        _COEF[key] = torch.tensor([m * (1.0 - m) ** (V - 1 - v) for v in range(V)], dtype=dtype, device=device)
    return _COEF[key]


from .grouped_bn import GroupedBatchNorm2d, group_spec, bn_groups     # noqa: E402


class ViewBatchNorm2d(GroupedBatchNorm2d):
    """BatchNorm2d with per-view statistics for a backbone pass that holds ALL camera views.  Two mechanisms, same numbers:
    * a ``GroupSpec`` attached (``grouped_bn.bn_groups``; round 5): the grouped channels-last HIP kernels
      (``sp3d_gbn_forward`` / ``_backward``) with group_of[n] = n % V - what ``PoseResNet.forward_views`` uses on the GPU when
      the backbone's weights are channels_last (the format its convolutions train fastest in: 43.3 against 51.1 ms forward +
      backward for ten 512x960 images, tools/probe/backbone_format_probe.py), with the ReLU behind it fused;
    * ``views = V > 1`` (round 4; plain NCHW tensors, any device): see below.
    BatchNorm2d that, in TRAIN mode with ``views = V > 1``, normalises an (B*V, C, H, W) batch stacked sample-major
    (image n = b*V + v) with the statistics of EACH VIEW's B images separately - what the reference's per-view loop
    (`lib/models/multi_person_posenet.py:44-47`: one backbone call per camera) computes - in one pass: the contiguous batch
    viewed as (B, V*C, H, W) is a plain BatchNorm over V*C channels.  The running statistics receive the V sequential
    momentum updates of the loop (view 0 first) in closed form.  Same parameters, buffers and state_dict keys as BatchNorm2d;
    with ``views == 1`` or in eval mode it IS BatchNorm2d."""
    views = 1

    def forward(self, x):
        V = self.views
        if self.groups is not None and self.training:
            return self.grouped_forward(x, False)
        if V <= 1 or not self.training:
            return nn.BatchNorm2d.forward(self, x)
        N, C, H, W = x.shape
        if N % V or self.momentum is None or not self.track_running_stats:
            raise ValueError("ViewBatchNorm2d: batch must hold V views of every sample; momentum must be a number")
        x = x.contiguous()
        mean = x.new_zeros(V * C)
        var = x.new_ones(V * C)
        y = F.batch_norm(x.view(N // V, V * C, H, W), mean, var, self.weight.repeat(V), self.bias.repeat(V), True, 1.0, self.eps)
        with torch.no_grad():       # momentum 1.0 left the batch mean / unbiased variance of every (view, channel) in the buffers
            m = float(self.momentum)
            coef = _view_coefficients(V, m, mean.dtype, mean.device)
            keep = (1.0 - m) ** V
            self.running_mean.mul_(keep).add_(coef @ mean.view(V, C))
            self.running_var.mul_(keep).add_(coef @ var.view(V, C))
            self.num_batches_tracked += V
        # a plain (autograd) view: the in-place ReLU behind it then costs a CopySlices node, but nothing depends on whether a
        # backend's batch_norm backward reads its saved output (round-4 advice; the fast path is the channels_last one above)
        return y.view(N, C, H, W)


def _bn2(c):
    return ViewBatchNorm2d(c, momentum=_BN_MOM)


def _bn_add_relu(bn, relu, y, identity):
    """relu(bn(y) + identity), the tail of a residual block; with a GroupSpec attached in train mode ONE pass"""
    if getattr(bn, "groups", None) is not None and bn.training:
        return bn.grouped_forward(y, relu=True, residual=identity)
    return relu(bn(y) + identity)


def _bn_relu(bn, relu, x):
    """relu(bn(x)); with a GroupSpec attached in train mode BatchNorm and ReLU are ONE pass (the ReLU's mask is recomputed
    from the input in the backward, grouped_bn.py)"""
    if getattr(bn, "groups", None) is not None and bn.training:
        return bn.grouped_forward(x, relu=True)
    return relu(bn(x))


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
        y = self.conv2(_bn_relu(self.bn1, self.relu, self.conv1(x)))
        return _bn_add_relu(self.bn2, self.relu, y, x if self.downsample is None else self.downsample(x))


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
        y = _bn_relu(self.bn1, self.relu, self.conv1(x))
        y = _bn_relu(self.bn2, self.relu, self.conv2(y))
        return _bn_add_relu(self.bn3, self.relu, self.conv3(y), x if self.downsample is None else self.downsample(x))


_SPEC = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
         101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}


class PoseResNet(nn.Module):
    batch_views_in_training = True      # forward_views in train mode: one pass with per-view BatchNorm statistics (False: V calls)

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
        x = self.maxpool(_bn_relu(self.bn1, self.relu, self.conv1(x)))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        feat = x
        mods = list(self.deconv_layers)
        i = 0
        while i < len(mods):                                  # (ConvTranspose2d, BatchNorm, ReLU) triples: BatchNorm + ReLU fused when grouped
            if i + 2 < len(mods) and isinstance(mods[i + 1], ViewBatchNorm2d) and isinstance(mods[i + 2], nn.ReLU):
                feat = _bn_relu(mods[i + 1], mods[i + 2], mods[i](feat))
                i += 3
            else:
                feat = mods[i](feat)
                i += 1
        out = self.final_layer(feat) if head else None
        return (out, feat) if attn else out

    def forward_views(self, views: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """list[V] of (B,3,H,W) -> list[V] of (B,J,h,w): one (V*B)-image pass instead of V passes.  In eval mode BatchNorm uses
        running statistics and the batch is free to hold all views; in train mode ViewBatchNorm2d keeps the statistics of
        each view's B images apart, so both are the per-view loop's numbers (the loop itself remains for CPU tensors,
        channels_last weights, partly frozen BatchNorm or ``batch_views_in_training = False``).

        On the GPU the 1x1 head runs with its filter bank zero-padded to ceil4(J) outputs in channels_last, so its
        result IS the (V,B,h,w,Jp) buffer the unprojection kernel gathers from; the returned per-view tensors are
        (B,J,h,w) views of it (``project_layer.nhwc_heatmap_views``) and the re-tiling pass disappears."""
        V, B = len(views), views[0].shape[0]
        norm = nn.modules.batchnorm._BatchNorm          # BatchNorm2d, ViewBatchNorm2d, SyncBatchNorm after a conversion, ...
        if self.training and any(isinstance(m, norm) and m.training for m in self.modules()):
            # weights in channels_last make every convolution emit channels_last, which the (B, V*C, H, W) view of
            # ViewBatchNorm2d cannot address: such a backbone keeps the per-view loop (MultiPersonPoseNet.use_channels_last
            # leaves a training backbone in the plain format)
            bns = [m for m in self.modules() if isinstance(m, norm)]
            ours = all(isinstance(m, ViewBatchNorm2d) and m.training and m.track_running_stats and m.momentum is not None
                       for m in bns)
            cl_weights = not self.conv1.weight.is_contiguous()
            if (self.batch_views_in_training and V > 1 and views[0].is_cuda and ours and cl_weights and
                    views[0].dtype in (torch.float32, torch.float64)):
                # round 5: channels_last weights (what the convolutions train fastest in) + the grouped channels-last
                # BatchNorm kernels: image n = b * V + v belongs to group v; running statistics updated view 0 first
                spec = group_spec([B] * V, views[0].device, group_of=[n % V for n in range(B * V)])
                x = torch.stack(list(views), 1).flatten(0, 1).contiguous(memory_format=torch.channels_last)
                with bn_groups(self, spec):
                    y = self.forward(x)
                y = y.view(B, V, *y.shape[1:])
                return [y[:, v] for v in range(V)]
            if (not self.batch_views_in_training or V == 1 or not views[0].is_cuda or cl_weights or not ours):
                return [self.forward(v) for v in views]
            # one pass over all views with per-view BatchNorm statistics (ViewBatchNorm2d): the V x fewer, V x larger
            # kernels of the same arithmetic; gradients of the shared weights need no accumulation across calls
            for m in bns:
                m.views = V
            try:
                # image n = b * V + v, plain NCHW: the (B, V*C, H, W) view of ViewBatchNorm2d needs it (frames that arrive with
                # channels-last strides would make every convolution emit channels-last and every BatchNorm copy)
                y = self.forward(torch.stack(list(views), 1).flatten(0, 1).contiguous(memory_format=torch.contiguous_format))
            finally:
                for m in bns:
                    m.views = 1
            y = y.view(B, V, *y.shape[1:])
            return [y[:, v] for v in range(V)]
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


def set_backbone_memory_format(net: nn.Module, channels_last: bool) -> nn.Module:
    """channels_last weights for a backbone: what MIOpen's NHWC kernels want, in eval mode AND (round 5) in train mode - the
    batched training pass keeps its per-view BatchNorm statistics through the grouped channels-last kernels
    (``PoseResNet.forward_views``); a CPU backbone in train mode stays in the plain format (round 4's NCHW view trick)"""
    on_gpu = next(net.parameters()).is_cuda
    cpu_batched_training = isinstance(net, PoseResNet) and net.training and net.batch_views_in_training and not on_gpu
    return net.to(memory_format=torch.channels_last if (channels_last and not cpu_batched_training) else torch.contiguous_format)


class PoseResAttnNet(nn.Module):
    """attention head of the SSL model: a PoseResNet followed by a sigmoid (pose_resnet.py:287-300; state_dict
    prefix ``backbone.``)"""

    def __init__(self, cfg, num_layers):
        super().__init__()
        self.backbone = PoseResNet(cfg, num_layers)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return self.sigmoid(self.backbone(x))

    def forward_views(self, views: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """list[V] of (B,3,H,W) -> list[V] of (B,J,h,w): the views in one pass of the backbone (PoseResNet.forward_views:
        running statistics in eval mode, per-view BatchNorm statistics in train mode) == [self(v) for v in views]"""
        return [self.sigmoid(y) for y in self.backbone.forward_views(views)]


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
