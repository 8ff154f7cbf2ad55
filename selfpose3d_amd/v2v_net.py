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
    # nn.BatchNorm3d (same parameters / buffers / state_dict keys) that can keep the statistics of GROUPS of samples apart
    # while a GroupSpec is attached in train mode: all candidate slots of the training pose net in one call (grouped_bn.py)
    from .grouped_bn import GroupedBatchNorm3d
    return GroupedBatchNorm3d(c)


def _grouped(bn, training: bool) -> bool:
    return training and getattr(bn, "groups", None) is not None


def fft_len(n: int) -> int:
    """smallest even m >= n whose prime factors are <= 13 (rocFFT radices) and that has at most ONE distinct odd prime
    factor: mixed odd radices are slow in rocFFT (72 beats 70 = 2*5*7 and 88 = 8*11 beats 90 = 2*9*5 by more than
    their size ratio, tools/exp_fftconv.py)"""
    m = n + (n & 1)
    while True:
        r, odd = m, 0
        for q in (2, 3, 5, 7, 11, 13):
            if q > 2 and r % q == 0:
                odd += 1
            while r % q == 0:
                r //= q
        if r == 1 and odd <= 1:
            return m
        m += 2


class _FreqConv3d(torch.autograd.Function):
    """Stride-1 'same' Conv3d (odd cubic kernel) in the frequency domain, forward AND backward, for the 7x7x7 opening
    conv of the V2V nets (v2v_net.py:113-117) on the GPU: zero-padded rFFT (rocFFT through torch.fft) -> channel
    contraction (sp3d_freq_contract_ex) -> irFFT.  With xp = x shifted by p = k//2 into an S-sized zero volume:
        y  = crop_[0,X)  irfft( sum_c  rfft(xp)[b,c] * conj(rfft(w)[o,c]) )
        dx = crop_[p,p+X) irfft( sum_o  rfft(gy)[b,o] * rfft(w)[o,c] )
        dw = crop_[0,k)  irfft( sum_b  conj(rfft(gy)[b,o]) * rfft(xp)[b,c] )
    (cross-correlation theorem; no wrap-around because S >= X + k - 1 per axis)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        from . import _lib
        B, C, X, Y, Z = x.shape
        O, k = int(w.shape[0]), int(w.shape[2])
        p = k // 2
        S = (fft_len(X + k - 1), fft_len(Y + k - 1), fft_len(Z + k - 1))
        xp = x.new_zeros((B, C) + S)
        xp[:, :, p:p + X, p:p + Y, p:p + Z] = x
        Wf = torch.fft.rfftn(w.float(), s=S, dim=(2, 3, 4))
        Yf = _lib.freq_contract_ex(torch.fft.rfftn(xp, dim=(2, 3, 4)), Wf, "fwd")
        y = torch.fft.irfftn(Yf, s=S, dim=(2, 3, 4))[:, :, :X, :Y, :Z]
        if bias is not None:
            y = y + bias.view(1, O, 1, 1, 1)
        cl = x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous()
        ctx.save_for_backward(x, w)
        ctx.geom = (S, p, bias is not None, cl)
        return y.contiguous(memory_format=torch.channels_last_3d if cl else torch.contiguous_format)

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        x, w = ctx.saved_tensors
        S, p, has_bias, cl = ctx.geom
        B, C, X, Y, Z = x.shape
        O, k = int(w.shape[0]), int(w.shape[2])
        gp = gy.new_zeros((B, O) + S)
        gp[:, :, :X, :Y, :Z] = gy
        Gf = torch.fft.rfftn(gp, dim=(2, 3, 4))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            Wf = torch.fft.rfftn(w.float(), s=S, dim=(2, 3, 4))
            gx = torch.fft.irfftn(_lib.freq_contract_ex(Gf, Wf, "dx"), s=S, dim=(2, 3, 4))[:, :, p:p + X, p:p + Y, p:p + Z]
            gx = gx.contiguous(memory_format=torch.channels_last_3d if cl else torch.contiguous_format)
        if ctx.needs_input_grad[1]:
            xp = x.new_zeros((B, C) + S)
            xp[:, :, p:p + X, p:p + Y, p:p + Z] = x
            Xf = torch.fft.rfftn(xp, dim=(2, 3, 4))
            gw = torch.fft.irfftn(_lib.freq_contract_ex(Gf, Xf, "dw"), s=S, dim=(2, 3, 4))[:, :, :k, :k, :k]
            gw = gw.contiguous().to(w.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3, 4))
        return gx, gw, gb


class PadCinConv3d(nn.Conv3d):
    """Conv3d whose input-channel count is rounded up to a multiple of 4 AT RUN TIME (zero input
    channels x zero weight slices: same math, parameters and state_dict unchanged).  MIOpen's
    7^3 kernels for Cin = 15 run 3.2x slower than for Cin = 16 on gfx950 (4.84 ms vs 1.51 ms at
    (4,15,80,80,20), tools/bench_cin.py); callers may also hand in an already padded tensor."""

    # 7^3 kernels on the GPU run through _FreqConv3d.  The weight spectrum (O*C volumes) has to be recomputed on every
    # call while the weights train - and once more, plus the O*C inverse transforms of the weight gradient, in the
    # backward - so under autograd it only pays for batches of at least `freq_domain_min_batch` cubes
    # (full-size training step, batch 2 per GPU: 102 ms direct vs 140 ms in the frequency domain, measured).
    freq_domain = True
    freq_domain_min_batch = 8

    def forward(self, x):
        cin = self.in_channels
        have = x.shape[1]
        if self.freq_domain and x.is_cuda and x.dtype == torch.float32 and self.kernel_size == (7, 7, 7) and \
                self.stride == (1, 1, 1) and self.padding == (3, 3, 3) and self.dilation == (1, 1, 1) and \
                self.groups == 1 and have >= cin and \
                (not torch.is_grad_enabled() or x.shape[0] >= self.freq_domain_min_batch):
            return _FreqConv3d.apply(x[:, :cin] if have > cin else x, self.weight, self.bias)
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
        if _grouped(self.block[1], self.training):          # BatchNorm + ReLU in one pass (the ReLU mask is recomputed
            return self.block[1].grouped_forward(self.block[0](x), relu=True)    # from the input in the backward)
        return self.block(x)


class Residual3d(nn.Module):
    """`.res_branch` (conv-bn-relu-conv-bn) + `.skip_con` (identity or 1^3 conv-bn) (v2v_net.py:23-45)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.res_branch = nn.Sequential(
            nn.Conv3d(cin, cout, 3, 1, 1), _bn(cout), nn.ReLU(True), nn.Conv3d(cout, cout, 3, 1, 1), _bn(cout))
        self.skip_con = nn.Sequential() if cin == cout else nn.Sequential(nn.Conv3d(cin, cout, 1, 1, 0), _bn(cout))

    def forward(self, x):
        rb = self.res_branch
        if _grouped(rb[1], self.training):          # BatchNorm + ReLU, and the block's tail BatchNorm + add + ReLU, one pass each
            h = rb[3](rb[1].grouped_forward(rb[0](x), relu=True))
            return rb[4].grouped_forward(h, relu=True, residual=self.skip_con(x))
        return F.relu(self.res_branch(x) + self.skip_con(x), True)


class Up2x3d(nn.Module):
    """`.block` = ConvTranspose3d(k2,s2) -> BN -> ReLU (v2v_net.py:57-69)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(nn.ConvTranspose3d(cin, cout, 2, stride=2, padding=0, output_padding=0), _bn(cout),
                                   nn.ReLU(True))

    def forward(self, x):
        if _grouped(self.block[1], self.training):
            return self.block[1].grouped_forward(self.block[0](x), relu=True)
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


class TiledZSpectrum:
    """What the fused root-grid unprojection hands to the opening conv INSTEAD of cubes (ProjectLayer.get_voxel_zspectrum):
    the z-spectrum of the (B, cin, X, Y, Z) cubes, (B, cin, SZ//2+1, X/4, Y/4, 16) complex64 in 4 x 4 tiles.  Quacks like the
    cubes where the plan looks (shape / device / dtype)."""

    def __init__(self, spec: torch.Tensor, X: int, Y: int, Z: int, SZ: int):
        self.spec, self.X, self.Y, self.Z, self.SZ = spec, int(X), int(Y), int(Z), int(SZ)
        self.shape = (int(spec.shape[0]), int(spec.shape[1]), self.X, self.Y, self.Z)
        self.device, self.is_cuda, self.dtype = spec.device, spec.is_cuda, torch.float32


class _FoldedV2V:
    """Inference execution plan of a V2VNet: every BatchNorm3d (running statistics) is folded into the
    preceding (transposed) conv's weights, so a layer is one MIOpen conv (no bias) + ONE fused
    in-place epilogue (sp3d_channel_shift_act: shift [+ residual] [+ ReLU]) instead of conv + bias +
    BatchNorm + ReLU (+ add) kernels.  Same math as v2v_net.py:10-110 up to fp32 rounding of the
    folded weights.  Rebuilt whenever any parameter / buffer was replaced, moved or written through autograd-visible ops
    ((data_ptr, _version) of EVERY tensor), when the memory format of the 3x3x3 weights changes, after ``train()`` and
    after ``load_state_dict``.  Writes through ``.data`` leave no trace in tensor metadata: call
    ``V2VNet.invalidate_plan()`` after such weight surgery."""

    def __init__(self, net: "V2VNet"):
        self.net = net
        self.key = None
        self.t = {}

    @staticmethod
    def _key(net):
        ps = list(net.parameters()) + list(net.buffers())
        w3 = net.front_layers[1].res_branch[0].weight          # a 3x3x3 weight: its strides tell the memory format
        # (a 1x1x1 weight is "contiguous" in both formats)
        return (tuple((p.data_ptr(), p._version) for p in ps), str(w3.device),
                bool(w3.is_contiguous(memory_format=torch.channels_last_3d) and not w3.is_contiguous()))

    @staticmethod
    def _fold(conv, bn, transposed=False):
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shape = (1, -1, 1, 1, 1) if transposed else (-1, 1, 1, 1, 1)
        w = conv.weight * s.view(shape)
        fmt = torch.channels_last_3d if conv.weight.is_contiguous(memory_format=torch.channels_last_3d) and \
            not conv.weight.is_contiguous() else torch.contiguous_format
        b = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        return w.contiguous(memory_format=fmt), ((b - bn.running_mean) * s + bn.bias).contiguous()

    def _build(self):
        n, t = self.net, {}
        t["front"] = self._fold(n.front_layers[0].block[0], n.front_layers[0].block[1])
        def res(name, blk):
            from . import _lib
            w1, s1 = self._fold(blk.res_branch[0], blk.res_branch[1])
            w2, s2 = self._fold(blk.res_branch[3], blk.res_branch[4])
            # Winograd-domain weights for the wide (low-resolution) layers, see _conv3
            wino = lambda w: w.is_cuda and (w.shape[1] >= 64 or (w.shape[0] == 32 and w.shape[1] in (16, 32)) or
                                            (w.shape[0] == 64 and w.shape[1] == 32))
            u1 = _lib.wino_weights(w1) if wino(w1) else None
            u2 = _lib.wino_weights(w2) if wino(w2) else None
            # full-resolution layers (16 | 32 -> 32): direct convolution on the bf16 matrix pipe (three exact pieces)
            for wc, u in ((w1, u1), (w2, u2)):
                if u is not None and wc.shape[0] == 32 and wc.shape[1] in (16, 32):
                    u._sp3d_direct = _lib.conv_weights_split(wc)
            for u in (u1, u2):                 # layers of the fused kernels: weights split for the bf16 matrix pipe
                if u is not None and u.shape[2] == 32 and u.shape[1] in (16, 32):
                    u._sp3d_split = _lib.wino_weights_split(u)
                elif u is not None and u.shape[2] == 64 and u.shape[1] in (32, 64):
                    u._sp3d_split = _lib.wino_weights_split(u, 16)
            if len(blk.skip_con) > 0:
                ws, ss = self._fold(blk.skip_con[0], blk.skip_con[1])
                t[name] = (w1, s1, w2, (s2 + ss).contiguous(), ws, u1, u2)
            else:
                t[name] = (w1, s1, w2, s2, None, u1, u2)
        res("front_res", n.front_layers[1])
        ed = n.encoder_decoder
        for name in ("encoder_res1", "encoder_res2", "mid_res", "decoder_res2", "decoder_res1", "skip_res1", "skip_res2"):
            res(name, getattr(ed, name))
        for name in ("decoder_upsample2", "decoder_upsample1"):
            blk = getattr(ed, name).block
            wT, sT = self._fold(blk[0], blk[1], transposed=True)
            # (Cin, Cout, 2,2,2) -> GEMM form (Cin, 8*Cout), columns ordered (i,j,k,o): see _up2x
            wg = wT.permute(0, 2, 3, 4, 1).reshape(wT.shape[0], 8 * wT.shape[1]).contiguous()
            t[name] = (wT, sT, wg)
        self.t = t

    def _conv3(self, x, w, u, shift, mode, residual=None):
        """3x3x3 conv + fused epilogue.  Wide layers on small grids (the 1/4- and 1/2-resolution blocks: a GEMM with
        few rows, where MIOpen's implicit-GEMM kernels drop to ~60 TFLOP/s) go through Winograd F(2x2x2,3x3x3):
        HIP input/output transforms around one batched rocBLAS GEMM, 2.1x faster at (4,128,20,20,5) and 1.2x at
        (4,64,40,40,10) (tools/exp_wino.py); the rule below keeps the transformed tensor (64*T*C floats) within
        reach of the Infinity Cache, beyond which the transforms eat the gain."""
        from . import _lib
        if u is not None and self.net.winograd and x.is_contiguous(memory_format=torch.channels_last_3d) \
                and not x.is_contiguous():
            B, C, X, Y, Z = x.shape
            T = B * ((X + 1) // 2) * ((Y + 1) // 2) * ((Z + 1) // 2)
            u3 = getattr(u, "_sp3d_split", None) if getattr(self.net, "wino_split", True) else None
            if u3 is not None and u.shape[2] == 64:
                # half-resolution layers in one launch: 80 us instead of 115 at (4,64,40,40,10), 44 instead of 78
                # (MIOpen) at (4,32->64,40,40,10)
                return _lib.wino_fused_conv3d_(x, u, shift, mode, residual, u3)
            if C >= 128 or (C >= 64 and 64 * T * C * 4 <= 160e6):
                return _lib.wino_conv3d_(x, u, shift, mode, residual)
            w3 = getattr(u, "_sp3d_direct", None) if getattr(self.net, "wino_split", True) and getattr(self.net, "direct_conv", True) else None
            if w3 is not None and C in (16, 32) and u.shape[2] == 32:
                # implicit GEMM with exact three-piece bf16 splits: no Winograd transforms (which made the fused kernel
                # VALU-bound); 155 vs 163 us (ReLU), 159 vs 182 us (residual) at (4,32,80,80,20)
                return _lib.conv3_split_(x, w3, shift, mode, residual)
            if C in (16, 32) and u.shape[2] == 32:
                # full-resolution layers: the transformed tensor would be hundreds of MB, so everything (transforms,
                # v_mfma_f32_32x32x2_f32 products, epilogue) happens in one kernel: 225 us instead of 345 us at
                # (4,32,80,80,20), 0.99 instead of 1.34 ms for eight 64^3 pose cubes
                u3 = getattr(u, "_sp3d_split", None) if getattr(self.net, "wino_split", True) else None
                return _lib.wino_fused_conv3d_(x, u, shift, mode, residual, u3)
        return _lib.channel_shift_act_(F.conv3d(x, w, None, 1, 1), shift, mode, residual)

    def _res(self, x, name):
        w1, s1, w2, s2, ws, u1, u2 = self.t[name]
        h = self._conv3(x, w1, u1, s1, 1)
        r = x if ws is None else self._conv1(x, ws)
        return self._conv3(h, w2, u2, s2, 2, r)

    # -- library GEMM selection (OPT-IN) ---------------------------------------------------------------------------
    # The plan's GEMMs (batched Winograd products, transposed-conv and 1x1x1 GEMMs) are plain library calls; which
    # rocBLAS / hipBLASLt solution the library's heuristic picks matters: the 64 x (8000 x 64 x 64) batched product of a
    # half-resolution layer runs 85 us with the default solution and 47 us with the best one (tools/exp_blas.sh).
    # PyTorch's TunableOp can do that selection, but it is PROCESS-WIDE state and its choice comes from timing, so a
    # library forward must not switch it on behind the caller's back (round-2 advisor finding).  It is therefore off
    # unless the caller asks: ``V2VNet.tune_gemms(True)`` (what bench.py does, and says so in its JSON line) or
    # SP3D_TUNE_GEMM=1.  When on: the FIRST eager forward of each input shape runs with tuning enabled (a few ms per
    # GEMM shape, never during graph capture); later forwards only look the selection up.  TunableOp's enabled /
    # tuning flags are restored to the caller's values when the forward returns, its results file is neither renamed
    # nor written by this code: set PYTORCH_TUNABLEOP_FILENAME yourself to persist selections, and load the same
    # file on every rank (PYTORCH_TUNABLEOP_TUNING=0) for run-to-run / rank-to-rank identical kernels.
    _tuned_shapes: "set[tuple]" = set()
    _tune_gemms = None                 # None: follow SP3D_TUNE_GEMM (default off); True / False: explicit

    @staticmethod
    def _tunable():
        import os
        on = _FoldedV2V._tune_gemms
        if on is None:
            on = os.environ.get("SP3D_TUNE_GEMM", "0") not in ("0", "")
        return getattr(torch.cuda, "tunable", None) if on else None

    def _run_tuned(self, x):
        tun = self._tunable()
        if tun is None or not x.is_cuda:
            return self._run(x)
        capturing = torch.cuda.is_current_stream_capturing()
        shape_key = (tuple(x.shape), str(x.device), bool(self.net.winograd), bool(self.net.fft_front))
        was_on, was_tuning = tun.is_enabled(), tun.tuning_is_enabled()
        try:
            tun.enable(True)
            first = shape_key not in _FoldedV2V._tuned_shapes and not capturing
            tun.tuning_enable(bool(first))
            if first:
                tun.set_max_tuning_duration(30)
            out = self._run(x)
            if first:
                _FoldedV2V._tuned_shapes.add(shape_key)
            return out
        finally:                       # the caller's process-wide settings come back; the selections stay in TunableOp's
            tun.tuning_enable(was_tuning)      # in-memory table and are found again by the next forward of this plan
            tun.enable(was_on)

    @torch.no_grad()
    def run(self, x):
        return self._run_tuned(x)

    def _run(self, x):
        from . import _lib
        key = self._key(self.net)
        if key != self.key:
            self._build()
            self.key = key
        w0, s0 = self.t["front"]
        cin = w0.shape[1]
        have = x.shape[1]
        if isinstance(x, TiledZSpectrum):
            return self._tail(self._front_zspectrum(x, w0, s0))
        if self.net.fft_front and w0.shape[2] == 7 and have >= cin:
            x = self._front_fft(x, w0, s0)
            return self._tail(x)
        target = have if have > cin else ((cin + 3) // 4 * 4 if cin >= 3 else cin)
        if target != cin:
            w0 = F.pad(w0, (0, 0, 0, 0, 0, 0, 0, target - cin))
        if have < target:
            x = F.pad(x, (0, 0, 0, 0, 0, 0, 0, target - have))
        x = _lib.channel_shift_act_(F.conv3d(x, w0, None, 1, 3), s0, 1)
        return self._tail(x)

    def _zdft_ok(self, x, w0, S):
        """the direct z-DFT form of the opening conv applies: channels-last 16-channel cubes of a (Z, SZ) the kernels are
        built for, 16 output channels, channels-last conv stack behind it"""
        from . import _lib
        if not (getattr(self.net, "zdft", True) and x.dim() == 5 and x.is_cuda and x.dtype == torch.float32):
            return False
        w1 = self.t["front_res"][0]
        return bool((int(x.shape[4]), int(S[2]), int(x.shape[1])) in _lib.ZDFT_SHAPES and int(w0.shape[0]) == 16
                    and int(w0.shape[1]) <= 16 and self._is_cl(w1) and x.permute(0, 2, 3, 4, 1).is_contiguous())

    _fft_len = staticmethod(fft_len)

    @classmethod
    def _fft_shape(cls, X, Y, Z, k):
        """padded transform lengths: >= n + k//2 zeros behind the signal, rocFFT-friendly factors, and a z length that
        is a multiple of 4 so that every row of the buffer starts 16-byte aligned"""
        sz = cls._fft_len(Z + k - 1)
        while sz % 4:
            sz = cls._fft_len(sz + 2)
        return (cls._fft_len(X + k - 1), cls._fft_len(Y + k - 1), sz)

    def _fft_buffer(self, B, cin, S, device):
        """zero-padded input buffer for B volumes: ONE buffer per (cin, S, device), grown when a larger batch shows
        up and sliced otherwise (the padding is never written, so any prefix of it is a valid padded buffer)"""
        bkey = ("xpad", cin, S, str(device))
        buf = self.t.get(bkey)
        if buf is None or buf.shape[0] < B:
            buf = self.t[bkey] = torch.zeros((B, cin) + S, dtype=torch.float32, device=device)
        return buf[:B]

    def fft_input_view(self, B, X, Y, Z, device):
        """(B,cin,X,Y,Z) view of the zero-padded input buffer of the frequency-domain opening conv: a producer
        (ProjectLayer.get_voxel(out=...)) that fills it saves the pad/copy pass; `run` recognises the view."""
        key = self._key(self.net)
        if key != self.key:
            self._build()
            self.key = key
        w0 = self.t["front"][0]
        cin, k = int(w0.shape[1]), int(w0.shape[2])
        if not (self.net.fft_front and k == 7):
            return None
        S = self._fft_shape(X, Y, Z, k)
        view = self._fft_buffer(B, cin, S, device)[:, :, :X, :Y, :Z]
        view._sp3d_fft_shape = S
        return view

    @staticmethod
    def tag_fft_view(view, S):
        view._sp3d_fft_shape = S
        return view

    def _is_padded_view(self, x, cin, S):
        """is x the [:X,:Y,:Z] corner of whole samples of the zero-padded buffer this plan owns for (cin, S)?"""
        buf = self.t.get(("xpad", cin, S, str(x.device)))
        vol = S[0] * S[1] * S[2]
        if buf is None or x.dim() != 5 or x.shape[1] != cin or x.dtype != buf.dtype or \
                tuple(x.stride()) != (cin * vol, vol, S[1] * S[2], S[2], 1):
            return False
        off = x.data_ptr() - buf.data_ptr()
        per = cin * vol * buf.element_size()
        return off >= 0 and off % per == 0 and off // per + x.shape[0] <= buf.shape[0]

    def _weights_z(self, w0, S):
        """the opening conv's weight spectrum with kz as the slowest frequency index (what the direct z-DFT form contracts with)"""
        wkey, zkey = ("Wf", S), ("Wz", S)
        if zkey not in self.t:
            if wkey not in self.t:
                k = int(w0.shape[2])
                wp = torch.zeros(tuple(w0.shape[:2]) + S, dtype=torch.float32, device=w0.device)
                wp[:, :, :k, :k, :k] = w0.float()
                wp = torch.roll(wp, shifts=(-(k // 2),) * 3, dims=(2, 3, 4))
                self.t[wkey] = (torch.conj(torch.fft.rfftn(wp, dim=(2, 3, 4))).resolve_conj() /
                                float(S[0] * S[1] * S[2])).contiguous()
            self.t[zkey] = self.t[wkey].permute(0, 1, 4, 2, 3).contiguous()
            if not any(k[0] == "xpad" for k in self.t if isinstance(k, tuple)):
                del self.t[wkey]                       # nobody asked for the planar buffer: keep one spectrum only
        return self.t[zkey]

    def _weights_ty(self, w0, S):
        """(T, tw) of _lib.freq_contract_ty: the opening conv's taps transformed along z and x only (float64 on the host
        side of the plan build, stored fp32), as (G_p, S_1, D_1, ..., S_p, D_p) per (row = (kz, kx), o, c), and the
        (cos, sin)(2 pi ky u / SY) table.  W^[o,c,kz,kx,ky] = conj(rfftn(taps centred on the origin)) / N is what the kernel
        rebuilds from them per bin (csrc/sp3d_fftconv.hip)."""
        key = ("Wty", S)
        if key not in self.t:
            import math
            k = int(w0.shape[2])
            p = k // 2
            SX, SY, SZ = S
            w = w0.detach().double().cpu()                                       # (O, C, tx, ty, tz)
            t = torch.arange(k, dtype=torch.float64) - p
            ex = torch.exp(2j * math.pi * torch.arange(SX, dtype=torch.float64)[:, None] * t[None, :] / SX)            # (kx, tx)
            ez = torch.exp(2j * math.pi * torch.arange(SZ // 2 + 1, dtype=torch.float64)[:, None] * t[None, :] / SZ)   # (kz, tz)
            G = torch.einsum("ocxyz,kx,mz->mkocy", w.to(torch.complex128), ex, ez) / float(SX * SY * SZ)            # (kz,kx,O,C,ty)
            cols = [G[..., p]]
            for u in range(1, p + 1):
                cols += [G[..., p + u] + G[..., p - u], G[..., p + u] - G[..., p - u]]
            T = torch.view_as_real(torch.stack(cols, -1))                        # (kz,kx,O,C,1+2p,2)
            T = T.reshape(G.shape[0] * G.shape[1], G.shape[2], G.shape[3], 2 * (1 + 2 * p)).float().contiguous()
            ang = 2.0 * math.pi * torch.arange(SY, dtype=torch.float64)[:, None] * torch.arange(1, p + 1, dtype=torch.float64)[None, :] / SY
            tw = torch.stack([torch.cos(ang), torch.sin(ang)], -1).float().contiguous()
            self.t[key] = (T.to(w0.device), tw.to(w0.device))
        return self.t[key]

    def _contract(self, Xs, w0, S):
        """channel contraction of the opening conv on the kz-slowest spectrum: with the weight spectrum's y transform rebuilt
        per bin (17.7 MB of table instead of 223 MB of spectrum on the root grid) when the kernel covers the shape"""
        from . import _lib
        import os
        # measured (profiles/r06_contract_ty.md): 12.6x fewer weight bytes, but no faster in the step (1.4512 vs 1.4507 ms) - the
        # rebuilt spectrum costs the VALU what the stream cost the HBM.  Opt-in: V2VNet.contract_ty = True / SP3D_CONTRACT_TY=1.
        on = getattr(self.net, "contract_ty", os.environ.get("SP3D_CONTRACT_TY", "0") not in ("0", ""))
        if on and int(w0.shape[2]) == 7 and int(w0.shape[1]) <= 16 and 4 * S[1] <= 384:
            T, tw = self._weights_ty(w0, S)
            return _lib.freq_contract_ty(Xs, T, tw)
        return _lib.freq_contract(Xs, self._weights_z(w0, S))

    def _front_zspectrum(self, z: "TiledZSpectrum", w0, s0):
        """opening conv when the unprojection already delivered the z-spectrum (round 6): x,y plane transforms (un-tiling
        on load) -> contraction -> inverse x,y -> inverse z-DFT + shift + ReLU.  Same kernels and bits as the cubes path of
        _front_fft minus zdft_fwd_cl_kernel (and minus the cubes' trip through HBM)."""
        from . import _lib
        k = int(w0.shape[2])
        S = self._fft_shape(z.X, z.Y, z.Z, k)
        if S[2] != z.SZ or S[0] != 88 or S[1] != 88 or int(w0.shape[1]) != z.shape[1] or int(w0.shape[0]) != 16:
            raise _lib.Sp3dError(f"TiledZSpectrum of {z.shape} (SZ {z.SZ}) does not fit this net's opening conv (FFT shape {S})")
        Xs = _lib.cfft2d_88_tiled(z.spec, z.X, z.Y)
        Ys = _lib.cfft2d_(self._contract(Xs, w0, S), True, rows_out=z.X)
        return _lib.zdft_inv_cl(Ys, z.X, z.Y, z.Z, S[2], s0, True)

    def _front_fft(self, x, w0, s0):
        """the 7x7x7 opening conv in the frequency domain: zero-padded rFFT (rocFFT via torch.fft) ->
        sp3d_freq_contract -> irFFT -> crop + shift + ReLU.  3.3x (80x80x20, B=4) to 4.7x (64^3) faster than the
        direct fp32 convolution and ~10x closer to the float64 result (tools/exp_fftconv.py)."""
        from . import _lib
        B, _, X, Y, Z = x.shape
        cin, k = int(w0.shape[1]), int(w0.shape[2])
        S = self._fft_shape(X, Y, Z, k)
        wkey = ("Wf", S)
        if wkey not in self.t and not (("Wz", S) in self.t and self._zdft_ok(x, w0, S)):
            # the signal sits at the ORIGIN of the padded buffer (so a producer can write 16-byte aligned rows into it,
            # fft_input_view), hence the kernel is centred on the origin: taps -p..p wrap around to S-p..S-1
            wp = torch.zeros(tuple(w0.shape[:2]) + S, dtype=torch.float32, device=w0.device)
            wp[:, :, :k, :k, :k] = w0.float()
            wp = torch.roll(wp, shifts=(-(k // 2),) * 3, dims=(2, 3, 4))
            # correlation (conj); the inverse transform's 1/N is folded in here, so irfftn runs unnormalised
            self.t[wkey] = (torch.conj(torch.fft.rfftn(wp, dim=(2, 3, 4))).resolve_conj() /
                            float(S[0] * S[1] * S[2])).contiguous()
        if self._zdft_ok(x, w0, S):
            # root grid: direct z-DFTs around dense 2-D complex transforms; the spectrum's slowest frequency index is kz,
            # the weight spectrum is stored in the same order.  x is the unprojection's channels-last result.
            zkey = ("Wz", S)
            if zkey not in self.t:
                self.t[zkey] = self.t[wkey].permute(0, 1, 4, 2, 3).contiguous()
                if not any(k[0] == "xpad" for k in self.t if isinstance(k, tuple)):
                    del self.t[wkey]                       # nobody asked for the planar buffer: keep one spectrum only
            Xs = _lib.cfft2d_(_lib.zdft_fwd_cl(x, cin, S), False, rows_in=X)          # rows x >= X are zero padding
            Ys = _lib.cfft2d_(self._contract(Xs, w0, S), True, rows_out=X)              # ... and not read on the way back
            return _lib.zdft_inv_cl(Ys, X, Y, Z, S[2], s0, True)
        if self._is_padded_view(x, cin, S):
            # x IS the signal corner of this plan's zero-padded buffer (fft_input_view / input_chunk_views): no pad/copy
            # pass.  Recognised by address and strides, not by a Python attribute: a tensor that went through an
            # autograd Function under no_grad comes back as a NEW Python object for the same memory (advisor, round 2)
            buf = x.as_strided((B, cin) + S, x.stride(), x.storage_offset())
        else:
            buf = self._fft_buffer(B, cin, S, x.device)
            buf[:, :, :X, :Y, :Z].copy_(x[:, :cin])                             # borders stay zero across calls
        # cached hipFFT plans behind the C ABI: same rocFFT kernels (bit-identical spectra) as torch.fft.rfftn / irfftn,
        # minus the defensive clones torch makes around every real transform on ROCm (3 x ~55 MB per root-net step)
        Yf = _lib.freq_contract(_lib.rfft3d(buf), self.t[wkey])
        y = _lib.irfft3d_(Yf, S[2])                                  # unnormalised (1/N is in Wf); Yf is scratch
        w1 = self.t["front_res"][0]             # a 3x3x3 weight tells the layout the conv stack runs in
        cl = w1.is_contiguous(memory_format=torch.channels_last_3d) and not w1.is_contiguous()
        if cl and y.is_contiguous() and y.shape[1] % 4 == 0 and Z % 4 == 0 and 4 * Z * (y.shape[1] + 4) * 4 <= 65536:
            return _lib.crop_shift_act_cl(y, X, Y, Z, s0, True)     # crop + layout change + epilogue in one pass
        y = y[:, :, :X, :Y, :Z]
        y = y.contiguous(memory_format=torch.channels_last_3d if cl else torch.contiguous_format)
        return _lib.channel_shift_act_(y, s0, 1)

    def _tail(self, x):
        from . import _lib
        x = self._res(x, "front_res")
        skip1 = self._res(x, "skip_res1")
        x = self._res(self._pool(x), "encoder_res1")
        skip2 = self._res(x, "skip_res2")
        x = self._res(self._pool(x), "encoder_res2")
        x = self._res(self._res(x, "mid_res"), "decoder_res2")
        x = self._up2x(x, "decoder_upsample2", skip2)
        x = self._res(x, "decoder_res1")
        o = self.net.output_layer
        wT, sT, wg = self.t["decoder_upsample1"]
        if self.net.winograd and self._is_cl(x) and wg.is_cuda and wT.shape[1] == 32 and x.dtype == torch.float32:
            # the last up-sampling layer's only consumer is the 1x1x1 output conv: one kernel, no 32-channel tensor
            return _lib.upsample2x_head_(x, wg, sT, skip1, o.weight, o.bias)
        x = self._up2x(x, "decoder_upsample1", skip1)
        return self._conv1(x, o.weight, o.bias)

    def _pool(self, x):
        """MaxPool3d(2,2): own channels-last kernel (torch's also computes the arg-max indices: 2.4x the time)"""
        from . import _lib
        if self._is_cl(x) and x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and \
                all(int(v) % 2 == 0 for v in x.shape[2:]):
            return _lib.maxpool2x(x)
        return F.max_pool3d(x, 2, 2)

    @staticmethod
    def _is_cl(x):
        return x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous()

    def _up2x(self, x, name, skip):
        """ConvTranspose3d(2, stride 2) + BN + ReLU + skip: no overlapping taps, so on channels-last activations it is
        one GEMM + a scatter with the epilogue (sp3d_upsample2x_scatter) instead of MIOpen's backward-data kernel"""
        from . import _lib
        wT, sT, wg = self.t[name]
        if self.net.winograd and self._is_cl(x) and wg.is_cuda and wT.shape[1] % 4 == 0:
            return _lib.upsample2x_(x, wg, sT, skip)
        return _lib.channel_shift_act_(F.conv_transpose3d(x, wT, None, 2), sT, 3, skip)

    def _conv1(self, x, w, bias=None):
        """1x1x1 conv: on channels-last activations a plain (voxels, Cin) x (Cin, Cout) GEMM on the same memory"""
        if self.net.winograd and self._is_cl(x) and x.is_cuda:
            B, C, X, Y, Z = x.shape
            O = w.shape[0]
            x2 = x.permute(0, 2, 3, 4, 1).reshape(-1, C)
            y = torch.matmul(x2, w.reshape(O, C).t()) if bias is None else torch.addmm(bias, x2, w.reshape(O, C).t())
            return y.view(B, X, Y, Z, O).permute(0, 4, 1, 2, 3)
        return F.conv3d(x, w, bias, 1, 0)


class V2VNet(nn.Module):
    def __init__(self, input_channels: int, output_channels: int):
        super().__init__()
        self.front_layers = nn.Sequential(ConvBnRelu3d(input_channels, 16, 7), Residual3d(16, 32))
        self.encoder_decoder = _EncDec()
        self.output_layer = nn.Conv3d(32, output_channels, 1, 1, 0)
        self.fused_inference = True      # eval + no_grad + GPU: BatchNorm-folded plan with fused epilogues
        self.fft_front = True            # ... whose 7x7x7 opening conv runs in the frequency domain (rocFFT)
        self.winograd = True             # ... and whose wide low-resolution 3x3x3 convs run as Winograd F(2,3)
        self.zdft = True                 # ... root grid: direct z-DFTs + dense 2-D transforms instead of the 3-D real plans
        self.wino_split = True           # ... fused Winograd layers: exact 3-piece bf16 splits on the bf16 matrix pipe
        self.direct_conv = True          # ... full-resolution 3x3x3 layers: direct (implicit GEMM) split convolution
        self._plan = None
        self.reset_parameters()

    def reset_parameters(self):
        # N(0, 1e-3) weights, zero bias for every (transposed) conv (v2v_net.py:135-144)
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.normal_(m.weight, 0.0, 0.001)
                nn.init.zeros_(m.bias)

    def invalidate_plan(self):
        """drop the folded inference plan (BatchNorm-folded / Winograd / frequency-domain weights); it is rebuilt from
        the current parameters on the next eval forward"""
        self._plan = None
        return self

    def train(self, mode: bool = True):
        # the folded plan is dropped when the module ENTERS training (weights are about to change); model.eval() on a
        # module that already is in eval mode - every validation pass calls it - keeps the plan (and its padded
        # buffers).  Weight edits while in eval mode are still caught by the plan's (data_ptr, version) key.
        if mode and not self.training:
            self._plan = None
        return super().train(mode)

    @staticmethod
    def tune_gemms(on: bool = True):
        """opt in (or out) of library-GEMM selection through PyTorch's TunableOp for the inference plan's GEMMs
        (process-wide TunableOp state is touched only inside the plan's forward and restored afterwards; see
        _FoldedV2V._run_tuned).  Default: off, or SP3D_TUNE_GEMM=1."""
        _FoldedV2V._tune_gemms = None if on is None else bool(on)

    def _load_from_state_dict(self, *args, **kwargs):
        self._plan = None
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x):
        if isinstance(x, TiledZSpectrum):
            if self._plan is None:
                self._plan = _FoldedV2V(self)
            return self._plan.run(x)
        if self.fused_inference and not self.training and not torch.is_grad_enabled() and x.is_cuda \
                and x.dtype == torch.float32 and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0 and x.shape[4] % 4 == 0:
            if self._plan is None:
                self._plan = _FoldedV2V(self)
            return self._plan.run(x)
        return self.output_layer(self.encoder_decoder(self.front_layers(x)))

    def input_view(self, B, X, Y, Z, device):
        """where the next inference forward wants its (B,Cin,X,Y,Z) input written (a view of the FFT opening conv's
        padded buffer), or None when any tensor will do"""
        if not (self.wants_planar_input() and torch.device(device).type == "cuda" and X % 4 == 0 and Y % 4 == 0 and Z % 4 == 0):
            return None
        if self._plan is None:
            self._plan = _FoldedV2V(self)
        return self._plan.fft_input_view(B, X, Y, Z, device)

    def input_chunk_views(self, P, chunk, X, Y, Z, device):
        """for a caller that produces P inputs at once and feeds them in chunks of `chunk` (tail rounded up to a power
        of two): ONE zero-padded buffer for all of them -> (view of the first P inputs to write into, list of per-chunk
        views to pass to forward()).  None when the next forward does not take the FFT opening conv."""
        first = self.input_view(1, X, Y, Z, device)
        if first is None:
            return None
        S = first._sp3d_fft_shape
        cin = first.shape[1]
        sizes = []
        s0 = 0
        while s0 < P:
            n = min(chunk, P - s0)
            sizes.append((s0, n, 1 << (n - 1).bit_length()))
            s0 += n
        total = sizes[-1][0] + sizes[-1][2]
        buf = self._plan._fft_buffer(total, cin, S, device)
        whole = buf[:P, :, :X, :Y, :Z]
        chunks = [(n, _FoldedV2V.tag_fft_view(buf[a:a + m, :, :X, :Y, :Z], S)) for a, n, m in sizes]
        return whole, chunks

    def wants_channels_last_cubes(self, X, Y, Z) -> bool:
        """True when the next inference forward's opening conv reads channels-last 16-channel cubes directly (direct
        z-DFT form, root grid): the caller should ask the unprojection for its channels-last result, not fill input_view"""
        from . import _lib
        if not (self.wants_planar_input() and getattr(self, "zdft", True) and self.front_layers[0].block[0].kernel_size == (7, 7, 7)):
            return False
        w0, w1 = self.front_layers[0].block[0].weight, self.front_layers[1].res_branch[0].weight
        k = int(w0.shape[2])
        S = _FoldedV2V._fft_shape(X, Y, Z, k)
        cl = w1.is_contiguous(memory_format=torch.channels_last_3d) and not w1.is_contiguous()
        return bool(w0.is_cuda and cl and (int(Z), int(S[2]), 16) in _lib.ZDFT_SHAPES and int(w0.shape[0]) == 16
                    and int(w0.shape[1]) <= 16)

    def wants_zspectrum(self, X, Y, Z, cin: int):
        """SZ when the next inference forward can start from the cubes' z-spectrum (``TiledZSpectrum``: the unprojection
        fused with the opening conv's z pass, root grid), else None.  ``fuse_zdft`` (default on; SP3D_FUSE_ZDFT=0 turns it
        off for A/B) and the conditions of the direct z-DFT form + 88 x 88 planes + a float32 inference plan."""
        import os
        if not getattr(self, "fuse_zdft", os.environ.get("SP3D_FUSE_ZDFT", "1") not in ("0", "")):
            return None
        if self.training or torch.is_grad_enabled() or not self.fused_inference or not self.wants_channels_last_cubes(X, Y, Z):
            return None
        w0 = self.front_layers[0].block[0].weight
        S = _FoldedV2V._fft_shape(X, Y, Z, int(w0.shape[2]))
        if S[0] != 88 or S[1] != 88 or X % 4 or Y % 4 or int(w0.shape[1]) != int(cin) or w0.dtype != torch.float32:
            return None
        return int(S[2])

    def wants_planar_input(self) -> bool:
        """True when the next forward will take the FFT opening conv: it reads plain (B,C,X,Y,Z) cubes with the real
        channel count, so the caller need not pad channels or emit channels-last cubes."""
        return bool(self.fused_inference and self.fft_front and not self.training and not torch.is_grad_enabled())
