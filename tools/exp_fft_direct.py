"""torch.fft.rfftn / irfftn vs the cached hipFFT plans behind sp3d_rfft3d / sp3d_irfft3d (no defensive clones):
equality, input preservation, and time per call at the root-net and pose-net opening-conv shapes."""
import json
import sys

import torch

sys.path.insert(0, ".")
from selfpose3d_amd import _lib  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


out = {}
for name, shape in (("root_b4", (4, 16, 88, 88, 28)), ("pose_8cubes", (8, 15, 72, 72, 72)), ("pose_4cubes", (4, 15, 72, 72, 72))):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g).cuda()
    x0 = x.clone()
    ref = torch.fft.rfftn(x, dim=(2, 3, 4))
    got = _lib.rfft3d(x)
    same_in = bool(torch.equal(x, x0))
    err_f = float((torch.view_as_real(got) - torch.view_as_real(ref)).abs().max() / torch.view_as_real(ref).abs().max())
    S = shape[2:]
    back_ref = torch.fft.irfftn(ref, s=S, dim=(2, 3, 4), norm="forward")
    back = _lib.irfft3d_(got.clone(), S[2])
    err_b = float((back - back_ref).abs().max() / back_ref.abs().max())
    n = S[0] * S[1] * S[2]
    err_rt = float((back / n - x0).abs().max())
    spec = got.clone()
    r = {
        "input_preserved": same_in, "fwd_rel_err": err_f, "inv_rel_err": err_b, "roundtrip_abs_err": err_rt,
        "torch_rfftn_us": timeit(lambda: torch.fft.rfftn(x, dim=(2, 3, 4))),
        "sp3d_rfft3d_us": timeit(lambda: _lib.rfft3d(x)),
        "torch_irfftn_us": timeit(lambda: torch.fft.irfftn(ref, s=S, dim=(2, 3, 4), norm="forward")),
        "sp3d_irfft3d_us": timeit(lambda: _lib.irfft3d_(spec, S[2])),
    }
    out[name] = r
print(json.dumps(out, indent=1))
