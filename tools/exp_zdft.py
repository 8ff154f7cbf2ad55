"""per-stage times of the root grid's opening conv in its direct z-DFT form (bench shape)"""
import json
import sys

import torch

sys.path.insert(0, ".")
from selfpose3d_amd import _lib  # noqa: E402


def timeit(fn, iters=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) * 1e3 / iters, 2)


B, C, X, Y, Z, S = 4, 16, 80, 80, 20, (88, 88, 28)
x = torch.randn(B, C, X, Y, Z).cuda().contiguous(memory_format=torch.channels_last_3d)
W = torch.randn(16, 15, 15, 88, 88, dtype=torch.complex64).cuda()
shift = torch.randn(16).cuda()
spec = _lib.zdft_fwd_cl(x, 15, S)
ys = _lib.freq_contract(spec, W)
out = {
    "zdft_fwd_cl_us": timeit(lambda: _lib.zdft_fwd_cl(x, 15, S)),
    "cfft2d_fwd_us": timeit(lambda: _lib.cfft2d_(spec, False, rows_in=X)),
    "cfft2d_fwd_full_rows_us": timeit(lambda: _lib.cfft2d_(spec, False)),
    "cfft2d_fwd_library_us": timeit(lambda: _lib.cfft2d_(spec, False, library=True)),
    "freq_contract_us": timeit(lambda: _lib.freq_contract(spec, W)),
    "cfft2d_inv_us": timeit(lambda: _lib.cfft2d_(ys, True, rows_out=X)),
    "cfft2d_inv_library_us": timeit(lambda: _lib.cfft2d_(ys, True, library=True)),
    "zdft_inv_cl_us": timeit(lambda: _lib.zdft_inv_cl(ys, X, Y, Z, 28, shift, True)),
}
print(json.dumps(out))
