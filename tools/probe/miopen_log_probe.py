"""What does MIOpen print about the kernels it picks?  One small conv2d + conv3d forward/backward in PyTorch's default
(immediate) mode with MIOPEN_LOG_LEVEL in {5,6}; stderr goes to gpurun_out/ so the format can be read (round 5, pins logging)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    x = torch.randn(2, 16, 24, 32, device="cuda", requires_grad=True)
    c = torch.nn.Conv2d(16, 32, 3, padding=1).cuda()
    c(x).sum().backward()
    x3 = torch.randn(2, 16, 16, 16, 16, device="cuda", requires_grad=True)
    c3 = torch.nn.Conv3d(16, 32, 3, padding=1).cuda()
    c3(x3).sum().backward()
    torch.cuda.synchronize()
    print("miopen version", torch.backends.cudnn.version(), "hip", torch.version.hip)
else:
    os.makedirs("gpurun_out", exist_ok=True)
    for lvl in ("5", "6"):
        env = dict(os.environ, MIOPEN_LOG_LEVEL=lvl, MIOPEN_ENABLE_LOGGING_CMD="1")
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        open(f"gpurun_out/miopen_log_level{lvl}.txt", "w").write(r.stdout[-2000:] + "\n=====\n" + r.stderr[:400000])
        print(lvl, r.returncode, len(r.stderr))
