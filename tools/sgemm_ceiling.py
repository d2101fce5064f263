"""What a library f32 GEMM reaches on this GPU (the practical ceiling next to the 157.3 TFLOP/s f32 MFMA peak)."""
import time, torch
torch.backends.cuda.matmul.allow_tf32 = False
for (m, n, k) in ((65536, 512, 2304), (8192, 8192, 8192), (16384, 4096, 4096)):
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda")
    for _ in range(3): (a @ b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = 10
    for _ in range(it): (a @ b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    print(f"sgemm {m}x{n}x{k}: {dt*1e3:.3f} ms  {2.0*m*n*k/dt/1e12:.1f} TFLOP/s")
