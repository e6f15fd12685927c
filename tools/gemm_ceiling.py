"""Vendor-library ceiling on this box: fp16 GEMM through torch (hipBLASLt/rocBLAS) at the bench's GEMM shape and at a square."""
import torch, time
def run(M, N, K, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    b = torch.randn(N, K, device="cuda", dtype=torch.float16)
    for _ in range(5): c = a @ b.t()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): c = a @ b.t()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
run(1024, 1048576 - 2048, 1536)     # the bench's big stages together: queries x corpus rows
run(1024, 131072, 1536)
run(8192, 8192, 8192)
run(16384, 16384, 1536)
