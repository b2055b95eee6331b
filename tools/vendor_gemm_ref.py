"""Reference point, NOT part of the product: what the vendor fp32 GEMM (rocBLAS / hipBLASLt behind torch.matmul) takes for the step's
products on this GPU, beside this library's kernels (bench line classes).  python tools/vendor_gemm_ref.py"""
import torch

def t(fn, it=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3

if __name__ == "__main__":
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = "cuda"
    shapes = [("encode  [256,3707]x[3707,992]", 256, 992, 3707, False), ("decode  [256,993]x[993,3706]", 256, 3706, 993, False),
              ("dE(D)   [256,3706]x[992,3706]^T", 256, 992, 3706, True), ("decode(G) [128,993]x[993,3706]", 128, 3706, 993, False),
              ("dE(G)   [128,3706]x[992,3706]^T", 128, 992, 3706, True), ("dF      [128,992]x[3706,992]^T", 128, 3706, 992, True),
              ("gen     [128,250]x[3706,250]^T", 128, 3706, 250, True), ("gWd     [256,993]^Tx[256,3706]", 993, 3706, 256, None),
              ("scores  [6040,250]x[3706,250]^T", 6040, 3706, 250, True)]
    for name, M, N, K, bt in shapes:
        if bt is None:
            A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
            f = lambda: torch.matmul(A.t(), B)
        elif bt:
            A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
            f = lambda: torch.matmul(A, B.t())
        else:
            A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
            f = lambda: torch.matmul(A, B)
        us = t(f)
        print("%-34s %8.2f us  %6.1f TFLOP/s" % (name, us, 2.0 * M * N * K / us / 1e6), flush=True)
