"""What does the vendor GEMM (hipBLASLt through torch.matmul) reach on the plain GEMMs of the SAME shapes as the benchmark model's 3x3 / stride-1
convolutions (M = N H W pixels, N = Cout, K = 9 Cin; operands already in GEMM form: no gather, no halo, no epilogue)?  A yardstick for the
roofline fractions of the hand-written convolution kernels, not part of the product path.  usage: python tools/probes/gemm_yardstick.py"""
import torch
dev = torch.device('cuda:0')
LAYERS = [('104x64->64', 32 * 104 * 104, 64, 576), ('52x128->128', 32 * 52 * 52, 128, 1152), ('26x256->256', 32 * 26 * 26, 256, 2304),
          ('13x512->512', 32 * 13 * 13, 512, 4608), ('52x128->256', 32 * 52 * 52, 256, 1152), ('26x256->512', 32 * 26 * 26, 512, 2304),
          ('8192^3 (library sweet spot)', 8192, 8192, 8192), ('wgrad 52x128 (dW = dY^T X: 128 x 1152 over 86528 pixels)', 128, 1152, 32 * 52 * 52)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / n


for name, M, N, K in LAYERS:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = min(timed(lambda: torch.matmul(x, w.t(), out=y)) for _ in range(3))
    print('%-60s M %7d N %5d K %6d  %7.1f us  %6.0f TFLOP/s' % (name, M, N, K, t, 2.0 * M * N * K / t / 1e6), flush=True)
