"""cpc_gemm_tn (split-in-kernel weight-gradient product) at the shapes the context network and the predictors give it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpc2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for (M, N, R, what) in ((768, 256, 16512, "GRU dW, H=256"), (3072, 256, 16384, "predictor dW, H=256"), (1536, 512, 16512, "GRU dW, H=512"),
                        (6144, 512, 16384, "predictor dW, H=512"), (2048, 256, 16384, "FFN dW1"), (256, 2048, 16384, "FFN dW2")):
    A = torch.randn(R, M, device=dev) * (torch.rand(R, M, device=dev) > 0.3)
    B = torch.randn(R, N, device=dev)
    C = torch.empty(M, N, device=dev)
    nb = lib.cpc_gemm_tn_scratch_bytes(M, N, R)
    sc = torch.empty(nb, dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.cpc_gemm_tn(_lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(C), N, M, N, R, _lib.ptr(sc), nb, _lib.stream_ptr(dev)), "tn")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{what:22s} M {M:5d} N {N:4d} R {R}: {dt*1e6:7.1f} us  {2.0*M*N*R/dt/1e12:6.1f} TFLOP/s  scratch {nb/1e6:.0f} MB")
