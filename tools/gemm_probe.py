"""GEMM-only probe (conv1 forward shape by default) for rocprofv3 --pmc runs.
   python tools/gemm_probe.py [nt|tn] [reps]      env: PROBE_M rows, PROBE_TAPS k, PROBE_STRIDE s, PROBE_H channels
   (A rows overlap: lda = s*H, K = k*H; the dgrad phase GEMMs are PROBE_TAPS=2 PROBE_STRIDE=1)
   PROBE_PAD_A / PROBE_PAD_B: extra floats added to lda / ldb (channel-interleave experiments)"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpc2_amd import _lib
lib = _lib.load()
kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
N, H, L = 128, int(os.environ.get("PROBE_H", 256)), 1024
k, s = int(os.environ.get("PROBE_TAPS", 8)), int(os.environ.get("PROBE_STRIDE", 4))
Rv = L + 2
pad_a, pad_b = int(os.environ.get("PROBE_PAD_A", 0)), int(os.environ.get("PROBE_PAD_B", 0))
M = int(os.environ.get("PROBE_M", N * Rv))
torch.manual_seed(0)
if kind == "nt":
    Y = torch.randn(M * (s * H + pad_a) + k * H, device=dev)
    W = torch.randn(H, k * H + pad_b, device=dev)
    C = torch.empty(M, H, device=dev)
    bias = torch.randn(H, device=dev)
    def run():
        _lib.check(lib.cpc_gemm_nt(_lib.ptr(Y), s * H + pad_a, _lib.ptr(W), k * H + pad_b, _lib.ptr(C), H, _lib.ptr(bias), M, H, k * H, _lib.stream_ptr(dev)))
    flops = 2.0 * M * H * k * H
else:
    dU = torch.randn((M + 2) * H, device=dev)
    Y = torch.randn(M * s * H + k * H, device=dev)
    C = torch.empty(H, k * H, device=dev)
    nb = lib.cpc_gemm_tn_scratch_bytes(H, k * H, M)
    sc = torch.empty(nb, dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.cpc_gemm_tn(_lib.ptr(dU), H, _lib.ptr(Y), s * H, _lib.ptr(C), k * H, H, k * H, M, _lib.ptr(sc), nb, _lib.stream_ptr(dev)))
    flops = 2.0 * M * H * k * H
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{kind} M={M} K={k*H} lda={s*H}+{pad_a} ldb+{pad_b}: {dt*1e3:.3f} ms  {flops/dt/1e12:.1f} TFLOP/s")
