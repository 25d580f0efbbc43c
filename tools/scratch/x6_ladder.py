"""Where the split-in-kernel NT family (gemm_nt_x6_kernel) spends its time on the short-K products of the context / predictor /
transformer layers: probe builds (tools/build_variant.sh gemm_f32.hip -DX6_PROBE, CPC2_HIP_LIB=tools/variant/libcpc2_hip.so) with
X6_DBG = 1 no epilogue, 2 no split arithmetic, 4 no MFMAs, 8 no loads after the first K step (numbers wrong, timing valid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpc2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
SHAPES = [("predictor 7424 x 3072 x 256", 7424, 3072, 256), ("dC 7424 x 256 x 3072", 7424, 256, 3072), ("predictor 8192 x 3072 x 256", 8192, 3072, 256), ("GRU input 16384 x 768 x 256", 16384, 768, 256),
          ("FFN up 16384 x 2048 x 256", 16384, 2048, 256), ("FFN down 16384 x 256 x 2048", 16384, 256, 2048),
          ("QKV 16384 x 256 x 256", 16384, 256, 256), ("large GRU input 16384 x 1536 x 512", 16384, 1536, 512)]
torch.manual_seed(0)
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev).relu_()
    B = torch.randn(N, K, device=dev) * 0.05
    C = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    row = []
    base = int(os.environ.get("X6_BASE", "0"))            # 100: the pipelined kernel (round 6), 0: the two-barrier kernel
    if base == 0:
        os.environ["CPC_GEMM_X6_OLD"] = "1"
    for dbg in (base + d for d in (0, 1, 2, 4, 8, 3, 6, 14, 15)):
        os.environ["X6_DBG"] = str(dbg)
        def run():
            _lib.check(lib.cpc_gemm_nt(_lib.ptr(A), K, _lib.ptr(B), K, _lib.ptr(C), N, _lib.ptr(bias), M, N, K, _lib.stream_ptr(dev)))
        for _ in range(3): run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): run()
        torch.cuda.synchronize()
        row.append((dbg, (time.perf_counter() - t0) / 20 * 1e6))
    fl = 2.0 * M * N * K
    print(f"{name}: " + "  ".join(f"dbg {d}: {us:.1f} us" + (f" ({fl / us / 1e6:.0f} TF)" if d % 100 == 0 else "") for d, us in row), flush=True)
