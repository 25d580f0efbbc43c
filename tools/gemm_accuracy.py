"""Error of cpc_gemm_nt against an fp64 product, for the bf16x6 kernel (default) and the native-f32-MFMA kernel
(CPC_GEMM_NATIVE_F32=1): python tools/gemm_accuracy.py  (spawns itself once per mode)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure():
    import torch
    from cpc2_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    out = []
    for (M, N, K, scale) in [(4096, 256, 2048, "randn"), (4096, 256, 512, "randn"), (2048, 768, 256, "wide"), (4096, 256, 1024, "relu")]:
        g = torch.Generator(device="cpu").manual_seed(M + K)
        A = torch.randn(M, K, generator=g)
        B = torch.randn(N, K, generator=g) / K ** 0.5
        if scale == "wide":      # 2^-20 .. 2^20 magnitudes: exercises the residual terms over many exponents
            A = A * torch.exp2(torch.randint(-20, 21, (M, K), generator=g).float())
        if scale == "relu":
            A = A.clamp_min(0)
        Ad, Bd = A.to(dev), B.to(dev)
        C = torch.empty(M, N, device=dev)
        _lib.check(lib.cpc_gemm_nt(_lib.ptr(Ad), K, _lib.ptr(Bd), K, _lib.ptr(C), N, None, M, N, K, _lib.stream_ptr(dev)))
        ref = A.double() @ B.double().t()
        mag = A.double().abs() @ B.double().abs().t()          # sum |a||b|: the scale rounding errors are relative to
        err = (C.cpu().double() - ref).abs()
        out.append(("nt", M, N, K, scale, float((err / mag).max()), float((err / mag).pow(2).mean().sqrt())))
    for (M, N, R, scale) in [(256, 1024, 16384, "randn"), (256, 2048, 8000, "relu")]:      # C[M,N] = A[R,M]^T B[R,N]
        g = torch.Generator(device="cpu").manual_seed(R)
        A = torch.randn(R, M, generator=g)
        B = torch.randn(R, N, generator=g)
        if scale == "relu":
            B = B.clamp_min(0)
        Ad, Bd = A.to(dev), B.to(dev)
        C = torch.empty(M, N, device=dev)
        nb = lib.cpc_gemm_tn_scratch_bytes(M, N, R)
        sc = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.cpc_gemm_tn(_lib.ptr(Ad), M, _lib.ptr(Bd), N, _lib.ptr(C), N, M, N, R, _lib.ptr(sc), nb, _lib.stream_ptr(dev)))
        ref = A.double().t() @ B.double()
        mag = A.double().abs().t() @ B.double().abs()
        err = (C.cpu().double() - ref).abs()
        out.append(("tn", M, N, R, scale, float((err / mag).max()), float((err / mag).pow(2).mean().sqrt())))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for r in measure():
            print("%-7s %s M=%d N=%d K=%d %-5s  max|err|/sum|a||b| = %.3e   rms = %.3e" % ((sys.argv[1],) + r))
    else:
        for mode, env in (("bf16x6", {}), ("f32mfma", {"CPC_GEMM_NATIVE_F32": "1"})):
            e = dict(os.environ); e.update(env)
            subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=e, check=True)
