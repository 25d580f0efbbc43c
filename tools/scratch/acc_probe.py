import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpc2_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
def planes(x):
    rows, cols = x.shape
    plane = (cols // 16) * rows * 16
    out = torch.zeros(3 * plane, dtype=torch.int16, device=x.device)
    _lib.check(lib.cpc_split_planes(_lib.ptr(x), cols, rows, cols, _lib.ptr(out), plane, 0, rows, _lib.stream_ptr(x.device)))
    return out, plane
for data in ("relu", "randn"):
    for K in (256, 512, 1024, 2048, 4096):
        g = torch.Generator().manual_seed(K)
        M, N = 1024, 256
        a = torch.randn(M, K, generator=g)
        if data == "relu": a = a.clamp_min(0)
        b = torch.randn(N, K, generator=g) / K ** 0.5
        ad, bd = a.to(DEV), b.to(DEV)
        ref = a.double() @ b.double().t(); mag = a.double().abs() @ b.double().abs().t()
        res = {}
        for mode in (0, 1):
            prev = lib.cpc_gemm_set_mode(mode)
            c = torch.empty(M, N, device=DEV)
            _lib.check(lib.cpc_gemm_nt(_lib.ptr(ad), K, _lib.ptr(bd), K, _lib.ptr(c), N, None, M, N, K, _lib.stream_ptr(c.device)))
            lib.cpc_gemm_set_mode(prev)
            res["split" if mode == 0 else "f32"] = c.cpu().double()
        ap, pa = planes(ad); bp, pb = planes(bd)
        c = torch.empty(M, N, device=DEV)
        _lib.check(lib.cpc_gemm_nt_planes(_lib.ptr(ap), pa, 0, 0, M, 0, 0, _lib.ptr(bp), pb, _lib.ptr(c), N, None, M, N, K, _lib.stream_ptr(c.device)))
        res["planes"] = c.cpu().double()
        res["torch"] = (ad @ bd.t()).cpu().double()
        out = []
        for k, v in res.items():
            e = (v - ref).abs() / mag
            out.append(f"{k} rms {float(e.pow(2).mean().sqrt()):.2e} max {float(e.max()):.2e}")
        print(data, K, " | ".join(out), f"| |C|/mag rms {float((ref.abs()/mag).pow(2).mean().sqrt()):.3f}")
