"""Probe of the plane-fed GEMMs (gemm_planes.hip): timing + check against an fp64 product.
   python tools/planes_probe.py [reps]      env: PROBE_N windows, PROBE_L frames, PROBE_H channels, PROBE_TAPS k, PROBE_STRIDE s,
                                                 PROBE_COLS output columns (default H), PROBE_OLD=1 also times cpc_gemm_nt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpc2_amd import _lib
_lib.LIB_PATH = os.environ.get('CPC_LIB', _lib.LIB_PATH)
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
N, H, L = int(os.environ.get("PROBE_N", 128)), int(os.environ.get("PROBE_H", 256)), int(os.environ.get("PROBE_L", 1024))
k, s = int(os.environ.get("PROBE_TAPS", 8)), int(os.environ.get("PROBE_STRIDE", 4))
cols = int(os.environ.get("PROBE_COLS", H))
sl = s.bit_length() - 1
R = s * (L + 2)                       # signal rows per window
K = k * H
M = N * L
torch.manual_seed(0)
rows = N * R + 2 * s
Y = torch.randn(rows, H, device=dev).relu_()
W = torch.randn(cols, K, device=dev) * 0.05
bias = torch.randn(cols, device=dev)
C = torch.empty(M, cols, device=dev)
st = _lib.stream_ptr(dev)
rts = rows // s
ya = (H // 16) * s * rts * 16
wa = W.numel()
Yp = torch.zeros(3 * ya, dtype=torch.int16, device=dev)
Wp = torch.zeros(3 * wa, dtype=torch.int16, device=dev)
_lib.check(lib.cpc_split_planes(_lib.ptr(Y), H, rows, H, _lib.ptr(Yp), ya, sl, rts, st))
# K order of the plane-fed GEMM: (chunk c, tap 0, s, 1, s + 1, ...)
taps = [(jj >> 1) + (jj & 1) * s for jj in range(k)] if k > 1 else [0]
Wk = W.view(cols, k, H // 16, 16)[:, taps].permute(0, 2, 1, 3).reshape(cols, K).contiguous()
_lib.check(lib.cpc_split_planes(_lib.ptr(Wk), K, cols, K, _lib.ptr(Wp), wa, 0, cols, st))
# the planes add up to the operand (weights: chunk c of row n at (c * cols + n) * 16)
recw = sum(Wp[i * wa:(i + 1) * wa].view(torch.bfloat16).float() for i in range(3)).view(K // 16, cols, 16).permute(1, 0, 2).reshape(cols, K)
print("split residual, weights:", float((recw - Wk).abs().max() / W.abs().max()))
recy = sum(Yp[i * ya:(i + 1) * ya].view(torch.bfloat16).float() for i in range(3)).view(H // 16, s, rts, 16).permute(2, 1, 0, 3).reshape(rts * s, H)
print("split residual, signal:", float((recy[:rows] - Y).abs().max() / Y.abs().max()))
def run():
    _lib.check(lib.cpc_gemm_nt_planes(_lib.ptr(Yp), ya, k.bit_length() - 1, sl, rts, L, R // s, _lib.ptr(Wp), wa,
                                      _lib.ptr(C), cols, _lib.ptr(bias), M, cols, K, st))
run(); torch.cuda.synchronize()
# check windows 0, N//2, N-1 against fp64
worst = 0.0
for n in sorted({0, N // 2, N - 1}):
    A = torch.as_strided(Y, (L, K), (s * H, 1), n * R * H).double()
    ref = A @ W.double().t() + bias.double()
    got = C[n * L:(n + 1) * L].double()
    scale = (A.abs() @ W.double().abs().t()).max()
    worst = max(worst, float((got - ref).abs().max() / scale))
print(f"max |C - fp64| / max sum|a||b| = {worst:.3e}")
flops = 2.0 * M * cols * K
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record()
torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / reps
print(f"planes nt M={M} N={cols} K={K} stride={s}: {dt*1e3:.3f} ms  {flops/dt/1e12:.1f} TFLOP/s ({flops/dt/416.7e12:.3f} of 416.7)")
if os.environ.get("PROBE_OLD"):
    Cv = torch.empty(N * (L + 2), cols, device=dev)
    def old():
        _lib.check(lib.cpc_gemm_nt(_lib.ptr(Y), s * H, _lib.ptr(W), K, _lib.ptr(Cv), cols, _lib.ptr(bias), N * (L + 2), cols, K, st))
    for _ in range(3): old()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): old()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"split-in-kernel nt (virtual rows {N*(L+2)}): {dt*1e3:.3f} ms  {flops/dt/1e12:.1f} TFLOP/s")

# ---- weight-gradient (TN) form: dW[co][j*H + ci] = sum_m dU[m + 1][co] * Y[m*s + j][ci], m over the N*(L+2) virtual rows
if os.environ.get("PROBE_TN", "1") != "0":
    Rv = L + 2
    Rr = N * Rv
    rowsd = (Rr + 2 + 63) // 32 * 32
    dU = torch.randn(rowsd, H, device=dev) * 0.1
    dU[Rr + 1:] = 0
    da = (H // 16) * rowsd * 16
    dUp = torch.zeros(3 * da, dtype=torch.int16, device=dev)
    _lib.check(lib.cpc_split_planes(_lib.ptr(dU), H, rowsd, H, _lib.ptr(dUp), da, 0, rowsd, st))
    dW = torch.empty(H, K, device=dev)
    nb = lib.cpc_gemm_tn_planes_scratch_bytes(H, K, Rr)
    sc = torch.empty(nb, dtype=torch.uint8, device=dev)
    def run_tn():
        _lib.check(lib.cpc_gemm_tn_planes(_lib.ptr(dUp), da, 0, rowsd, 1, H, _lib.ptr(Yp), ya, sl, rts, 0, H, _lib.ptr(dW), K, H, K, Rr,
                                          _lib.ptr(sc), nb, st))
    run_tn(); torch.cuda.synchronize()
    A = torch.as_strided(Y, (Rr, K), (s * H, 1), 0)
    ref = torch.zeros(H, K, dtype=torch.float64, device=dev)
    scale = torch.zeros(H, K, dtype=torch.float64, device=dev)
    for r0 in range(0, Rr, 16384):
        a = A[r0:r0 + 16384].double(); d = dU[1 + r0:1 + r0 + a.shape[0]].double()
        ref += d.t() @ a; scale += d.abs().t() @ a.abs()
    print(f"tn: max |dW - fp64| / max sum|a||b| = {float((dW.double() - ref).abs().max() / scale.max()):.3e}")
    for _ in range(3): run_tn()
    e0.record()
    for _ in range(reps): run_tn()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / reps
    flops = 2.0 * Rr * H * K
    print(f"planes tn M={H} N={K} R={Rr}: {dt*1e3:.3f} ms  {flops/dt/1e12:.1f} TFLOP/s ({flops/dt/416.7e12:.3f} of 416.7)")
    if os.environ.get("PROBE_OLD"):
        nb2 = lib.cpc_gemm_tn_scratch_bytes(H, K, Rr)
        sc2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
        dW2 = torch.empty(H, K, device=dev)
        def old_tn():
            _lib.check(lib.cpc_gemm_tn(_lib.ptr(dU[1:]), H, _lib.ptr(Y), s * H, _lib.ptr(dW2), K, H, K, Rr, _lib.ptr(sc2), nb2, st))
        for _ in range(3): old_tn()
        e0.record()
        for _ in range(reps): old_tn()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        print(f"split-in-kernel tn: {dt*1e3:.3f} ms  {flops/dt/1e12:.1f} TFLOP/s; max diff to planes {float((dW2 - dW).abs().max()):.3e}")
