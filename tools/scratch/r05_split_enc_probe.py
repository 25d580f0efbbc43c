"""Round 5: what does running the encoder as two half-batches (past / future) cost against one call on 2b windows?
forward + backward of CPCEncoder alone, hidden 256 / 512, N = 128 once against N = 64 twice (same stream), torch events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cpc2_amd

dev = torch.device("cuda:0")
for hidden in (256, 512):
    torch.manual_seed(0)
    enc = cpc2_amd.CPCEncoder(hidden).to(dev)
    x = 0.05 * torch.randn(128, 1, 20480, device=dev)
    g128 = torch.randn(128, 128, hidden, device=dev)

    def one():
        z = enc.forward_channel_last(x)
        z.backward(g128)

    def two():
        za = enc.forward_channel_last(x[:64])
        zb = enc.forward_channel_last(x[64:])
        zb.backward(g128[64:])
        za.backward(g128[:64])

    def fwd_one():
        with torch.no_grad():
            enc.forward_channel_last(x)

    def fwd_two():
        with torch.no_grad():
            enc.forward_channel_last(x[:64]); enc.forward_channel_last(x[64:])

    for name, fn in (("1 x 128 fwd+bwd", one), ("2 x 64 fwd+bwd", two), ("1 x 128 fwd", fwd_one), ("2 x 64 fwd", fwd_two)):
        for _ in range(3):
            fn()
            for p in enc.parameters():
                p.grad = None
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
            for p in enc.parameters():
                p.grad = None
        b.record()
        torch.cuda.synchronize()
        print(f"hidden {hidden}: {name:18s} {a.elapsed_time(b) / 10:.3f} ms")
