"""Encoder output and parameter gradients against the fp64 oracle at a given window count (hidden 256): rel_err per tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cpc2_amd
from oracle import cpc_oracle as O, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
hidden = 256
params = synth.encoder_params(hidden, seed=5)
enc = cpc2_amd.CPCEncoder(hidden)
enc.load_state_dict({k[len("gEncoder."):]: v for k, v in params.items()})
enc = enc.to("cuda:0")
x = synth.audio_windows(n, 20480, seed=6)
p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
ref = O.encoder_forward(x.double(), p64, "gEncoder.")
gout = synth.features(tuple(ref.shape), seed=7)
(ref * gout.double()).sum().backward()
out = enc(x.to("cuda:0"))
(out * gout.to("cuda:0")).sum().backward()
def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())
print("fusion", "off" if os.environ.get("CPC_NO_NORM_FUSION") else "on", "n", n, "out", f"{rel(out.detach(), ref.detach()):.2e}")
for name, p in enc.named_parameters():
    print(f"  {name:22s} {rel(p.grad, p64['gEncoder.' + name].grad):.2e}")
