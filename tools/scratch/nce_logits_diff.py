"""Which logits differ between the streaming forward kernel and the register-gather kernel (CPC_NCE_NO_DMA=1)?
   python tools/scratch/nce_logits_diff.py b T har henc K nneg"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 7:
    import torch, cpc2_amd
    from oracle import synth
    b, T, har, henc, K, nn = (int(v) for v in sys.argv[1:7])
    crit = cpc2_amd.CPCUnsupersivedCriterion(K, har, henc, nn, rnnMode="linear", sizeInputSeq=T - K).to("cuda:0")
    crit.load_state_dict(synth.predictor_params(K, har, henc, 3))
    c = synth.features((b, T, har), 1).to("cuda:0"); z = synth.features((b, T, henc), 2, relu=True).to("cuda:0")
    crit.seed(7)
    preds, _ = crit.getPrediction(c, z, None)
    torch.save(torch.stack(preds).cpu(), sys.argv[7])
else:
    outs = []
    for i, env in enumerate(({}, {"CPC_NCE_NO_DMA": "1"})):
        o = f"/tmp/nce_logits_{i}.pt"
        subprocess.check_call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:7] + [o], env=dict(os.environ, **env))
        import torch
        outs.append(torch.load(o))
    a, r = outs                                     # [K][b][1 + nn][W]
    d = (a - r).abs()
    bad = torch.nonzero(d > 1e-5 * r.abs().max())
    print("shape", tuple(a.shape), "max diff", float(d.max()), "bad", bad.shape[0])
    import collections
    print("by candidate:", sorted(collections.Counter(bad[:, 2].tolist()).items())[:40])
    print("by t:", sorted(collections.Counter(bad[:, 3].tolist()).items())[:40])
    print("by k:", sorted(collections.Counter(bad[:, 0].tolist()).items()))
    for row in bad[:8].tolist():
        print(row, float(a[tuple(row)]), float(r[tuple(row)]))
