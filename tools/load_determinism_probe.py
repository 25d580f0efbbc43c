"""Is the encoder's backward pass bitwise repeatable while ANOTHER process keeps the same GPU busy?  (round-2 review item 1: the
conv0 block of a rank's gradient differed slightly from run to run only when two processes shared the card.)

    python tools/load_determinism_probe.py [hidden=64] [steps=60] [out=gpurun_out/load_determinism.json]

One measuring process: every training step, the encoder's backward library call is enqueued twice with the same inputs and
the two sets of gradients are compared bit for bit (as tools/dp_conv0_probe.py does).  Beside it, one after the other:
nothing / an independent training loop of the same model / a bf16 matmul loop (torch) / a device-to-device copy loop (torch)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(torch, hidden):
    import cpc2_amd
    from cpc2_amd.train import buildOptimizer
    from oracle import synth
    dev = torch.device("cuda:0")
    mp = synth.encoder_params(hidden, 21)
    mp.update(synth.gru_params(hidden, hidden, 1, 22))
    model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(hidden), cpc2_amd.CPCAR(hidden, hidden, False, 1))
    model.load_state_dict(mp)
    crit = cpc2_amd.CPCUnsupersivedCriterion(12, hidden, hidden, 16, rnnMode="linear", sizeInputSeq=128)
    crit.load_state_dict(synth.predictor_params(12, hidden, hidden, 23))
    model, crit = model.to(dev), crit.to(dev)
    x = synth.audio_windows(2, 20480, 100).to(dev)
    return dev, model, crit, buildOptimizer(model, crit, lr=1e-3), x


def load_main(kind, seconds):
    sys.path.insert(0, ROOT)
    import torch
    t_end = time.time() + seconds
    if kind == "train":
        from cpc2_amd.train import cpcStep
        dev, model, crit, opt, x = build(torch, int(sys.argv[4]))
        crit.seed(5)
        label = torch.zeros(2, dtype=torch.long, device=dev)
        while time.time() < t_end:
            for _ in range(10):
                tot, _, _ = cpcStep(x, x, label, model, crit)
                tot.backward()
                opt.step()
                opt.zero_grad()
            torch.cuda.synchronize()
    elif kind in ("enc", "encfwd", "gru", "crit"):
        # one component of the training step alone, forward + backward (encfwd: forward only)
        import cpc2_amd
        from oracle import synth
        hidden = int(sys.argv[4])
        dev = torch.device("cuda:0")
        if kind in ("enc", "encfwd"):
            mod = cpc2_amd.CPCEncoder(hidden)
            mod.load_state_dict({k[len("gEncoder."):]: v for k, v in synth.encoder_params(hidden, 21).items()})
            mod = mod.to(dev)
            inp = synth.audio_windows(4, 20480, 100).to(dev)
            run = lambda: mod.forward_channel_last(inp)
        elif kind == "gru":
            mod = cpc2_amd.CPCAR(hidden, hidden, False, 1).to(dev)
            inp = synth.features((4, 128, hidden), 7, relu=True).to(dev).requires_grad_(True)
            run = lambda: mod(inp)
        else:
            mod = cpc2_amd.CPCUnsupersivedCriterion(12, hidden, hidden, 16, rnnMode="linear", sizeInputSeq=128).to(dev)
            mod.seed(3)
            c = synth.features((2, 128, hidden), 8).to(dev).requires_grad_(True)
            z = synth.features((2, 128, hidden), 9, relu=True).to(dev).requires_grad_(True)
            lab = torch.zeros(2, dtype=torch.long, device=dev)
            run = lambda: mod(c, z, lab)[0]
        while time.time() < t_end:
            for _ in range(10):
                out = run()
                if kind != "encfwd":
                    out.sum().backward()
            torch.cuda.synchronize()
    elif kind == "matmul":
        a = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16)
        b = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16)
        while time.time() < t_end:
            for _ in range(10):
                a @ b
            torch.cuda.synchronize()
    elif kind == "copy":
        a = torch.empty(1 << 28, device="cuda:0", dtype=torch.float32)
        b = torch.empty_like(a)
        while time.time() < t_end:
            for _ in range(10):
                b.copy_(a)
            torch.cuda.synchronize()


def measure_main(hidden, steps, out):
    sys.path.insert(0, ROOT)
    import torch
    from cpc2_amd import _lib, model as M
    from cpc2_amd._lib import ptr, ptr_array, stream_ptr
    from cpc2_amd.train import cpcStep
    dev, model, crit, opt, x = build(torch, hidden)
    lib = _lib.load()
    tape = []
    orig = M._EncoderFn.backward

    def twice(ctx, dz):
        ret = orig(ctx, dz)
        xx, saved, *params = ctx.saved_tensors
        n, length, hid = ctx.dims
        arena = _lib.scratch(lib.cpc_encoder_scratch_bytes(n, length, hid), xx.device)
        g1 = [g.clone() for g in ret[2:]]
        g2 = [torch.empty_like(g) for g in ret[2:]]
        dzc = dz.contiguous()
        _lib.check(lib.cpc_encoder_backward(ptr(xx), ptr_array(params), ptr(dzc), ptr(saved), ptr(arena), ptr_array(g2), n, length,
                                            hid, ctx.eps, stream_ptr(xx.device)), "second run")
        tape.append(torch.stack([(a != b).sum() for a, b in zip(g1, g2)]))      # on-stream, read at the end
        return ret

    M._EncoderFn.backward = staticmethod(twice)
    crit.seed(1234)
    label = torch.zeros(2, dtype=torch.long, device=dev)
    for _ in range(steps):
        tot, _, _ = cpcStep(x, x, label, model, crit)
        tot.backward()
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    counts = torch.stack(tape).cpu()                      # [steps][20 parameters]
    names = []
    for i in range(5):
        names += [f"conv{i}.weight", f"conv{i}.bias", f"norm{i}.weight", f"norm{i}.bias"]
    bad_steps = torch.nonzero(counts.sum(1)).view(-1).tolist()
    json.dump({"steps": steps, "steps_that_differ": bad_steps,
               "elements_by_parameter": {names[j]: int(counts[:, j].sum()) for j in range(20) if int(counts[:, j].sum())}}, open(out, "w"))


def measure_step_main(hidden, steps, out):
    """The WHOLE step twice from the same state (same parameters, same negative indices): every gradient must repeat bit for
    bit -- covers every kernel of forward + backward, not just the encoder's backward."""
    sys.path.insert(0, ROOT)
    import torch
    from cpc2_amd.train import cpcStep
    dev, model, crit, opt, x = build(torch, hidden)
    label = torch.zeros(2, dtype=torch.long, device=dev)
    names = [n for n, _ in crit.named_parameters()] + [n for n, _ in model.named_parameters()]
    bounds = list(opt.offsets) + [opt.flat_grad.numel()]
    tape, loss_tape = [], []
    for step in range(steps):
        passes, losses = [], []
        for _ in range(2):
            crit.seed(1000 + step)
            tot, ls, _ = cpcStep(x, x, label, model, crit)
            tot.backward()
            opt._gather_stray_grads()
            passes.append(opt.flat_grad.clone())
            losses.append(ls.detach().clone())
            if len(passes) == 1:
                opt.zero_grad()
        d = passes[0] != passes[1]
        tape.append(torch.stack([d[bounds[i]:bounds[i + 1]].sum() for i in range(len(names))]))
        loss_tape.append((losses[0] != losses[1]).sum())
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    counts = torch.stack(tape).cpu()
    json.dump({"steps": steps, "steps_that_differ": torch.nonzero(counts.sum(1)).view(-1).tolist(),
               "steps_whose_losses_differ": torch.nonzero(torch.stack(loss_tape).cpu()).view(-1).tolist(),
               "elements_by_parameter": {names[j]: int(counts[:, j].sum()) for j in range(len(names)) if int(counts[:, j].sum())}},
              open(out, "w"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "measure_step":
        return measure_step_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    if len(sys.argv) > 1 and sys.argv[1] == "load":
        return load_main(sys.argv[2], float(sys.argv[3]))
    if len(sys.argv) > 1 and sys.argv[1] == "measure":
        return measure_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    hidden = sys.argv[1] if len(sys.argv) > 1 else "64"
    steps = sys.argv[2] if len(sys.argv) > 2 else "60"
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "load_determinism.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    env = dict(os.environ, PYTHONPATH=ROOT)
    result = {"hidden": int(hidden)}
    kinds = os.environ.get("CPC_LOAD_KINDS", "none,train,matmul,copy,train+train").split(",")
    for kind in kinds:
        loads = []
        for k in ([] if kind == "none" else kind.split("+")):
            k, _, h = k.partition(":")                      # "train:256" = the load trains another width
            loads.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "load", k, "25", h or hidden], env=env))
        if loads:
            time.sleep(12)                                # the load's own start-up (imports, first kernels)
        tmp = out + ".part"
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), os.environ.get("CPC_MEASURE", "measure"), hidden, steps, tmp], env=env)
        for p in loads:
            p.wait()
        result[kind] = json.load(open(tmp)) if rc == 0 and os.path.exists(tmp) else {"rc": rc}
        print(kind, json.dumps(result[kind]), flush=True)
    if os.path.exists(out + ".part"):
        os.remove(out + ".part")
    json.dump(result, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
