"""Which layer gets a page of the summed gradient wrong when two ranks share one MI355X?  (round-2 review, item 1)

    python tools/dp_page_probe.py [steps=40] [out=gpurun_out/dp_page_probe.json]

The launcher (this file without arguments that start with "rank") never touches the GPU: it starts two rank processes
plus one independent single-process training job as background load, all on cuda:0, and collects the ranks' records.

Every rank, every step, WITHOUT any device-wide synchronisation between backward and the readers:
  clone   = flat_grad.clone()                        on the compute stream (a device kernel on the same queue: the
                                                     stream's own view of the buffer -- the yardstick)
  h_event = pinned.copy_(flat_grad, non_blocking)    on ANOTHER stream that waits for an event recorded on the compute
                                                     stream (what torch's gloo backend does with device tensors)
  h_sync  = flat_grad.cpu()                          on the compute stream (round 2's "explicit staging")
then everything is synchronised and compared with clone: a mismatch names the reader, the offsets (-> parameter), and
whether the wrong values are zeros, the previous step's gradient, or something else.  After that the transport:
  all_reduce(flat_grad)  gloo on the DEVICE tensor    vs    all_reduce(clone.cpu())  gloo on a CPU tensor
(two addends: the sum is exact whatever the order, so the comparison is bitwise).
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launcher():
    steps = sys.argv[1] if len(sys.argv) > 1 else "40"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "dp_page_probe.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(2):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", str(r), port, steps, out + f".r{r}"], env=env))
    load = subprocess.Popen([sys.executable, os.path.abspath(__file__), "rankload", "0", port, steps, out + ".load"], env=env)
    rc = [p.wait() for p in procs]
    load.wait()
    recs = []
    for r in range(2):
        if os.path.exists(out + f".r{r}"):
            recs.append(json.load(open(out + f".r{r}")))
            os.remove(out + f".r{r}")
    summary = {"steps": int(steps), "rc": rc, "ranks": recs,
               "events_total": sum(len(r["events"]) for r in recs)}
    json.dump(summary, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "ranks"}))
    for r in recs:
        for e in r["events"][:12]:
            print(json.dumps(e))
    sys.exit(max(rc))


def rank_main():
    mode, rank, port, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import cpc2_amd
    from cpc2_amd.train import buildOptimizer, cpcStep
    from oracle import synth

    dev = torch.device("cuda:0")
    hidden, b, k, nneg = 64, 2, 12, 16
    mp = synth.encoder_params(hidden, 21)
    mp.update(synth.gru_params(hidden, hidden, 1, 22))
    model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(hidden), cpc2_amd.CPCAR(hidden, hidden, False, 1))
    model.load_state_dict(mp)
    crit = cpc2_amd.CPCUnsupersivedCriterion(k, hidden, hidden, nneg, rnnMode="linear", sizeInputSeq=128)
    crit.load_state_dict(synth.predictor_params(k, hidden, hidden, 23))
    model, crit = model.to(dev), crit.to(dev)
    opt = buildOptimizer(model, crit, lr=1e-3)
    names = [n for n, _ in crit.named_parameters()] + [n for n, _ in model.named_parameters()]
    bounds = list(zip(opt.offsets, names))

    def where(idx):
        name = "?"
        for off, n in bounds:
            if off <= idx:
                name = n
        return name

    label = torch.zeros(b, dtype=torch.long, device=dev)
    x = synth.audio_windows(b, 20480, 100 + rank).to(dev)
    crit.seed(1234 + rank)
    if mode == "rankload":                       # background load: an independent job on the same GPU
        t_end = time.time() + 1.5 * steps * 0.05 + 20
        n = 0
        while n < 4 * steps and time.time() < t_end:
            tot, _, _ = cpcStep(x, x, label, model, crit)
            tot.backward()
            opt.step()
            opt.zero_grad()
            n += 1
        torch.cuda.synchronize()
        return
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    side = torch.cuda.Stream(device=dev)
    n = opt.flat_grad.numel()
    pinned = torch.empty(n, dtype=torch.float32).pin_memory()
    prev = torch.zeros(n)
    events = []

    def describe(tag, step, got, ref):
        bad = torch.nonzero(got != ref).view(-1)
        if bad.numel() == 0:
            return
        lo, hi = int(bad.min()), int(bad.max())
        g, r, p = got[bad], ref[bad], prev[bad]
        events.append({"rank": rank, "step": step, "reader": tag, "count": int(bad.numel()), "first": lo, "last": hi,
                       "first_byte_mod_4096": (lo * 4) % 4096, "param_first": where(lo), "param_last": where(hi),
                       "got_is_zero": int((g == 0).sum()), "got_is_prev_step": int((g == p).sum()),
                       "max_abs_diff": float((g - r).abs().max()), "ref_absmax": float(r.abs().max()),
                       "sample_got": [float(v) for v in g[:4]], "sample_ref": [float(v) for v in r[:4]],
                       "sample_prev": [float(v) for v in p[:4]]})

    for step in range(steps):
        tot, _, _ = cpcStep(x, x, label, model, crit)
        tot.backward()
        cur = torch.cuda.current_stream(dev)
        clone = opt.flat_grad.clone()
        ev = torch.cuda.Event()
        ev.record(cur)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            pinned.copy_(opt.flat_grad, non_blocking=True)
        h_sync = opt.flat_grad.cpu()
        side.synchronize()
        h_event = pinned.clone()
        torch.cuda.synchronize()
        ref = clone.cpu()
        describe("stream-ordered .cpu()", step, h_sync, ref)
        describe("event-ordered copy on another stream", step, h_event, ref)
        describe("flat_grad after a full sync", step, opt.flat_grad.cpu(), ref)
        # transport
        host_sum = ref.clone()
        dist.all_reduce(host_sum)
        dist.all_reduce(opt.flat_grad)
        torch.cuda.synchronize()
        describe("gloo all_reduce on the device tensor (vs CPU-tensor all_reduce of the clones)", step, opt.flat_grad.cpu(), host_sum)
        opt.flat_grad.copy_(host_sum)            # continue from the right sum
        prev = ref
        opt.step(grad_scale=0.5)
        opt.zero_grad()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    json.dump({"rank": rank, "steps": steps, "events": events}, open(out, "w"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1].startswith("rank"):
        rank_main()
    else:
        launcher()
