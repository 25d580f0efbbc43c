"""Where does the conv0 block of a rank's gradient go wrong?  (round-2 review item 1; tools/dp_pair_repeat.py showed: only
gEncoder.conv0.* / batchNorm0.*, one rank at a time, relative 1e-3..1e-2, most often step 1 of the DistributedDataParallel
pair.)

    python tools/dp_conv0_probe.py [steps=4] [out=gpurun_out/dp_conv0_probe.json]

Two ranks on cuda:0, the reference's DDP arrangement over gloo (as tests/dp_job.py "ddp").  The encoder's backward is
wrapped (here, not in the package): after the library call has been enqueued, the scratch arena is cloned ON STREAM, the SAME
library call is enqueued a second time with the same inputs into spare gradient buffers, and the arena is cloned again.  Same
inputs, same stream: every byte must repeat.  After the loop the host compares the two runs region by region of the arena
(dU of layer 1 -> dY0 -> conv0's partial sums -> sums) and gradient by gradient, and recomputes dY0 from the cloned dU and the
saved backward-data operand in fp64: the first region that differs names the kernel, the rows that differ name the place."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launcher():
    """argv: [steps=4] [out] [hidden=64] [poison]   (poison: ONE process, no process group; the second run of every backward
    pass starts from register files and LDS full of NaN patterns -- tools/poison_state.hip)"""
    steps = sys.argv[1] if len(sys.argv) > 1 else "4"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "dp_conv0_probe.json")
    hidden = sys.argv[3] if len(sys.argv) > 3 else "64"
    poison = len(sys.argv) > 4 and sys.argv[4] in ("poison", "solo")      # solo: one process beside an independent training loop
    solo = len(sys.argv) > 4 and sys.argv[4] == "solo"
    os.makedirs(os.path.dirname(out), exist_ok=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    load = None
    if solo:
        import time
        load = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "load_determinism_probe.py"), "load", "train", "45", hidden], env=env)
        time.sleep(12)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", str(r), port, steps, out + f".r{r}", hidden,
                               sys.argv[4] if poison else "ddp"], env=env) for r in range(1 if poison else 2)]
    rc = [p.wait() for p in procs]
    if load is not None:
        load.wait()
    recs = []
    for r in range(len(procs)):
        if os.path.exists(out + f".r{r}"):
            recs.append(json.load(open(out + f".r{r}")))
            os.remove(out + f".r{r}")
    json.dump({"rc": rc, "ranks": recs}, open(out, "w"), indent=1)
    for rec in recs:
        for st in rec["steps"]:
            print(json.dumps(st)[:1800])
    sys.exit(max(rc))


def rank_main():
    rank, port, steps, out = int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    hidden_arg, poison, solo = int(sys.argv[6]), sys.argv[7] == "poison", sys.argv[7] == "solo"
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import cpc2_amd
    from cpc2_amd import _lib, model as M
    from cpc2_amd._lib import ptr, ptr_array, stream_ptr
    from cpc2_amd.train import buildOptimizer, cpcStep
    from oracle import synth

    dev = torch.device("cuda:0")
    H, b, k, nneg = hidden_arg, 2, 12, 16
    N, LEN = 2 * b, 20480
    mp = synth.encoder_params(H, 21)
    mp.update(synth.gru_params(H, H, 1, 22))
    model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(H), cpc2_amd.CPCAR(H, H, False, 1))
    model.load_state_dict(mp)
    crit = cpc2_amd.CPCUnsupersivedCriterion(k, H, H, nneg, rnnMode="linear", sizeInputSeq=128)
    crit.load_state_dict(synth.predictor_params(k, H, H, 23))
    model, crit = model.to(dev), crit.to(dev)
    opt = buildOptimizer(model, crit, lr=1e-3)

    # ---- the encoder's layouts (csrc/encoder.hip: enc_layout, hidden sizes without planes)
    conv = ((10, 5, 3), (8, 4, 2), (4, 2, 1), (4, 2, 1), (4, 2, 1))
    L = [LEN]
    for kk, s, p in conv:
        L.append((L[-1] + 2 * p - kk) // s + 1)
    Rv = [0] + [L[i + 1] + 2 for i in range(1, 5)]
    R = [conv[i + 1][1] * Rv[i + 1] for i in range(4)]

    class Carver:
        def __init__(self):
            self.off, self.where = 0, {}

        def take(self, name, count, size=4):
            self.off = (self.off + 255) // 256 * 256
            self.where[name] = (self.off, count * size)
            self.off += count * size

    sv, sc = Carver(), Carver()
    for i in range(4):
        sv.take(f"Y{i}", (N * R[i] + conv[i + 1][0]) * H)
    for i in range(1, 5):
        sv.take(f"Xh{i}", N * Rv[i] * H)
        sv.take(f"rstd{i}", N * Rv[i])
    sv.take("stats0", N * L[1] * 2)
    for i in range(1, 5):
        sv.take(f"Wd{i}", conv[i][0] * H * H)
    for i in range(1, 5):
        sc.take(f"Wf{i}", conv[i][0] * H * H)
    sc.take("dYa", N * L[1] * H)
    sc.take("dYb", N * L[2] * H)
    sc.take("dU", (N * Rv[1] + 2) * H)
    sc.take("part", max(768 * 13 * H, 2048 * 3 * H))
    sc.take("sums", 13 * H)
    sc.take("cs", (32 * 13 * H * 4 + 255) // 256 * 256 // 4)
    sc.take("dbg", 768 * 8)                      # head of the weight-gradient scratch (diagnostic build: load checksums)
    arena_used = (sc.off + 255) // 256 * 256
    lib = _lib.load()
    deep = H == 64                                   # (the arena analysis below knows the layout without planes only)
    assert not deep or lib.cpc_encoder_saved_bytes(N, LEN, H) == (sv.off + 255) // 256 * 256, "saved layout drifted from encoder.hip"
    if not deep:
        arena_used = 256
    poison_lib, sink = None, None
    if poison and not solo:
        import ctypes
        poison_lib = ctypes.CDLL(os.path.join(ROOT, "tools", "variant", "libpoison.so"))
        poison_lib.poison_state.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(1, dtype=torch.int32, device=dev)

    tape = []
    orig = M._EncoderFn.backward

    def deep_backward(ctx, dz):
        ret = orig(ctx, dz)
        x, saved, *params = ctx.saved_tensors
        n, length, hidden = ctx.dims
        nbytes = lib.cpc_encoder_scratch_bytes(n, length, hidden)
        arena = _lib.scratch(nbytes, x.device)
        rec = {"grads1": [g.clone() for g in ret[2:]], "arena1": arena[:arena_used].clone()}
        dzc = dz.contiguous()
        grads2 = [torch.empty_like(g) for g in ret[2:]]
        if poison_lib is not None:
            rc = poison_lib.poison_state(0x7FC0DEAD, sink.data_ptr(), stream_ptr(x.device))
            assert rc == 0, rc
        _lib.check(lib.cpc_encoder_backward(ptr(x), ptr_array(params), ptr(dzc), ptr(saved), ptr(arena), ptr_array(grads2),
                                            n, length, hidden, ctx.eps, stream_ptr(x.device)), "encoder_backward (second run)")
        rec.update(grads2=grads2, arena2=arena[:arena_used].clone())
        if deep:
            rec.update(Wd1=saved[sv.where["Wd1"][0]:sv.where["Wd1"][0] + sv.where["Wd1"][1]].clone(),
                       stats0=saved[sv.where["stats0"][0]:sv.where["stats0"][0] + sv.where["stats0"][1]].clone(),
                       p0=[p.detach().clone() for p in params[:4]], x=x.detach().clone())
        tape.append(rec)
        return ret

    M._EncoderFn.backward = staticmethod(deep_backward)

    poison = poison or solo                          # (below: "no process group")
    if poison:
        ddp_model, ddp_crit = model, crit
    else:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2")
        dist.init_process_group("gloo", rank=rank, world_size=2)
        ddp_model = DDP(model, device_ids=[0], find_unused_parameters=True)
        ddp_crit = DDP(crit, device_ids=[0], find_unused_parameters=True)
    crit.seed(1234 + rank)
    x = synth.audio_windows(b, LEN, 100 + rank).to(dev)
    label = torch.zeros(b, dtype=torch.long, device=dev)
    for _ in range(steps):
        tot, _ls, _acc = cpcStep(x, x, label, ddp_model, ddp_crit)
        tot.backward()
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    if not poison:
        dist.barrier()
        dist.destroy_process_group()

    names = ["conv0.weight", "conv0.bias", "norm0.weight", "norm0.bias"]
    for i in range(1, 5):
        names += [f"conv{i}.weight", f"conv{i}.bias", f"norm{i}.weight", f"norm{i}.bias"]

    def region(buf, name, where=sc.where):
        off, nb = where[name]
        return buf[off:off + nb].view(torch.float32)

    def rows_that_differ(a, c, width):
        bad = torch.nonzero((a != c).view(-1, width).any(dim=1)).view(-1)
        return {"rows": int(bad.numel()), "first": bad[:8].tolist(), "max_abs": float((a - c).abs().max())}

    steps_out = []
    for step, rec in enumerate(tape):
        st = {"rank": rank, "step": step, "grads_differ": {}, "arena_differs": {}}
        for nm, g1, g2 in zip(names, rec["grads1"], rec["grads2"]):
            g1, g2 = g1.cpu(), g2.cpu()
            if not torch.equal(g1, g2):
                d = (g1 - g2).abs()
                st["grads_differ"][nm] = {"count": int((g1 != g2).sum()), "max_abs": float(torch.nan_to_num(d, nan=1e30).max()),
                                          "absmax": float(g1.abs().max()), "nonfinite_in_run2": int((~torch.isfinite(g2)).sum())}
        if not deep:
            steps_out.append(st)
            continue
        a1, a2 = rec["arena1"].cpu(), rec["arena2"].cpu()
        for nm, width in (("dU", H), ("dYb", H), ("dYa", H), ("part", 13 * H), ("sums", 13 * H)):
            r1, r2 = region(a1, nm), region(a2, nm)
            if nm == "part":
                r1, r2 = r1[:768 * 13 * H], r2[:768 * 13 * H]
            if not torch.equal(r1, r2):
                st["arena_differs"][nm] = rows_that_differ(r1, r2, width)
        if os.environ.get("CPC2_HIP_LIB"):           # diagnostic build (-DCPC_C0_DBG): what each block LOADED, run 1 against run 2
            d1, d2 = region(a1, "dbg").view(torch.int32).view(768, 8)[:256], region(a2, "dbg").view(torch.int32).view(768, 8)[:256]
            part1, part2 = region(a1, "part")[:768 * 13 * H].view(768, 13 * H)[:256], region(a2, "part")[:768 * 13 * H].view(768, 13 * H)[:256]
            blocks_part = (part1 != part2).any(dim=1)
            st["loads_differ"] = {nm: {"blocks": int((d1[:, q] != d2[:, q]).sum()),
                                       "of_those_with_different_sums": int(((d1[:, q] != d2[:, q]) & blocks_part).sum())}
                                  for q, nm in enumerate(("params", "dy", "stats", "xs"))}
            st["blocks_with_different_sums"] = int(blocks_part.sum())
            for tag, d in (("run1", d1), ("run2", d2)):
                moved = d[:, 4] != 0
                st[f"hwid_{tag}"] = {"blocks_whose_staged_x_changed": int(moved.sum()), "of_those_with_different_sums": int((moved & blocks_part).sum()),
                                     "words_changed": d[moved, 4].tolist()[:12], "highest_changed_index": d[moved, 3].tolist()[:12],
                                     "which_blocks": torch.nonzero(moved).view(-1).tolist()[:12],
                                     "blocks_with_different_sums": torch.nonzero(blocks_part).view(-1).tolist()[:12],
                                     "ticks_median": int(d[:, 5].median()), "ticks_max": int(d[:, 5].max()),
                                     "ticks_of_differing_blocks": d[blocks_part, 5].tolist()[:12],
                                     "vmid_field_values": sorted(set(((d[:, 6] >> 24) & 0xF).tolist()))}
            # what a block SHOULD have loaded, from the buffers cloned on stream after the call (wrapping 32-bit sums of bit patterns)
            import numpy as np
            bits = lambda t: t.contiguous().view(torch.int32).numpy().view(np.uint32).astype(np.uint64)
            p_sum = int(sum(int(bits(p_.cpu()).sum()) for p_ in rec["p0"]) * 16) & 0xFFFFFFFF
            dy_b = bits(region(a2, "dYa")).reshape(N * L[1] // 64, 64 * H).sum(1) & 0xFFFFFFFF
            st_b = bits(rec["stats0"].cpu().view(torch.float32)).reshape(N * L[1] // 64, 128).sum(1) & 0xFFFFFFFF
            for tag, d in (("run1", d1), ("run2", d2)):
                du = d.numpy().view(np.uint32).astype(np.uint64)
                st[f"loaded_vs_expected_{tag}"] = {"params_wrong_blocks": np.nonzero(du[:, 0] != p_sum)[0].tolist()[:12],
                                                   "dy_wrong_blocks": np.nonzero(du[:, 1] != dy_b)[0].tolist()[:12],
                                                   "stats_wrong_blocks": np.nonzero(du[:, 2] != st_b)[0].tolist()[:12]}
            # all blocks load the same parameters: how many distinct checksums in each run?
            st["param_checksums_distinct"] = [int(torch.unique(d1[:, 0]).numel()), int(torch.unique(d2[:, 0]).numel())]
        # every block whose partial sums differ between the two runs: which run is wrong, and is the error made of whole
        # frames (a dropped / doubled frame, a dropped lane group)?  fp64 contributions of the tile's 64 frames
        part1 = region(a1, "part")[:768 * 13 * H].view(768, 13 * H).double()
        part2 = region(a2, "part")[:768 * 13 * H].view(768, 13 * H).double()
        blocks = torch.nonzero((part1 != part2).any(dim=1)).view(-1).tolist()
        if blocks:
            w, bb, gam, bet = [p.cpu().double().view(H, -1).squeeze(-1) if p.numel() == H else p.cpu().double().view(H, 10) for p in rec["p0"]]
            xs_all = torch.nn.functional.pad(rec["x"].cpu().double().view(N, LEN), (3, 16))
            stats = rec["stats0"].cpu().view(torch.float32).view(N, L[1], 2).double()
            dy = region(a2, "dYa").view(N, L[1], H).double()
            st["blocks"] = []
            for blk in blocks[:6]:
                n, t0 = blk // 64, (blk % 64) * 64
                contrib = torch.zeros(64, 13 * H, dtype=torch.float64)
                for slot in range(64):
                    t = t0 + slot
                    xr = xs_all[n, 5 * t:5 * t + 10]
                    acc = bb + w @ xr
                    mean, rstd = stats[n, t, 0], stats[n, t, 1]
                    xhat = (acc - mean) * rstd
                    g = dy[n, t] * ((xhat * gam + bet) > 0)
                    gx = g * gam
                    s1, s2 = gx.mean(), (gx * xhat).sum() / (H - 1)
                    du = rstd * (gx - s1 - xhat * s2)
                    contrib[slot, :10 * H] = (xr[:, None] * du[None, :]).reshape(-1)
                    contrib[slot, 10 * H:11 * H] = du
                    contrib[slot, 11 * H:12 * H] = g * xhat
                    contrib[slot, 12 * H:] = g
                total = contrib.sum(0)
                e1, e2 = part1[blk] - total, part2[blk] - total
                wrong = e1 if e1.abs().max() > e2.abs().max() else e2
                alpha = torch.linalg.lstsq(contrib.t(), wrong[:, None]).solution.view(-1)
                resid = wrong - contrib.t() @ alpha
                big = torch.nonzero(alpha.abs() > 0.05).view(-1).tolist()
                st["blocks"].append({"block": blk, "err_run1": float(e1.abs().max()), "err_run2": float(e2.abs().max()),
                                     "scale": float(total.abs().max()), "frames_involved": big,
                                     "alpha": [round(float(alpha[i]), 3) for i in big][:16],
                                     "residual_after_frames": float(resid.abs().max()),
                                     "err_by_section": {nm: float(wrong[a_:b_].abs().max()) for nm, a_, b_ in
                                                        (("dW", 0, 10 * H), ("db", 10 * H, 11 * H), ("dgamma", 11 * H, 12 * H), ("dbeta", 12 * H, 13 * H))},
                                     # where in the [13][H] block of sums: by channel (a lane owns 4 consecutive channels)
                                     "channels_over_10pct": torch.nonzero(wrong.view(13, H).abs().amax(0) > 0.1 * wrong.abs().max()).view(-1).tolist(),
                                     "rows_q_over_10pct": torch.nonzero(wrong.view(13, H).abs().amax(1) > 0.1 * wrong.abs().max()).view(-1).tolist(),
                                     "nonfinite": int((~torch.isfinite(part1[blk])).sum() + (~torch.isfinite(part2[blk])).sum())})
        # dY0 against an fp64 product of the cloned dU rows with the saved backward-data operand of layer 1
        s1, p1 = conv[1][1], conv[1][2]
        wd = rec["Wd1"].cpu().view(torch.float32).view(s1 * H, 2 * H).double()
        for tag, arena in (("run1", a1), ("run2", a2)):
            du = region(arena, "dU").view(N * Rv[1] + 2, H).double()
            a_rows = torch.cat([du[:N * Rv[1]], du[1:N * Rv[1] + 1]], dim=1)                 # [dU(m), dU(m+1)]
            o = (a_rows @ wd.t()).view(N, Rv[1], s1, H)
            ref = torch.zeros(N, L[1], H, dtype=torch.float64)
            for t_hi in range(L[2] + 1):
                for j in range(s1):
                    row = t_hi * s1 - p1 + j
                    if 0 <= row < L[1]:
                        ref[:, row] = o[:, t_hi, j]
            got = region(arena, "dYa").view(N, L[1], H).double()
            err = (got - ref).abs().amax(dim=2)                                              # per (window, row)
            tol = 1e-5 * float(ref.abs().max())
            bad = torch.nonzero(err > tol)
            st[f"dY0_vs_fp64_{tag}"] = {"bad_rows": int(bad.shape[0]), "first": bad[:8].tolist(), "max_err": float(err.max()),
                                       "scale": float(ref.abs().max())}
        steps_out.append(st)
    json.dump({"rank": rank, "steps": steps_out}, open(out, "w"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rank":
        rank_main()
    else:
        launcher()
