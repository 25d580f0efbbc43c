#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec of CPC training (1.28 s windows @16 kHz, 128 negatives).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one optimisation step of the reference's trainStep (train.py:95-113) on one batch of
synthetic windows already resident in HBM: encoder on cat([past, future]) (2b windows), the context
network on the b windows whose context train.py:102 keeps (the other half's context is sliced away
there and its gradient is identically zero: same outputs, gradients and update -- the reference's own
2b-window, T-frame dataflow is measured beside it as `small_strict`; a recurrent context network also
stops after the W = T - nPredicts frames criterion.py:296 keeps), InfoNCE criterion, backward, gradient
all-reduce (N > 1), fused Adam.
Workload at every N: BASELINE.json configs[1] per GPU -- CPC-small (hiddenEncoder = hiddenGar = 256,
GRU x1, nPredicts = 12, 128 negatives, linear predictors), 64 windows of 20480 samples per GPU
(weak scaling; N = 8 is configs[2]).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Eight hardware queues per priority instead of the runtime's four (read when HIP initialises, i.e. at the first GPU call): the
# library places its own streams on queues it has TESTED to run beside the training stream's, but RCCL's stream is torch's to pick,
# and with four queues it shares the training stream's one time in four (profiles/r06_process_group_queues.md).  One RCCL rank:
# 4.87-4.98 against 4.94-5.03 ms per step, six alternating pairs (profiles/r06_ab_hw_queues_dist.txt); no effect without a process group.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# CPC_BENCH_PIN_CORES=n: run this process (and every thread it starts: the sampler's worker, RCCL's proxy and watchdog) on the first
# n cores of its affinity mask -- the CPU share a rank gets when eight ranks share one host.  Before any GPU call, in-process (not
# `taskset`: under a profiler that would be an exec behind an initialised GPU).
PINNED_CORES = None
if os.environ.get("CPC_BENCH_PIN_CORES"):
    _cores = sorted(os.sched_getaffinity(0))[:max(1, int(os.environ["CPC_BENCH_PIN_CORES"]))]
    os.sched_setaffinity(0, _cores)
    PINNED_CORES = len(_cores)

T_START = time.perf_counter()
WINDOW = 20480
SECONDS_PER_WINDOW = WINDOW / 16000.0
# MI355X_MICROARCH.md: dense bf16 MFMA peak ~2.5 PFLOP/s.  The f32 GEMMs run as six bf16 partial products per f32
# product (exact three-term operand split), so the matrix-pipe ceiling for ALGORITHMIC f32 flops is 2500 / 6;
# the f32 MFMA's own dense peak (157.3 TFLOP/s, what the round's first kernels were priced against) is below it.
BF16_MFMA_PEAK_TFLOPS = 2500.0
GEMM_PEAK_TFLOPS = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1)
FP32_MFMA_PEAK_TFLOPS = 157.3
CONFIGS = {
    "small": dict(hidden=256, layers=1, npred=12, nneg=128, ar="GRU"),
    "large": dict(hidden=512, layers=2, npred=12, nneg=256, ar="GRU"),
    "transformer": dict(hidden=256, layers=1, npred=12, nneg=128, ar="transformer"),    # BASELINE configs[3]
    # the fork's default predictor (rnnMode='transformer', criterion.py:136-143) on the GRU model
    "transformer_pred": dict(hidden=256, layers=1, npred=12, nneg=128, ar="GRU", rnn="transformer"),
    # the fork's default autoregressive network (arMode='LSTM', cpc_default_config.py; model.py:171-173)
    "lstm": dict(hidden=256, layers=1, npred=12, nneg=128, ar="LSTM"),
    # the documented training recipe (docs/training_and_eval.md:6-9): the defaults + --n-levels-gru=2 --multihead-rnn
    "recipe": dict(hidden=256, layers=2, npred=12, nneg=128, ar="LSTM", rnn="transformer", multihead=True),
    # NOT the headline arithmetic: CPC-small with the encoder's GEMMs in the opt-in three-product mode (cpc_gemm_set_mode(2):
    # a0 b0 + a0 b1 + a1 b0 of the bf16 split, 16 bits of product mantissa -- TF32, what the reference's own convolutions get on
    # its GPUs by default, keeps 10).  Reported beside the exact mode so that the cost of exactness is a measured number.
    "small_3term": dict(hidden=256, layers=1, npred=12, nneg=128, ar="GRU", gemm_mode=2),
    # NOT the headline either: CPC-small with ONE encoder / AR pass when past is future (no augmentation: dataset.py:308-321 yields
    # the same window twice, so the two halves of train.py:99's 2b-window batch are identical; SURVEY 8d allows the one-pass
    # step as a labelled extra).  Bit-identical losses and parameter updates at about half the encoder work.
    "small_dedup": dict(hidden=256, layers=1, npred=12, nneg=128, ar="GRU", dedup=True),
    # the reference's own dataflow: CPCModel.forward on all 2b windows (the context network also on the b windows whose context
    # train.py:102 drops) -- what the headline was measured on up to round 4; identical results (tests/test_gpu_parity.py)
    "small_strict": dict(hidden=256, layers=1, npred=12, nneg=128, ar="GRU", strict=True),
    # the REAL loop around the same step (dataset.py:300-325,366-408 -> train.py:95-134): WAV files on disk -> findAllSeqs ->
    # AudioBatchData (flat audio resident in HBM) -> getDataLoader(64, "samespeaker", randomOffset) -> cpc2_amd.train.trainStep
    # (its own logging at the reference's default logging_step = 1000), one epoch of >= 200 steps: measure_feeder()
    "small_feeder": dict(hidden=256, layers=1, npred=12, nneg=128, ar="GRU", feeder=True),
}
GATES = {"GRU": 3, "LSTM": 4, "RNN": 1}
CONV = ((10, 5, 3), (8, 4, 2), (4, 2, 1), (4, 2, 1), (4, 2, 1))


def encoder_lengths():
    lens = [WINDOW]
    for k, s, p in CONV:
        lens.append((lens[-1] + 2 * p - k) // s + 1)
    return lens


def planes_nt_algorithmic_flops(b, cfg, dedup=False):
    """Algorithmic FLOPs (2 per MAC) of everything that runs on gemm_planes_kernel<0, false> in ONE step, and its launches:
    conv1..4 forward and backward-data (the s phases side by side; the boundary row of backward-data is a small VALU
    kernel and is not counted).  Junk / padding rows are NOT counted."""
    h, n = cfg["hidden"], (b if dedup else 2 * b)
    lens = encoder_lengths()
    flops, launches = 0.0, 0
    for i in range(1, 5):
        k, s, _p = CONV[i]
        flops += 2 * (2.0 * n * lens[i + 1] * k * h * h)        # forward + backward-data
        launches += 2
    return flops, launches


def ar_windows(b, cfg, dedup=False):
    """Windows the context network runs on: b (cpcStep's default and dedup), 2b in the reference's own dataflow (strict)."""
    return 2 * b if (cfg.get("strict") and not dedup) else b


def gemm_nt_algorithmic_flops(b, cfg, dedup=False):
    """Algorithmic FLOPs of what still runs on the split-in-kernel gemm_nt_x6_kernel (operands split while staged): the
    context network's input projection + its dX, the predictor GEMM + its dC (hidden sizes other than 256 / 512: also the
    convolutions)."""
    h, n = cfg["hidden"], (b if dedup else 2 * b)
    lens = encoder_lengths()
    flops, launches = 0.0, 0
    if h % 256 != 0:
        f, l = planes_nt_algorithmic_flops(b, cfg, dedup)
        flops, launches = flops + f, launches + l
    t_len = lens[5]
    din = h
    n = ar_windows(b, cfg, dedup)                       # (the context network's share: encoder 2 passes, context network 1)
    # (a recurrent context network only runs the W frames the criterion reads -- criterion.py:296 -- unless the reference's own
    #  dataflow is asked for)
    t_ar = t_len if (cfg.get("strict") or cfg["ar"] == "transformer") else t_len - cfg["npred"]
    for _layer in range(cfg["layers"]):
        if cfg["ar"] == "transformer":                # QKV(3) + Wo + lin1 + lin2 + last_linear, and their 5 dX GEMMs
            flops += 2 * (2.0 * n * t_len * (5 * h * h + 2 * h * 2048))
            launches += 12
        else:
            flops += 2 * (2.0 * n * t_ar * din * GATES[cfg["ar"]] * h)    # GI and dX
            launches += 2
        din = h
    w = t_len - cfg["npred"]
    if cfg.get("multihead"):                            # one transformer, the FFN emits K branches, K x last_linear
        flops += 2 * (2.0 * b * w * (16 * h * h + 13 * h * 2048))
        launches += 12
    elif cfg.get("rnn") == "transformer":               # K one-layer transformer predictors on [b, W, H]
        flops += cfg["npred"] * 2 * (2.0 * b * w * (5 * h * h + 2 * h * 2048))
        launches += cfg["npred"] * 12
    else:
        flops += 2 * (2.0 * b * w * cfg["npred"] * h * h)   # P and dC
        launches += 2
    return flops, launches


def similarity_algorithmic_flops(b, cfg):
    """The InfoNCE similarity matmul alone (north_star's kernel target), forward: 2 K W (Nneg + 1) H per window."""
    w = encoder_lengths()[5] - cfg["npred"]
    return 2.0 * cfg["npred"] * w * (cfg["nneg"] + 1) * cfg["hidden"] * b


def measured_traffic(kernel):
    """HBM bytes per launch of a kernel FAMILY from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    tools/summarize_pmc.py from the --pmc FETCH_SIZE / WRITE_SIZE runs of this same command); None when absent.  `kernel` is
    "<name prefix>:<config>": every instantiation of the newest profile whose name starts with the prefix counts (the plane-fed
    NT kernel runs as <0, false, 6, false> and, with the norm in its epilogue, <0, false, 6, true>), weighted by its launches."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
        prefix, cfg = kernel.rsplit(":", 1)
        prefix = prefix.rstrip(">")
        hits = [v for k, v in table.items() if k.endswith(":" + cfg) and k.startswith(prefix)]
        if not hits:
            return None
        newest = max(h.get("source", "") for h in hits)
        hits = [h for h in hits if h.get("source", "") == newest]
        w = [h.get("launches_per_step", 1.0) for h in hits]
        return int(round(sum(h["bytes_per_launch"] * x for h, x in zip(hits, w)) / sum(w)))
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def build(cfg, device):
    import cpc2_amd
    from cpc2_amd.train import buildOptimizer
    torch.manual_seed(0)                                   # identical init on every rank
    enc = cpc2_amd.CPCEncoder(cfg["hidden"], "layerNorm")
    if cfg["ar"] == "transformer":
        from cpc2_amd.transformers import buildTransformerAR
        ar = buildTransformerAR(cfg["hidden"], cfg["hidden"], cfg["layers"], WINDOW // 160, False)
    else:
        ar = cpc2_amd.CPCAR(cfg["hidden"], cfg["hidden"], False, cfg["layers"], mode=cfg["ar"])
    model = cpc2_amd.CPCModel(enc, ar).to(device)
    crit = cpc2_amd.CPCUnsupersivedCriterion(cfg["npred"], cfg["hidden"], cfg["hidden"], cfg["nneg"],
                                             rnnMode=cfg.get("rnn", "linear"), sizeInputSeq=WINDOW // 160,
                                             multihead_rnn=cfg.get("multihead", False)).to(device)
    opt = buildOptimizer(model, crit, lr=2e-4)
    return model, crit, opt


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: affinity mask, capped by the cgroup quota and by 16 (the GPU box's
    per-GPU share; its host has many more cores than the job may use)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("CPC_BENCH_MAX_CORES", "16"))))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(cfg, seconds_budget):
    """The oracle's train step (CPU restatement of the reference, fp32 torch-CPU ops) timed on this host's cores on a bounded
    sample of the same workload: b = 8 (the reference's default batchSizeGPU) and b = 16 windows, CPU model and
    core count in the record.  `value` is the b = 8 rate."""
    from oracle import cpc_oracle as O, synth
    from oracle.mt19937 import MT19937
    torch.set_num_threads(host_cores())
    log(f"cpu baseline on {torch.get_num_threads()} threads")
    h = cfg["hidden"]
    ar = cfg["ar"] if cfg["ar"] in GATES else "GRU"          # the transformer configs keep the GRU-model baseline

    def run(b, budget, max_steps):
        mp = synth.encoder_params(h, 1)
        mp.update(synth.gru_params(h, h, cfg["layers"], 2, gates=GATES[ar]))
        cp = synth.predictor_params(cfg["npred"], h, h, 3)
        x = synth.audio_windows(b, WINDOW, 4)
        params = {k: v.clone().requires_grad_(True) for k, v in list(cp.items()) + list(mp.items())}
        opt = O.Adam({k: v.data for k, v in params.items()})
        mt = MT19937(1234)

        def step():
            tot, _l, _a = O.train_step_loss(x, x, {k: params[k] for k in mp}, {k: params[k] for k in cp}, mt,
                                            cfg["npred"], cfg["nneg"], cfg["layers"], ar)
            grads = torch.autograd.grad(tot, list(params.values()))
            opt.step(dict(zip(params, grads)))

        step()                                                  # warm-up
        log(f"cpu baseline b={b}: warm-up step done")
        t0 = time.perf_counter()
        n = 0
        while n < 2 or (time.perf_counter() - t0 < budget and n < max_steps):
            step()
            n += 1
            log(f"cpu baseline b={b}: step {n}")
        dt = (time.perf_counter() - t0) / n
        return {"value": round(b * SECONDS_PER_WINDOW / dt, 3), "windows": b, "steps": n, "s_per_step": round(dt, 3)}

    r8 = run(8, 0.45 * seconds_budget, 5)
    r16 = run(16, 0.3 * seconds_budget, 3)
    return {"value": r8["value"], "unit": "audio-seconds/sec", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
            "kind": "port",
            "sample": f"oracle train step (fwd+bwd+Adam, reference semantics), fp32 torch-CPU: b=8 windows, {r8['steps']} steps, "
                      f"{r8['s_per_step']:.2f} s/step; b=16 windows, {r16['steps']} steps, {r16['s_per_step']:.2f} s/step",
            "by_batch": {"b8": r8, "b16": r16}}


class GpuBoxSnapshot:
    """What the GPU box itself was doing: this GPU's shader-clock level and power and how many OTHER GPUs of the host were busy (the
    box is one GPU of a shared 8-GPU host), ONE read of sysfs taken OUTSIDE the timed region while a few extra steps are queued on
    the device.  (Read from a thread inside the timed region -- the first form, at 50 Hz and then once -- it was the disturbance:
    every read is a request to the GPU's power controller; a rank pinned to two cores went from 4.9 to 20 ms per step, and a single
    read cost the region 20 ms of wall time and once a 23 ms step: profiles/r06_process_group_queues.md, section 5.)"""

    def __init__(self, device):
        import glob
        self.own, self.others, self.power = None, [], None
        try:
            pr = torch.cuda.get_device_properties(device)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:                                         # noqa: BLE001
            want = None
        for c in sorted(glob.glob("/sys/class/drm/card*/device")):
            if not os.path.exists(os.path.join(c, "pp_dpm_sclk")):
                continue
            if want is not None and os.path.basename(os.path.realpath(c)) == want:
                self.own = c
            else:
                self.others.append(c)
        if self.own is not None:
            hits = glob.glob(os.path.join(self.own, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(self.own, "hwmon", "hwmon*", "power1_input"))
            self.power = hits[0] if hits else None

    @staticmethod
    def _read(path):
        try:
            return open(path).read()
        except OSError:
            return ""

    def take(self):
        if self.own is None:
            return None
        rec = {"card": self.own.split("/")[4], "taken": "during extra steps after the timed region", "other_gpus_of_the_host": len(self.others)}
        for name in ("sclk", "mclk", "fclk", "socclk"):
            for ln in self._read(os.path.join(self.own, "pp_dpm_" + name)).splitlines():
                if ln.rstrip().endswith("*"):
                    digits = "".join(ch for ch in ln.split(":")[-1] if ch.isdigit())
                    if digits:
                        rec[name + "_level_mhz"] = int(digits)
        if self.power is not None:
            txt = self._read(self.power).strip()
            if txt.isdigit():
                rec["power_w"] = round(int(txt) / 1e6)
        busy = 0
        for c in self.others:
            txt = self._read(os.path.join(c, "gpu_busy_percent")).strip()
            busy += 1 if (txt.isdigit() and int(txt) >= 20) else 0
        rec["other_gpus_busy"] = busy
        return rec


def hold_the_collector():
    """Python's cyclic collector now and then walks every tracked object of the process: with torch loaded that is a 70-165 ms stop
    of the training thread about once per 15-25 thousand steps, longer than the host's lead over the device, i.e. one step of that
    length (profiles/r06_long_soak.txt; with the collector off the same 40 000 steps had none).  gc.freeze() after the warm-up --
    what cpc2_amd.train.run() does in front of its epoch loop -- moves everything alive so far out of the collector's way; objects
    made from here on are collected as before.  CPC_BENCH_NO_GC_FREEZE=1 leaves the collector alone."""
    import gc
    if os.environ.get("CPC_BENCH_NO_GC_FREEZE"):
        return "untouched"
    gc.unfreeze()              # (an earlier configuration of this process: its objects are ordinary garbage by now)
    gc.collect()
    gc.freeze()
    return "frozen after the warm-up steps (gc.freeze)"


def measure(args, cfg_name, device, rank, world, use_dist, steps, warmup, cpu_seconds):
    """Build CONFIGS[cfg_name], run `warmup` untimed and `steps` timed steps (barrier + synchronize on both sides, MAX over
    ranks) and return the result record (rank 0: the dict that is printed; other ranks: None)."""
    from cpc2_amd import _lib
    from cpc2_amd.train import DataParallelContext, cpcStep
    cfg = CONFIGS[cfg_name]
    lib = _lib.load()
    prev_mode = lib.cpc_gemm_set_mode(cfg.get("gemm_mode", 0))
    try:
        return _measure(args, cfg_name, cfg, lib, device, rank, world, use_dist, steps, warmup, cpu_seconds)
    finally:
        lib.cpc_gemm_set_mode(prev_mode)


def _measure(args, cfg_name, cfg, lib, device, rank, world, use_dist, steps, warmup, cpu_seconds):
    from cpc2_amd import _lib
    from cpc2_amd.train import DataParallelContext, backward, cpcStep
    model, crit, opt = build(cfg, device)
    dedup = bool(args.dedup or cfg.get("dedup"))
    strict = bool(cfg.get("strict"))
    # N > 1: the criterion / context-network gradient slices are all-reduced under the encoder's backward (train.py)
    dp = DataParallelContext(opt, early_params=list(crit.parameters()) + list(model.gAR.parameters()),
                             overlap=not os.environ.get("CPC_BENCH_NO_OVERLAP"), timing=True)
    if os.environ.get("CPC_BENCH_FOLLOW_TORCH"):
        # the module's default: the negatives come from torch's global CPU generator, as the reference's torch.randint calls do
        torch.manual_seed(1234 + rank)
    else:
        crit.seed(1234 + rank)                              # per-rank negative stream
    crit.sampler.prefetch = True                            # host draws step i+1's MT19937 words during step i
    g = torch.Generator().manual_seed(1000 + rank)          # per-rank shard of the synthetic utterances
    x = (0.05 * torch.randn(args.batch, 1, WINDOW, generator=g)).to(device)
    label = torch.zeros(args.batch, dtype=torch.long, device=device)

    def step():
        tot, losses, _acc = cpcStep(x, x, label, model, crit, dedup=dedup, dp=dp, strict=strict)
        backward(tot)                                       # (train.py:109; cpc2_amd.train.backward seeds it with a cached 1.0)
        dp.reduce_and_step()
        opt.zero_grad()
        return losses

    log(f"rank {rank}: {cfg_name}: model built, starting {warmup} warm-up steps")
    for i in range(warmup):
        losses = step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    torch.cuda.synchronize()
    log("warm-up done")
    gc_note = hold_the_collector()
    if use_dist:
        dist.barrier()
    prof = not args.no_prof
    roof = "gemm_planes_nt" if cfg["hidden"] % 256 == 0 else "gemm_nt"
    if prof:
        lib.cpc_prof_enable(2 if roof == "gemm_planes_nt" else 3)   # the roofline kernel only inside the timed region
    torch.cuda.synchronize()
    dp.timing_reset()
    # what a step costs the HOST, and whether any one step stands out: one event per step on the compute stream (a barrier packet,
    # ~1 us), the host's clocks around step(), the seconds the host spent blocked by cause (cpc2_amd._lib.HOST_WAITS)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    _lib.HOST_WAITS.clear()
    enqueue_s = 0.0
    cpu0, thr0 = time.process_time(), time.thread_time()
    marks[0].record()
    t0 = time.perf_counter()
    for i in range(steps):
        ta = time.perf_counter()
        losses = step()
        enqueue_s += time.perf_counter() - ta
        marks[i + 1].record()
    t_enqueued = time.perf_counter() - t0
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0                  # this rank's own clock (before the closing barrier)
    cpu_s, thr_s = time.process_time() - cpu0, time.thread_time() - thr0
    waits = dict(_lib.HOST_WAITS)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    med = sorted(step_ms)[steps // 2]
    worst = max(range(steps), key=lambda i: step_ms[i])
    host_rec = {
        "cores": PINNED_CORES if PINNED_CORES is not None else host_cores(), "pinned": PINNED_CORES is not None,
        # wall time the host spent inside step() (cpcStep + backward + reduce_and_step + zero_grad), waits included
        "enqueue_ms_per_step": round(1e3 * enqueue_s / steps, 3),
        # ... of which blocked: the sampler's worker thread, its buffer-release events (they keep the host <= 2 steps ahead of the
        # device), a collective's work.wait()
        "blocked_ms_per_step": {k: round(1e3 * v / steps, 3) for k, v in sorted(waits.items())},
        "busy_ms_per_step": round(1e3 * (enqueue_s - sum(waits.values())) / steps, 3),
        "thread_cpu_ms_per_step": round(1e3 * thr_s / steps, 3),        # CPU time of the training thread
        "process_cpu_ms_per_step": round(1e3 * cpu_s / steps, 3),       # ... of the whole process (worker, RCCL threads)
        "host_done_before_device_ms": round(1e3 * (own_elapsed - t_enqueued), 3),   # device work left when the last step() returned
        # per step on the compute stream (events): a one-off stall reads here, not as a slower average
        "step_ms_median": round(med, 3), "step_ms_min": round(min(step_ms), 3), "step_ms_max": round(step_ms[worst], 3),
        "step_ms_max_index": worst, "steps_over_2x_median": [i for i in range(steps) if step_ms[i] > 2 * med],
        "python_gc": gc_note,
    }
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    exposed_ms = dp.timing_read()                            # per step: compute stream held by the exchange (events)
    lib.cpc_prof_enable(0)
    _lib.check(lib.cpc_async_error_check(_lib.stream_ptr(device)), "async error check")
    log(f"{cfg_name}: timed region done: {1e3 * elapsed / steps:.2f} ms/step")
    rank_ms = [1e3 * own_elapsed / steps] * 2
    if use_dist:
        tmax = torch.tensor([elapsed, own_elapsed, -own_elapsed, exposed_ms if exposed_ms is not None else 0.0], dtype=torch.float64,
                            device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        rank_ms = [1e3 * float(-tmax[2].item()) / steps, 1e3 * float(tmax[1].item()) / steps]     # min, max over the ranks
        exposed_ms = float(tmax[3].item()) if exposed_ms is not None else None
    final_loss = [round(float(v), 4) for v in losses.detach().cpu().view(-1)]

    kernels = {}

    def read_classes(names, n_steps):
        for name in names:
            tot, cnt = ctypes.c_double(0), ctypes.c_long(0)
            lib.cpc_prof_read(name.encode(), ctypes.byref(tot), ctypes.byref(cnt))
            if cnt.value:
                kernels[name] = {"ms_per_step": round(tot.value / n_steps, 4), "launches_per_step": cnt.value / n_steps,
                                 "avg_launch_us": round(1e3 * tot.value / cnt.value, 2)}

    if prof:
        read_classes((roof, "side_wait"), steps)           # measured over the timed region
        if "side_wait" in kernels:
            # not a kernel: the time the training stream stood still at its joins with the library's side stream (events around the
            # waits).  ~0.1 ms when the deferred work runs BESIDE the backward pass; its whole length (~1 ms) when it does not
            host_rec["training_stream_held_by_side_stream_ms_per_step"] = kernels.pop("side_wait")["ms_per_step"]
        # the other classes: a few extra steps after the clock has stopped (timing every class costs ~0.1 ms per step)
        extra = 5
        lib.cpc_prof_enable(1)
        for _ in range(extra):
            step()
        host_rec["gpu_box"] = GpuBoxSnapshot(device).take()     # (the device is busy with the extra steps; the clock has stopped)
        torch.cuda.synchronize()
        lib.cpc_prof_enable(0)
        tot, cnt = ctypes.c_double(0), ctypes.c_long(0)
        lib.cpc_prof_read(roof.encode(), ctypes.byref(tot), ctypes.byref(cnt))   # discard: already taken from the timed region
        read_classes(tuple(c for c in ("gemm_planes_nt", "gemm_planes_tn", "gemm_nt", "gemm_tn", "infonce_fwd", "infonce_bwd",
                                       "gru_fwd", "gru_bwd", "conv0_fwd", "conv0_bwd") if c != roof), extra)
    cur = _lib.stream_ptr(device)
    if use_dist and dp.active:
        # RCCL's own stream is torch's to pick: does it share the training stream's hardware queue?  A 2 ms spin kernel on the
        # training stream, a one-element all-reduce issued from an idle third stream (so that the collective does not wait for the
        # spin by its own ordering rule): finished while the spin still runs = on another queue.  After the clock has stopped.
        # EVERY rank makes the same calls in the same order -- a barrier (the ranks start the probe together: a late peer would read
        # as "behind the spin"), then the all-reduce whatever the spin's launch returned: collectives issued by rank 0 alone hang
        # the job at N > 1 (this probe did, in the round-6 build before this one, when rehearsed with two gloo ranks).
        probe = torch.zeros(1, device=device)
        idle = torch.cuda.Stream(device)
        torch.cuda.synchronize()
        dist.barrier()
        lib.cpc_stream_spin.restype = ctypes.c_int
        spun = torch.cuda.Event()
        spin_rc = lib.cpc_stream_spin(cur, ctypes.c_long(200000))
        spun.record()
        with torch.cuda.stream(idle):
            work = dist.all_reduce(probe, async_op=True)
            work.wait()
            done = torch.cuda.Event()
            done.record()
        done.synchronize()
        host_rec["rccl_stream_runs_beside_training_stream"] = (not spun.query()) if spin_rc == 0 else "spin kernel not launched"
        torch.cuda.synchronize()
    if rank != 0:
        return None
    ms = 1e3 * elapsed / steps
    value = world * args.batch * SECONDS_PER_WINDOW * steps / elapsed
    out = {
        "metric": "audio-seconds/sec CPC training (1.28 s @16 kHz, 128 neg)",
        "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if cfg.get("gemm_mode", 0) == 0 else "f32 storage, conv products a0b0+a0b1+a1b0 of the bf16 split (16-bit product mantissa; opt-in, not the headline)",
        "data": "synthetic",
        "config": {"workload": f"CPC-{cfg_name} (hidden {cfg['hidden']}, {cfg['ar']} x{cfg['layers']}, nPredicts "
                               f"{cfg['npred']}, {cfg['nneg']} negatives, {('multi-head ' if cfg.get('multihead') else '') + cfg.get('rnn', 'linear')} predictors), {args.batch} x 1.28 s "
                               f"windows per GPU, " + ("past==future deduplicated (encoder+AR on b windows), "
                                                         if dedup else
                                                         "reference trainStep dataflow (encoder+AR on 2b windows), " if strict else
                                                         "reference trainStep semantics: encoder on the 2b windows of cat([past, future]), "
                                                         "context network on the b context windows (train.py:102 drops the other "
                                                         "half's context) and, when recurrent, on the W = T - nPredicts frames the criterion "
                                                         "reads (criterion.py:296 drops the rest); identical outputs, gradients and update, ")
                               + "fwd+bwd+allreduce+Adam",
                   "windows_per_gpu": args.batch, "global_batch": world * args.batch,
                   "parallelism": f"dp{world}", "final_losses": final_loss,
                   "inputs": "x = 0.05 randn, generator seed 1000 + rank; criterion MT19937 stream seed 1234 + rank; model "
                             "init torch.manual_seed(0) on every rank (SURVEY 8d's recipe with per-rank shards)"},
    }
    default_workload = args.batch == 64 and not dedup
    if prof and roof in kernels:
        planes = roof == "gemm_planes_nt"
        flops, launches = (planes_nt_algorithmic_flops if planes else gemm_nt_algorithmic_flops)(args.batch, cfg, dedup)
        k = kernels[roof]
        achieved = flops / (k["ms_per_step"] * 1e-3) / 1e12
        terms = 3 if (planes and cfg.get("gemm_mode", 0) == 2) else 6
        gemm_peak = round(BF16_MFMA_PEAK_TFLOPS / terms, 1)
        kname = ("gemm_planes_kernel<0, false, 3>" if terms == 3 else "gemm_planes_kernel<0, false>") if planes else "gemm_nt_x6_kernel"
        out["roofline"] = {"bound": "mfma",
                           "kernel": kname + (" (conv1-4 forward and backward-data on pre-split bf16 planes)" if planes else
                                              " (conv1-4 forward and backward-data, context / predictor projections)"),
                           "achieved": round(achieved, 2), "peak": gemm_peak, "unit": "TFLOP/s",
                           "frac": round(achieved / gemm_peak, 4),
                           "peak_note": f"algorithmic f32 flops; peak = 2500 TFLOP/s dense bf16 MFMA / {terms} bf16 products per "
                                        "f32 product (" + ("f32-accurate bf16x6 split" if terms == 6 else "three-product mode")
                                        + f"); the f32 MFMA's own peak is {FP32_MFMA_PEAK_TFLOPS} TFLOP/s",
                           # HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE of the committed PMC passes), default workload only
                           "traffic": measured_traffic(kname + ":" + cfg_name) if default_workload else None,
                           "algorithmic_gflop_per_launch": round(flops / max(k["launches_per_step"], 1.0) / 1e9, 3),
                           "avg_launch_us": k["avg_launch_us"], "launches_per_step": k["launches_per_step"]}
    if "infonce_fwd" in kernels and cfg.get("rnn", "linear") == "linear":
        # north_star's one explicit kernel target: the [context x negatives] similarity matmul, >= 60 % of the f32 MFMA peak
        k = kernels["infonce_fwd"]
        sim = similarity_algorithmic_flops(args.batch, cfg)
        ach = sim / (k["ms_per_step"] * 1e-3) / 1e12
        hid = cfg["hidden"]
        sim_kernel = f"infonce_fwd_dma_kernel<{hid}>" if hid in (256, 512) else f"infonce_fwd_kernel<{hid}>"
        out["roofline_similarity"] = {"bound": "mfma", "kernel": sim_kernel + " (gather + similarity + cross-entropy fused)",
                                      "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                                      "algorithmic_gflop_per_launch": round(sim / 1e9, 3), "avg_launch_us": k["avg_launch_us"],
                                      "traffic": measured_traffic(sim_kernel + ":" + cfg_name) if default_workload else None}
    out["kernels"] = kernels
    if kernels:
        # the ten timed classes against the step (caller's stream and the library's side stream together; the small kernels
        # between them are not timed): what is left is launch gaps + untimed kernels, or -- when it jumps -- a stream that idled
        host_rec["timed_kernel_classes_ms_per_step"] = round(sum(k["ms_per_step"] for k in kernels.values()), 3)
    # the library's side stream (deferred backward work) and the training stream on different hardware queues?  Tested, not assumed
    # (cpc2_hip.h, cpc_stream_create_apart; 200 us, after the clock has stopped)
    side = ctypes.c_void_p()
    cur = _lib.stream_ptr(device)
    if lib.cpc_side_stream(cur, ctypes.byref(side)) == 0:
        host_rec["side_stream_runs_beside_training_stream"] = lib.cpc_streams_overlap(cur, side) == 1
    wst = ctypes.c_void_p()
    if lib.cpc_negidx_stream(crit.sampler._h, ctypes.byref(wst)) == 0 and wst.value:
        host_rec["sampler_stream_runs_beside_training_stream"] = lib.cpc_streams_overlap(cur, wst) == 1
    if getattr(dp, "_helper", None) is not None:
        host_rec["exchange_helper_stream_runs_beside_training_stream"] = lib.cpc_streams_overlap(cur, ctypes.c_void_p(dp._helper.cuda_stream)) == 1
    host_rec["streams_handed_out_untested"] = int(lib.cpc_stream_apart_failures())
    out["host"] = host_rec
    out["_gradient_bytes"] = 4 * opt.flat_grad.numel()
    out["_comm"] = {"early_bytes": 4 * sum(hi - lo for lo, hi in dp.early), "late_bytes": 4 * sum(hi - lo for lo, hi in dp.late)
                    if dp.early else 4 * opt.flat_grad.numel(),
                    "exposed_ms_per_step": None if exposed_ms is None else round(exposed_ms, 4),
                    "rank_ms_per_step_min": round(rank_ms[0], 3), "rank_ms_per_step_max": round(rank_ms[1], 3)}
    del model, crit, opt, dp, x
    torch.cuda.empty_cache()
    return out


def write_synthetic_corpus(root, n_windows, n_speakers=16, files_per_speaker=8, seed=7):
    """16-bit mono 16 kHz WAV files of gaussian noise at speech-like RMS, `n_speakers` directories: a little more than
    n_windows * 20480 samples in all (librispeech-like layout: speaker / file)."""
    import numpy as np
    import struct
    rs = np.random.RandomState(seed)
    n_files = n_speakers * files_per_speaker
    per_file = (n_windows * WINDOW) // n_files + WINDOW + 1
    for spk in range(n_speakers):
        d = os.path.join(root, f"spk{spk:03d}")
        os.makedirs(d, exist_ok=True)
        for i in range(files_per_speaker):
            pcm = np.clip(rs.standard_normal(per_file) * (0.05 * 32768.0), -32768, 32767).astype("<i2").tobytes()
            with open(os.path.join(d, f"utt{i:03d}.wav"), "wb") as fh:
                fh.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
                         + b"data" + struct.pack("<I", len(pcm)) + pcm)
    return n_files * per_file


def measure_feeder(args, cfg_name, device, rank, world, use_dist, resident_rate):
    """One epoch of cpc2_amd.train.trainStep fed by the window feeder from files on disk; returns the record (rank 0) or None."""
    import random
    import shutil
    import tempfile
    from cpc2_amd import _lib
    from cpc2_amd.dataset import AudioBatchData, findAllSeqs
    from cpc2_amd.train import DataParallelContext, backward, cpcStep, trainStep
    if world > 1:
        # (every rank's loader decides its own number of batches -- the same-speaker sampler's partial batches -- and a step is a
        #  collective: ranks with different counts would wait for each other for ever, as the reference's loop would)
        raise RuntimeError("the feeder configuration is measured on one GPU (the ranks' epochs differ in length)")
    cfg = CONFIGS[cfg_name]
    lib = _lib.load()
    steps_wanted = max(200, args.steps * 10)
    model, crit, opt = build(cfg, device)
    dp = DataParallelContext(opt, early_params=list(crit.parameters()) + list(model.gAR.parameters()), timing=False)
    crit.seed(1234 + rank)
    crit.sampler.prefetch = True
    tmp = tempfile.mkdtemp(prefix="cpc_feeder_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        t_a = time.perf_counter()
        total = write_synthetic_corpus(tmp, steps_wanted * args.batch, seed=7 + rank)
        t_b = time.perf_counter()
        random.seed(11 + rank)
        seqs, speakers = findAllSeqs(tmp, extension=".wav")
        data = AudioBatchData(tmp, WINDOW, seqs, None, len(speakers), device=device)
        t_c = time.perf_counter()
        log(f"{cfg_name}: {total / 16000.0 / 3600.0:.2f} h of synthetic audio in {len(seqs)} files written in {t_b - t_a:.1f} s, loaded to the device in {t_c - t_b:.1f} s")
        # warm-up = whole epochs, timed apart: the same-speaker sampler ends every speaker with a partial batch whose size changes with
        # the epoch's random offset, and every new batch size has first-use costs (pinned staging rings, gradient buffers, kernel
        # variants the runtime loads at their first launch: ~12 ms each under a kernel trace) that a one-second epoch would otherwise
        # be charged with and a training run is not.  Measured (tools/feeder_epochs.py): 5.52 / 5.13 / 4.63 / 4.69 / 4.68 ms per step
        # for epochs 0 .. 4.
        import contextlib
        import io
        warm_ms = []
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(3):
                t_w = time.perf_counter()
                first_logs = trainStep(data.getDataLoader(args.batch, "samespeaker", True), model, crit, opt, None, 1000, dp=dp)
                torch.cuda.synchronize()
                warm_ms.append(round(1e3 * (time.perf_counter() - t_w) / max(1, int(first_logs["iter"])), 3))
        if use_dist:
            dist.barrier()
        loader = data.getDataLoader(args.batch, "samespeaker", True)
        n_batches = len(loader)
        _lib.HOST_WAITS.clear()
        cpu0, thr0 = time.process_time(), time.thread_time()
        printed = io.StringIO()
        with contextlib.redirect_stdout(printed):
            t0 = time.perf_counter()
            seen = [0]
            sizes, marks, box = [], [], []

            def counted(it):                          # (the same-speaker sampler ends every speaker with a partial batch)
                for item in it:
                    if len(marks) == n_batches - 2:     # (near the end of the epoch, two steps still queued on the device)
                        box.append(GpuBoxSnapshot(device).take())
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()                       # on the training stream, in front of this iteration's work
                    marks.append(ev)
                    sizes.append(int(item[0].size(0)))
                    seen[0] += sizes[-1]
                    yield item
            logs = trainStep(counted(loader), model, crit, opt, None, 1000, dp=dp)          # logging_step: the reference's default
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        cpu_s, thr_s = time.process_time() - cpu0, time.thread_time() - thr0
        waits = dict(_lib.HOST_WAITS)
        # per iteration on the device (event to event): full batches, batches of another size than their predecessor, outliers
        dev_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
        full = sorted(d for d, b in zip(dev_ms, sizes) if b == args.batch)
        changed = [d for i, d in enumerate(dev_ms) if i > 0 and sizes[i] != sizes[i - 1]]
        med_full = full[len(full) // 2] if full else None
        step_rec = {"full_batch_median_ms": None if med_full is None else round(med_full, 3),
                    "size_changes": len(changed), "after_size_change_mean_ms": round(sum(changed) / len(changed), 3) if changed else None,
                    "over_1p5x_full_median": sum(1 for d in dev_ms if med_full and d > 1.5 * med_full),
                    "ms_in_those": round(sum(d for d in dev_ms if med_full and d > 1.5 * med_full), 1)}
        if use_dist:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax[0].item())
        iters, windows = int(logs["iter"]), seen[0]
        if use_dist:
            wsum = torch.tensor([windows], dtype=torch.float64, device=device)
            dist.all_reduce(wsum)
            windows = int(wsum[0].item())
        _lib.check(lib.cpc_async_error_check(_lib.stream_ptr(device)), "async error check")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if rank != 0:
        return None
    ms = 1e3 * elapsed / iters
    return {
        "metric": "audio-seconds/sec CPC training (1.28 s @16 kHz, 128 neg)",
        "value": round(windows * SECONDS_PER_WINDOW / elapsed, 2), "unit": "audio-seconds/sec", "n_gpus": world,
        "steps": iters, "warmup": 3 * int(first_logs["iter"]), "ms_per_step": round(ms, 3), "warmup_epochs_ms_per_step": warm_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic audio FILES (16-bit WAV, written to a temporary directory) through the window feeder",
        "config": {"workload": f"CPC-{cfg_name}: the reference's loop around the headline step -- findAllSeqs -> AudioBatchData (flat audio in "
                               f"HBM) -> getDataLoader({args.batch}, 'samespeaker', randomOffset=True) -> cpc2_amd.train.trainStep (logging_step "
                               f"1000: train.py:596), the FOURTH epoch of {iters} steps of up to {args.batch} windows (the first three, with every batch size's first-use costs, are `warmup_epochs_ms_per_step`); dataset.py:300-325,366-408, train.py:95-134",
                   "windows_per_gpu": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}",
                   "final_losses": [round(float(v), 4) for v in logs["locLoss_train"]],
                   "windows": windows, "mean_batch": round(windows / max(1, iters * world), 2), "epoch_batches_announced": n_batches},
        # audio-seconds per second against the headline's (windows already resident, every batch full) in the same run
        "vs_resident_synthetic": None if not resident_rate else round(windows * SECONDS_PER_WINDOW / elapsed / resident_rate, 4),
        "host": {"cores": PINNED_CORES if PINNED_CORES is not None else host_cores(), "pinned": PINNED_CORES is not None,
                 "loop_ms_per_step": round(1e3 * t_host / iters, 3),          # wall time of trainStep's loop per step (waits included)
                 "blocked_ms_per_step": {k: round(1e3 * v / iters, 3) for k, v in sorted(waits.items())},
                 "busy_ms_per_step": round(1e3 * (t_host - sum(waits.values())) / iters, 3),
                 "thread_cpu_ms_per_step": round(1e3 * thr_s / iters, 3), "process_cpu_ms_per_step": round(1e3 * cpu_s / iters, 3),
                 "host_done_before_device_ms": round(1e3 * (elapsed - t_host), 3), "device_steps": step_rec, "gpu_box": box[0] if box else None,
                 "corpus_s": round(t_b - t_a, 1), "load_s": round(t_c - t_b, 1)},
    }


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Start `n` copies of this command, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in
    their environment, rendezvous on 127.0.0.1 at a free port), wait for them and return the worst exit code.  Rank 0
    prints the JSON line on the stdout the children inherit.  The parent never initialises the GPU
    (torch.cuda.device_count() does not)."""
    import subprocess
    backend = os.environ.get("CPC_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and n_dev < n and "--rendezvous-only" not in sys.argv:
        print(f"bench.py --gpus {n}: only {n_dev} GPU(s) visible and RCCL needs one per rank "
              "(CPC_BENCH_BACKEND=gloo rehearses the N-rank path with ranks sharing a GPU)", file=sys.stderr)
        return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    log(f"launcher: started {n} ranks (backend {backend}, port {port}), pids {[p.pid for p in procs]}")
    worst = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0:
                worst = worst or rc
                log(f"launcher: rank process {p.pid} exited with {rc}; stopping the others")
                for q in pending:              # exactly the processes started above
                    q.kill()
        time.sleep(0.05)
    return worst if worst >= 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="windows per GPU")
    ap.add_argument("--config", default="small", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-prof", action="store_true", help="skip the in-situ kernel timing")
    ap.add_argument("--dedup", action="store_true",
                    help="one encoder/AR pass when past is future (identical results; NOT the reference's 2b-window step)")
    ap.add_argument("--also", default=None,
                    help="comma-separated configs measured AFTER the headline's timed region and reported inside the same JSON "
                         "line under \"other_configs\" (default at one GPU with the small config: large,transformer = BASELINE "
                         "configs[4] per GPU and configs[3]; '' = none)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, all-reduce one CPU number over gloo, print it and exit (launcher self-test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher (no GPU call has been made in this process) and run
        # one rank per GPU as child processes -- the entry the reference takes through init_distributed_mode
        # (cpc/distributed_training/distributed_mode.py:75-86,129-142)
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree (unset WORLD_SIZE to let bench.py "
                         "start the ranks itself)")
    if args.rendezvous_only:
        # the launcher + rendezvous alone, on CPU tensors over gloo (tests/test_bench_launcher.py; no GPU needed)
        dist.init_process_group(backend="gloo", init_method="env://", world_size=world, rank=rank)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous": world, "rank_sum": float(t.item()), "master": os.environ["MASTER_ADDR"]}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    # one process per GPU; CPC_BENCH_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the
    # distributed code path on a one-GPU box (ranks then share the device)
    backend = os.environ.get("CPC_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # CPC_BENCH_FORCE_DIST=1: go through the process-group code path (RCCL init, broadcast, all-reduce) with one rank too
    use_dist = world > 1 or bool(os.environ.get("CPC_BENCH_FORCE_DIST"))
    if use_dist:
        if "MASTER_ADDR" not in os.environ:                 # (one forced rank started by hand)
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)

    # The CPU leg FIRST (rank 0 of a one-GPU run): ~20 s of host work on the oracle's step.  The GPU legs then start on a box that has
    # paged its libraries in and finished starting up -- the first process on a fresh box has been seen at 23 ms per step with the host
    # descheduled for 17 of them (tests/test_bench_gpu.py) -- and the timed region never shares the host with the CPU leg.
    cpu_rec = None
    if args.cpu_seconds > 0 and world == 1 and not CONFIGS[args.config].get("feeder"):
        cpu_rec = cpu_baseline(CONFIGS[args.config], args.cpu_seconds)
    if CONFIGS[args.config].get("feeder"):
        out = measure_feeder(args, args.config, device, rank, world, use_dist, None)
    else:
        out = measure(args, args.config, device, rank, world, use_dist, args.steps, args.warmup, args.cpu_seconds)
    if out is not None and cpu_rec is not None:
        out["cpu_baseline"] = cpu_rec
    if out is not None:
        comm = {"world": world, "process_group": backend if use_dist else None}
        if use_dist and backend == "nccl":
            comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            comm["env"] = {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "RCCL_MSCCL_ENABLE",
                                                          "HSA_ENABLE_IPC_MODE_LEGACY")}
        comm["gradient_bytes"] = out.pop("_gradient_bytes", None)
        # what the exchange costs a step: `exposed_ms_per_step` = time the compute stream is held between the start of the late
        # (encoder-slice) all-reduce and the end of the waits for the early ones, from events on that stream (max over ranks);
        # early / late bytes = what is reduced under the encoder's backward / after it; per-rank step times: min and max
        comm.update(out.pop("_comm", {}))
        comm["overlap"] = "criterion + context-network slices reduced under the encoder's backward" if (
            use_dist and not os.environ.get("CPC_BENCH_NO_OVERLAP")) else "one blocking all-reduce" if use_dist else None
        out["comm"] = comm
    also = args.also
    if also is None:
        # (N > 1: the line still carries BASELINE configs[4] -- CPC-large is DEFINED as the 8-GPU data-parallel config)
        also = ("large,transformer,small_strict,small_feeder,recipe,small_3term,small_dedup" if world == 1 else "large") \
            if (args.config == "small" and not args.no_prof) else ""
    others = []
    for name in [n for n in also.split(",") if n]:
        # (an extra configuration must never cost the headline its line: a failure is recorded in its place.  Every rank runs the
        #  same code on the same shapes, so an exception is raised on every rank alike and no collective is left half-entered)
        try:
            if CONFIGS[name].get("feeder"):
                rec = measure_feeder(args, name, device, rank, world, use_dist, out["value"] if (out and args.config == "small") else None)
            else:
                rec = measure(args, name, device, rank, world, use_dist, max(5, args.steps // 2), 3, 0.0)
        except Exception as exc:                                   # noqa: BLE001
            log(f"{name}: FAILED: {exc!r}")
            rec = {"config": {"workload": f"CPC-{name}"}, "error": repr(exc)[:400]} if rank == 0 else None
        if rec is not None:
            rec.pop("_gradient_bytes", None)
            comm_rec = rec.pop("_comm", None)
            if world > 1 and comm_rec is not None:                 # (N > 1: what the exchange cost this configuration)
                rec["comm"] = comm_rec
            others.append(rec)
    if rank == 0:
        if others:
            out["other_configs"] = others
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
