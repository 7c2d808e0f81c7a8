#!/usr/bin/env python3
"""bench.py -- LR-voxel-frames/sec of the x4 SR BMCNet training step on synthetic NFS-shaped event data.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one sequence batch (train.py:202-237 semantics): event->count scatter
of the batch's LR and HR frames, 8 recurrent BMCNet windows forward (state carried without detach), summed MSE,
one backward through all windows, gradient all-reduce (N > 1), Adam(amsgrad) step.  Workload at every N:
BASELINE.json configs[1] per GPU (BMCNet x4, 180x240 -> 720x960, bs=4/GPU, fp32, SEQL=9, SEQN=2) -> weak scaling;
configs[2] (bs=32 over 8 GPUs) is exactly the N=8 point.  Inputs (events) are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

At N = 1 the same line also carries `extra` blocks measured after the headline run (`--also config3,config4,infer,dist1`,
the default; `--also none` skips them): BASELINE configs[3] (EventZoom 31x56, its own bf16 arithmetic, and fp32 beside it),
the reference's literal NFS LR shape (45x80, bs 2: config/train_nfs.yml:71), configs[4]'s per-GPU shape (RGB 180x190, T = 16
windows, 8 sequences, per-window recompute) and the streaming-inference latency per window (infer_BMCNet.py:44-68)
with and without HIP-graph replay, and (`dist1`) the C2 step once more with the multi-GPU path's gradient reducer and an RCCL
all-reduce over a world of one rank in the timed region.  They are short (a few steps each), labelled with their own workload strings, and
never enter `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "bmcnet-esr_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

# algorithmic work (measured on the reference with torch.utils.flop_counter, 2 FLOP/MAC)
FLOP_PER_LRPX_FWD_BWD = 118_121_472      # BMCNet(4,128,5) one window forward+backward
FLOP_PER_LRPX_FWD = 41_574_912           # ... forward only (inference)
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6           # 16 x the fp32 rate (v_mfma_f32_32x32x16_bf16, dense, 2.4 GHz)
HBM_PEAK_GBPS = 8000.0                   # MI355X_MICROARCH.md: HBM3E spec peak (~6.3 TB/s achievable)
# peak of the dominant kernel per arithmetic mode, in algorithmic (fp32-equivalent) FLOP/s: the bf16x6 split spends six
# bf16 MFMAs per algorithmic product
KERNEL_PEAK = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "bf16": PEAK_BF16_MFMA_TFLOPS}
KERNEL_NAME = {"fp32": "conv_kernel<9,128>", "bf16x6": "conv_bf_kernel<9,128,8,3>", "bf16": "conv_bf_kernel<9,128,8,1>"}
PMC_KERNEL = {"fp32": "conv_kernel<9,128,8>", "bf16x6": "conv_bf_kernel<9,128,8,3>", "bf16": "conv_bf_kernel<9,128,8,1>"}
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")
DTYPE = {"fp32": "f32", "bf16x6": "f32 (3xbf16 split, 6 products)", "bf16": "bf16"}
ARITH = {"fp32": "fp32 (native fp32 MFMA; 3x3 convolutions through Winograd F(4x4,3x3) / F(2x2,3x3) transforms in fp32)", "bf16x6": "fp32-equivalent (3 bf16 planes, 6 products, fp32 accumulate)",
         "bf16": "bf16 operands, fp32 accumulate and storage"}
SHAPES = {(180, 240): "NFS-shaped 180x240 (BASELINE configs[1] / [2])", (31, 56): "EventZoom 31x56 (BASELINE configs[3] shape)",
          (180, 190): "RGB 180x190 (BASELINE configs[4] per-GPU shape)", (45, 80): "NFS 45x80 (the reference's own LR size, config/train_nfs.yml)"}


def shape_name(H, W):
    return SHAPES.get((H, W), "synthetic %dx%d" % (H, W))


def host_info():
    info = {"logical_cpus": os.cpu_count()}
    try:
        txt = open("/proc/cpuinfo").read()
        models = [l.split(":", 1)[1].strip() for l in txt.splitlines() if l.startswith("model name")]
        cores = {(a, b) for a, b in zip([l.split(":")[1].strip() for l in txt.splitlines() if l.startswith("physical id")],
                                         [l.split(":")[1].strip() for l in txt.splitlines() if l.startswith("core id")])}
        info["cpu_model"] = models[0] if models else None
        info["physical_cores"] = len(cores) or None
    except OSError:
        pass
    return info


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def cpu_baseline(dev=None, budget_hw=(180, 240), threads=16):
    """CPU oracle (the PyTorch-CPU restatement of the reference path, oracle/bmc_oracle.py) timed on this host, the protocol
    of SURVEY 8(d) / BASELINE.md 4: 2 warm-up + 3 timed iterations, median, thread and core counts stated.
      (ii) one BMCNet window forward+backward, B=1, one 180x240 LR frame -> the headline `value` in frames per second;
      (iii) `other_samples`: one window, and the full 8-window BPTT of train.py:202-237, at B=1 on the reference's own 45x80 frame;
      (i)  events_to_channels on one 180x240 LR frame (24 576 events) and one 720x960 HR frame (393 216 events), numpy
           restatement, one thread -- with the GPU scatter kernel's time for the same frames beside it.
    16 threads is the measured optimum of torch-CPU on the GPU box's host for this network (8: 1.08 s, 16: 0.67 s, 32: 1.08 s,
    64: 2.3 s, 128: 8.6 s per quarter frame); `all_host_cores` carries the same window on every logical CPU of the box beside it.
    Reported only; never the thing optimised."""
    import numpy as np
    import torch.nn.functional as F
    from models.BMCNet import BMCNet
    from oracle import bmc_oracle as O
    torch.manual_seed(3407)
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    scale, n_c, n_b = 4, 128, 5
    H, W = budget_hw
    m = BMCNet(scale, n_c, n_b)
    params, seen = {}, {}
    for k, v in m.state_dict().items():
        params[k] = seen.setdefault(v.data_ptr(), v.clone().requires_grad_())
    x = torch.poisson(torch.full((1, 2, 2, H, W), 0.284))
    gt = torch.poisson(torch.full((1, 2, scale * H, scale * W), 0.284))
    z = lambda c: torch.zeros(1, c, H, W)
    times = []
    for it in range(5):
        t0 = time.perf_counter()
        _, _, _, pred = O.bmcnet_forward(params, x, z(n_c), z(n_c), z(n_c), z(32), True, scale)
        loss = F.mse_loss(pred, gt)
        loss.backward()
        times.append(time.perf_counter() - t0)
        for p in seen.values():
            p.grad = None
    t = _median(times[2:])
    frames = (H * W) / (180.0 * 240.0)
    # SURVEY 8(d) asks for the host's cores with the count stated: the same window on ALL logical CPUs of the box beside the
    # measured optimum (torch-CPU's intra-op pool loses to its own synchronisation on this network from 32 threads up: 128 threads
    # took 8.6 s per quarter frame, and 256 threads did not finish a quarter frame in 20 minutes on the EPYC 9575F host of round 6).
    # So: the reference's own 45x80 frame, in a CHILD process with a hard 90 s limit -- a leg that does not finish says so instead of
    # taking the bench line with it.
    all_cores = None
    # (one thread per PHYSICAL core: with one per logical CPU -- 256 on the EPYC 9575F host -- not one 45x80 window finished in 90 s)
    ncpu = host_info().get("physical_cores") or os.cpu_count() or 1
    if ncpu > torch.get_num_threads():
        import subprocess
        code = (
            "import json, os, sys, time\n"
            "sys.path[:0] = [%r, %r]\n"
            "import torch, torch.nn.functional as F\n"
            "from models.BMCNet import BMCNet\n"
            "from oracle import bmc_oracle as O\n"
            "torch.manual_seed(3407); torch.set_num_threads(%d)\n"
            "m = BMCNet(4, 128, 5); params, seen = {}, {}\n"
            "for k, v in m.state_dict().items(): params[k] = seen.setdefault(v.data_ptr(), v.clone().requires_grad_())\n"
            "h, w = 45, 80\n"
            "x = torch.poisson(torch.full((1, 2, 2, h, w), 0.284)); gt = torch.poisson(torch.full((1, 2, 4 * h, 4 * w), 0.284))\n"
            "z = lambda c: torch.zeros(1, c, h, w); ts = []\n"
            "for it in range(3):\n"
            "    t0 = time.perf_counter()\n"
            "    _, _, _, pred = O.bmcnet_forward(params, x, z(128), z(128), z(128), z(32), True, 4)\n"
            "    F.mse_loss(pred, gt).backward(); ts.append(time.perf_counter() - t0)\n"
            "    for p in seen.values(): p.grad = None\n"
            "    print(json.dumps(ts), flush=True)\n"
        ) % (ROOT, os.path.join(ROOT, "bmcnet-esr_amd"), ncpu)
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(ncpu))
        sample = "one 45x80 window fwd+bwd (the reference's LR frame), B = 1, %d torch threads = one per physical core of the host (%d logical CPUs), child process" % (ncpu, os.cpu_count() or 1)
        try:
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=90)
            lines = [l for l in r.stdout.splitlines() if l.startswith("[")]
            ts = json.loads(lines[-1]) if lines else []
        except subprocess.TimeoutExpired as e:
            out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            lines = [l for l in out.splitlines() if l.startswith("[")]
            ts = json.loads(lines[-1]) if lines else []
            sample += "; stopped at the 90 s limit after %d of 3 iterations" % len(ts)
        if ts:
            tq = min(ts[1:]) if len(ts) > 1 else ts[0]
            all_cores = {"value": round((45 * 80) / (180.0 * 240.0) / tq, 5), "unit": "LR-voxel-frames/s", "cores": ncpu,
                         "sample": sample + " (seconds per iteration: %s; the 16-thread figure for the same window is other_samples.one_window_45x80)" % " ".join("%.2f" % v for v in ts)}
        else:
            all_cores = {"value": None, "unit": "LR-voxel-frames/s", "cores": ncpu, "sample": sample + ": no iteration finished"}
    # (iii) the reference's own LR frame (45x80, config/train_nfs.yml): one window forward+backward, and the FULL training-loop
    # body of train.py:202-237 -- 8 recurrent windows forward, summed MSE, one backward through all of them -- at B = 1.
    # (At 180x240 the 8-window BPTT of the CPU path keeps ~62 GB of activations -- SURVEY Appendix A.11 -- and takes > 60 s per
    # iteration: it is timed at 45x80, where it fits the bounded-sample budget; 1 warm-up + 2 timed iterations.)
    h2, w2, L = 45, 80, 9
    x2 = torch.poisson(torch.full((1, L, 2, h2, w2), 0.284))
    g2 = torch.poisson(torch.full((1, L, 2, scale * h2, scale * w2), 0.284))
    legs = {}
    for tag, nwin, iters in (("one_window_45x80", 1, 4), ("bptt_8_windows_45x80", L - 1, 3)):
        xs = [x2[:, i:i + 2].transpose(1, 2) for i in range(nwin)]
        gts = [g2[:, i + 1] for i in range(nwin)]
        ts = []
        for it in range(iters):
            t0 = time.perf_counter()
            loss, _, _ = O.bptt_loss(params, xs, gts, n_c, scale)
            loss.backward()
            ts.append(time.perf_counter() - t0)
            for p in seen.values():
                p.grad = None
        tm = _median(ts[1:])
        legs[tag] = {"seconds_per_iteration": round(tm, 3), "LR_frames_per_s": round(nwin / tm, 3), "frame": "%dx%d" % (h2, w2),
                     "windows": nwin, "B": 1, "iterations": "1 warm-up + %d timed (median)" % (iters - 1)}
    # (i) the event -> count scatter
    rng = np.random.default_rng(3407)
    ev = {}
    for tag, (h, w, n) in {"180x240": (180, 240, 24576), "720x960": (720, 960, 393216)}.items():
        xs, ys = rng.integers(0, w, n).astype(np.float32), rng.integers(0, h, n).astype(np.float32)
        ps = rng.choice([-1.0, 1.0], n).astype(np.float32)
        ts = []
        for it in range(5):
            t0 = time.perf_counter()
            O.events_to_channels_np(xs, ys, ps, (h, w))
            ts.append(time.perf_counter() - t0)
        ev[tag] = {"events": n, "cpu_ms": round(_median(ts[2:]) * 1e3, 3)}
        if dev is not None:
            from bmc_hip import ops
            d = [torch.tensor(a, device=dev) for a in (xs, ys, ps)]
            off = torch.tensor([0, n], dtype=torch.int64, device=dev)
            for _ in range(3):
                ops.events_to_channels_batched(*d, off, h, w, mutate=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.events_to_channels_batched(*d, off, h, w, mutate=False)
            e1.record()
            torch.cuda.synchronize()
            ev[tag]["gpu_ms_one_frame_launch"] = round(e0.elapsed_time(e1) / 20, 4)
    return {"value": round(frames / t, 5), "unit": "LR-voxel-frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "host": host_info(),
            "sample": "oracle/bmc_oracle.py BMCNet(4,128,5) 1 window fwd+bwd, B=1, LR %dx%d (%.2f of a 180x240 frame), "
                      "2 warm-up + 3 timed (median %.2fs; all five: %s), %d torch threads" %
                      (H, W, frames, t, " ".join("%.2f" % v for v in times), torch.get_num_threads()),
            "all_host_cores": all_cores,
            "other_samples": legs,
            "events_to_channels": dict(ev, note="numpy restatement (oracle.events_to_channels_np), 1 thread, 2 warm-up + 3 timed "
                                                "(median); gpu = bmc_events_to_channels, ONE frame per launch (the step batches 36 frames per launch)")}


# Executed / algorithmic multiplies of a kernel kind (ops.py's profile names): the Winograd kernels do not execute the
# multiplies the metric counts (F(4x4): 36 of 144 per 4x4 tile; F(2x2): 16 of 36 per 2x2 tile), the fused centre chain's
# backward executes 5 of the 6 C^2 it is credited with.  `roofline.frac` is built on EXECUTED work: a utilisation, <= 1.
EXECUTED = {"wino4_conv<9,128>": 36.0 / 144.0, "wino_conv<9,128>": 16.0 / 36.0, "wgrad_wino<9>": 16.0 / 36.0, "wgrad_wino4<9>": 36.0 / 144.0,
            "chain_kernel<bwd>": 5.0 / 6.0}
KERNEL_LABEL = {"wino4_conv<9,128>": "wino4_conv_kernel [Winograd F(4x4,3x3) on the fp32 MFMA, csrc/wino4.hip]",
                "wino_conv<9,128>": "wino2_conv_kernel [Winograd F(2x2,3x3) on the fp32 MFMA, csrc/wino.hip]",
                "wgrad_wino<9>": "wino_wgrad_kernel [Winograd F(2x2,3x3) weight gradient, csrc/wino_wgrad.hip]",
                "wgrad_wino4<9>": "wino4_wgrad_kernel [Winograd F(4x4,3x3) weight gradient, csrc/wino4_wgrad.hip]"}
PMC_NAME = {"wino4_conv<9,128>": "wino4_conv_kernel", "wino_conv<9,128>": "wino2_conv_kernel<8>"}


def dominant_kernel_roofline(step_fn, iso, math, shape_key, step_ms=None):
    """roofline block for the dominant kernel = the kernel kind with the largest share of the step's GPU time (round 4: the
    F(4x4) Winograd 3x3 convolution, forward + data gradients).  One extra, untimed step runs with an event pair around every
    launch on its launch stream (torch's current stream); per kind: algorithmic FLOPs / time (what the metric counts) and
    EXECUTED FLOPs / time (what its matrix instructions sustain).  `achieved` / `frac` are the executed figures -- a
    utilisation of the dense MFMA peak, never above 1; `achieved_algorithmic` / `frac_algorithmic` stand beside them.
    avg_launch_ms is directly comparable with rocprofv3 --stats' average for the kernel."""
    from bmc_hip import ops

    def profiled_step():
        ops.PROFILE = []
        step_fn()
        torch.cuda.synchronize()
        rec, ops.PROFILE = ops.PROFILE, None
        agg = {}
        for kind, flops, e0, e1, nbytes in rec:
            a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1); a[3] += nbytes
        return agg

    # As the timed region runs it: in the fp32 mode the weight-gradient kernels run on a second stream beside the data-gradient
    # chain (ops.wgrad_side), so a kernel's in-situ duration includes the CUs it shares with them -- these are the durations
    # rocprofv3 --kernel-trace --stats of this command reports.  `alone` (below): the same step once more with everything on one
    # stream, i.e. every kernel with the chip to itself.
    agg = profiled_step()
    concurrent = any(st.side for st in ops._SIDE.values())      # did that step's backward use the second stream?
    agg_alone = None
    if concurrent:
        side, ops.WGRAD_SIDE = ops.WGRAD_SIDE, "0"
        try:
            agg_alone = profiled_step()
        finally:
            ops.WGRAD_SIDE = side
    peak = KERNEL_PEAK[math]
    total_ms = sum(v[2] for v in agg.values())

    def row(kind, agg=agg, with_situ=False):
        """One kernel's figures over `agg`.  with_situ (the `kernels` list of a two-stream step): the unqualified fields are the
        kernel BY ITSELF (the one-stream step), the in-situ-with-sharing ones stand under `in_situ_shared`."""
        if with_situ and agg_alone is not None and kind in agg_alone:
            r = row(kind, agg_alone)
            rs = row(kind, agg)
            r["in_situ_shared"] = {k: rs[k] for k in ("avg_launch_ms", "ms_per_step", "share_of_profiled_kernel_time", "executed_tflops", "frac", "hbm_frac")}
            return r
        n, fl, ms, nb = agg[kind]
        total_ms = sum(v[2] for v in agg.values())
        alg = fl / (ms * 1e-3) / 1e12 if fl else None
        ex = alg * EXECUTED.get(kind, 1.0) if alg else None
        hbm = nb / (ms * 1e-3) / 1e9 if nb else None          # algorithmic bytes (operands once, output once) over the launch time
        return {"kernel": kind, "hbm_algorithmic_GBps": round(hbm, 1) if hbm else None,
                "hbm_frac": round(hbm / HBM_PEAK_GBPS, 4) if hbm else None, "launches": n, "avg_launch_ms": round(ms / n, 4), "ms_per_step": round(ms, 2),
                "share_of_profiled_kernel_time": round(ms / total_ms, 4),
                "achieved_algorithmic_tflops": round(alg, 2) if alg else None, "frac_algorithmic": round(alg / peak, 4) if alg else None,
                "executed_tflops": round(ex, 2) if ex else None, "frac": round(ex / peak, 4) if ex else None,
                "executed_over_algorithmic": round(EXECUTED.get(kind, 1.0), 4)}

    mfma_kinds = [k for k, v in agg.items() if v[1] > 0]
    dom = max(mfma_kinds, key=lambda k: agg[k][2])
    n, fl, ms, _ = agg[dom]
    d_situ = row(dom)
    # Headline figures: the kernel BY ITSELF (the one-stream step) when the timed step runs two streams -- a launch's in-situ
    # duration then contains CUs shared with the weight-gradient stream and understates every kernel; the
    # in-situ-with-sharing figures stand beside them under `in_situ_shared`
    d = row(dom, agg_alone) if agg_alone is not None and dom in agg_alone else d_situ
    # HBM bytes per launch and the PMC figures: from the committed rocprofv3 PMC passes over one bench step (the AVERAGE in-step
    # launch of this kernel, FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes; tools/pmc_summary.py) --
    # STATIC data, not measured in this run, attached ONLY when the summary was collected on this very workload and kernel
    traffic, pmc = None, None
    if os.path.exists(PMC_FILE):
        summ = json.load(open(PMC_FILE))
        if summ.get("_workload") == shape_key:
            entry = summ.get(math, {}).get(PMC_NAME.get(dom, PMC_KERNEL[math]))
            if entry:
                traffic = entry.get("hbm_bytes_per_launch")
                pmc = {k: entry[k] for k in ("mfma_busy_frac", "in_kernel_clock_GHz", "hbm_GBps", "lds_bank_conflict_frac") if k in entry}
                pmc.update({"static": True, "source": "profiles/" + os.path.basename(PMC_FILE), "collected_at_commit": summ.get("_commit")})
    out = {"bound": "mfma", "kernel": "%s (3x3 convolution fwd + dgrad, %d launches of one step)" % (KERNEL_LABEL.get(dom, KERNEL_NAME[math] if dom.startswith("conv_kernel<9") else dom), n),
           "achieved": d["executed_tflops"], "peak": round(peak, 1), "unit": "TFLOP/s", "frac": d["frac"],
           "achieved_algorithmic": d["achieved_algorithmic_tflops"], "frac_algorithmic": d["frac_algorithmic"],
           "hbm_algorithmic_GBps": d["hbm_algorithmic_GBps"], "hbm_frac": d["hbm_frac"],
           "note": "achieved / frac: EXECUTED matrix FLOPs of the kernel (algorithmic x %.4f) over its launch time, against the dense "
                   "fp32 MFMA peak -- a utilisation; the algorithmic rate (2 x 9 x Cin x Cout FLOP per pixel, what the metric counts) is "
                   "achieved_algorithmic.%s" % (EXECUTED.get(dom, 1.0),
                   "  In the timed step the weight-gradient kernels run CONCURRENTLY on a second stream: a launch's in-situ duration "
                   "(`in_situ_shared`, and rocprofv3 --stats of this command: profiles/r06_bench_fp32_kernel_stats.csv) contains CUs "
                   "shared with them and the kernel times add up to more than the step.  The headline figures (achieved, frac, "
                   "avg_launch_ms) are therefore the kernel BY ITSELF: the same step once more on one stream (BMC_WGRAD_STREAM=0; "
                   "rocprofv3 --stats of that command: profiles/r06_bench_fp32_onestream_kernel_stats.csv).  `step_frac_executed` is the "
                   "whole two-stream step as one utilisation, `mfma_floor_ms` what its executed matrix FLOPs cost at the dense peak." if concurrent else ""),
           "traffic": traffic, "traffic_static": traffic is not None,
           "avg_launch_ms": d["avg_launch_ms"], "flop_per_launch": fl / n, "launches_per_step": n,
           "isolated_2B_128to128": iso, "pmc": pmc,
           "streams": "weight-gradient kernels on a second stream beside the data-gradient chain (ops.wgrad_side): in-situ durations include shared CUs" if concurrent else "one stream",
           "in_situ_shared": None if agg_alone is None else dict(
               {k: d_situ[k] for k in ("avg_launch_ms", "ms_per_step", "executed_tflops", "frac", "achieved_algorithmic_tflops")},
               note="the kernel's launches inside the timed two-stream step, CUs shared with the weight-gradient stream"),
           "alone_ms_per_step": None if agg_alone is None else round(sum(v[2] for v in agg_alone.values()), 1),
           # every kernel kind with >= 4 % of the profiled kernel time, the same two rates each -- by itself (one-stream step) when the
           # timed step runs two streams, with the in-situ-with-sharing figures under `in_situ_shared`
           "kernels": [row(k, with_situ=True) for k in sorted(agg_alone or agg, key=lambda k: -(agg_alone or agg)[k][2])
                       if (agg_alone or agg)[k][2] >= 0.04 * sum(v[2] for v in (agg_alone or agg).values())],
           "profiled_kernel_ms_per_step": round(total_ms, 1)}
    if step_ms:
        # the whole step as a utilisation: executed matrix FLOPs of all profiled kernels over the step time
        ex_flops = sum(v[1] * EXECUTED.get(k, 1.0) for k, v in agg.items())
        out["step_executed_tflops_per_gpu"] = round(ex_flops / (step_ms * 1e-3) / 1e12, 2)
        out["step_frac_executed"] = round(ex_flops / (step_ms * 1e-3) / 1e12 / peak, 4)
        out["mfma_floor_ms"] = round(ex_flops / (peak * 1e12) * 1e3, 1)      # the step's executed matrix FLOPs at the dense peak
    return out


def isolated_conv(dev, B, H, W, n_c, iters=30):
    """The single launch shape that dominates (3x3 n_c->n_c over the doubled twin batch 2B), back to back."""
    from bmc_hip import ops
    from bmc_hip.ops import ConvSpec, View
    spec = ConvSpec.dense(n_c)
    x = torch.randn(2 * B, H, W, n_c, device=dev)
    w = torch.randn(n_c, n_c, 3, 3, device=dev) * 0.03
    b = torch.full((n_c,), 0.01, device=dev)      # (a dense bias: what every layer has after the first optimizer step; with an
                                                  #  exactly-zero bias the forward launch keeps F(2x2) -- ops.wino_ok's exact-zero rule)
    with torch.no_grad():
        for _ in range(5):
            ops.conv([View(x)], w, b, spec, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.conv([View(x)], w, b, spec, relu=True)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * (2 * B * H * W) * n_c * (9 * n_c)
    wn = ops.wino_ok(2 * B, H, W, n_c, 9, fwd=True, rule=b)
    ex = {4: 36.0 / 144.0, 2: 16.0 / 36.0}.get(wn, 1.0)
    return {"avg_launch_ms": round(ms, 4), "achieved_algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
            "executed_tflops": round(flops * ex / (ms * 1e-3) / 1e12, 2), "frac": round(flops * ex / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "kernel": {4: "wino4_conv_kernel", 2: "wino2_conv_kernel"}.get(wn, "conv_kernel<9,128>")}


def workload_string(n_c, n_b, H, W, B, L, math, scale=4, dist_on=False, recompute=False, graph=False):
    return ("BMCNet(scale=%d, n_c=%d, n_b=%d) x%d SR train step, %s -> %dx%d, bs=%d/GPU, arithmetic %s, SEQL=%d SEQN=2 (%d windows "
            "BPTT), event scatter + fwd + MSE + bwd + Adam(amsgrad)%s%s%s" %
            (scale, n_c, n_b, scale, shape_name(H, W), scale * H, scale * W, B, ARITH[math], L, L - 1,
             " + RCCL grad all-reduce" if dist_on else "", " [per-window recompute]" if recompute else "",
             " [HIP graph replay]" if graph else ""))


class Workload:
    """One training-step workload on this rank's GPU: model, optimizer, resident synthetic events, the step closure."""

    def __init__(self, dev, B, H, W, L, n_c, n_b, math, recompute=False, graph=False, use_dist=False, rank=0, scale=4):
        from models.BMCNet import BMCNet
        from bmc_hip import ops
        from bmc_hip.parallel import GradAllReducer
        from train_step import bptt_step, encode_sequence, synthetic_events
        self.dev, self.B, self.H, self.W, self.L, self.n_c, self.n_b, self.math = dev, B, H, W, L, n_c, n_b, math
        self.scale, self.recompute, self.graph_mode = scale, recompute, graph
        ops.set_math(math)
        torch.manual_seed(3407)                                   # same init on every rank (reference default seed)
        self.model = BMCNet(scale, n_c, n_b).to(dev)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-4, weight_decay=1e-5, amsgrad=True,    # config/train_nfs.yml:28-34
                                    capturable=graph)
        self.reducer = GradAllReducer(self.model, self.opt) if use_dist else None
        # event density of the reference NFS config (0.569 ev/px/frame), EventZoom / RGB windows likewise rounded to 1024
        n_lr = int(round(0.5689 * H * W / 1024)) * 1024 if (H, W) != (180, 240) else 24576
        self.ev = synthetic_events(B, L, H, W, scale, max(n_lr, 1024), dev, seed=3407 + rank)

        def step():
            inp, gt = encode_sequence(self.ev, B, L, H, W, scale)
            return bptt_step(self.model, self.opt, inp, gt, n_c, scale, recompute=recompute)

        self.eager_step = self.step = step
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):            # PyTorch's capture protocol: warm up on a side stream first
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.opt.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._g_out = self.eager_step()

            def replay():
                self._graph.replay()
                return self._g_out

            self.step = replay

    def step_flops(self):
        if (self.n_c, self.n_b, self.scale) != (128, 5, 4):
            return None
        return FLOP_PER_LRPX_FWD_BWD * self.H * self.W * self.B * (self.L - 1)

    def shape_key(self):
        return {"H": self.H, "W": self.W, "B": self.B, "L": self.L, "n_c": self.n_c, "n_b": self.n_b}


def pin_rank_to_cores(local_rank, world):
    """One rank's host threads on their own share of the box's cores, BEFORE anything touches the GPU: the small-frame steps are
    host-issue-bound (NOTEBOOK R5.9: ~15 us of Python per launch), and eight ranks whose launch threads, autograd threads and
    OpenMP pools float over the same cores contend.  Physical cores (with their SMT siblings) are dealt out in contiguous runs --
    contiguous core ids share a socket / NUMA node on the EPYC hosts of this pool; BMC_BENCH_PIN=0 leaves the affinity alone."""
    if world <= 1 or os.environ.get("BMC_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    cores, seen = [], set()
    for c in allowed:
        if c in seen:
            continue
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
            ids = set()
            for part in sib.split(","):
                lo, _, hi = part.partition("-")
                ids.update(range(int(lo), int(hi or lo) + 1))
        except (OSError, ValueError):
            ids = {c}
        ids &= set(allowed)
        seen |= ids
        cores.append(sorted(ids))
    per = len(cores) // world
    if per < 1:
        return None
    mine = [c for grp in cores[local_rank * per:(local_rank + 1) * per] for c in grp]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(per, 16)))
    return {"physical_cores": per, "logical_cpus": len(mine), "first": mine[0], "last": mine[-1]}


def timed(fn, warmup, steps, use_dist, dev):
    """W untimed + exactly K timed steps between barrier + synchronize pairs; max over ranks."""
    loss = None
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = fn()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, loss


def extra_train(dev, tag, B, H, W, L, math, steps, warmup, recompute=False, graph=False, use_dist=False):
    """A short side measurement of another BASELINE configuration on this GPU: same step code, its own shape and arithmetic.
    use_dist: with the gradient reducer (hooks + finish() + RCCL all-reduce) in the step; the process group must exist."""
    from bmc_hip import ops
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    wl = Workload(dev, B, H, W, L, 128, 5, math, recompute=recompute, graph=graph, use_dist=use_dist)
    dt, loss = timed(wl.step, warmup, steps, use_dist, dev)
    out = {"workload": workload_string(128, 5, H, W, B, L, math, recompute=recompute, graph=graph), "dtype": DTYPE[math],
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
           "value": round(B * (L - 1) * steps / dt, 2), "unit": "LR-voxel-frames/s",
           "step_achieved_tflops": round(wl.step_flops() * steps / dt / 1e12, 2),
           "step_frac_of_peak": round(wl.step_flops() * steps / dt / 1e12 / KERNEL_PEAK[math], 4),
           "peak_of": "%.1f TFLOP/s (%s MFMA, dense)" % (KERNEL_PEAK[math], "bf16" if math == "bf16" else "fp32" if math == "fp32" else "bf16 / 6"),
           "peak_mem_GiB": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1), "final_loss": round(float(loss), 6)}
    del wl
    ops.set_math("fp32")
    torch.cuda.empty_cache()
    return out


def extra_infer(dev, windows=24):
    """Streaming inference (infer_BMCNet.py:44-68): per-window latency of BMCNet(4,128,5), batch 1, fp32, events on the launch
    stream around the model call exactly where the reference puts its starter / ender pair; eager and HIP-graph replay."""
    from infer import StreamingSR
    from models.BMCNet import BMCNet
    torch.manual_seed(0)
    m = BMCNet(4, 128, 5).to(dev)
    out = {"workload": "BMCNet(4,128,5) streaming x4 SR inference, batch 1, fp32, SEQN=3 inputs (infer_BMCNet.py:147), per recurrent "
                       "window; mean over %d windows after 6 warm-up windows" % (windows - 6), "unit": "ms/window"}
    for H, W in ((45, 80), (180, 240)):
        frames = torch.poisson(torch.full((1, windows + 2, 2, H, W), 0.284)).to(dev)
        rec = {}
        for graph in (False, True):
            sr = StreamingSR(m, 128, 4, graph=graph)
            for i in range(windows):
                sr.step(frames[:, i:i + 3].transpose(1, 2))
            ms = sr.latency_ms(skip=6)
            rec["graph_replay" if graph else "eager"] = round(ms, 3)
        rec["frac_of_fp32_mfma_peak_graph"] = round(FLOP_PER_LRPX_FWD * H * W / (rec["graph_replay"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        rec["state_bytes_per_sequence"] = sr.state_bytes()
        # the recurrent feature state carried in bf16 between windows (StreamingSR(state_dtype=torch.bfloat16): half the resident
        # bytes per sequence; a storage format -- the kernels read and write fp32)
        sr = StreamingSR(m, 128, 4, state_dtype=torch.bfloat16)
        for i in range(windows):
            sr.step(frames[:, i:i + 3].transpose(1, 2))
        rec["eager_state_bf16"] = round(sr.latency_ms(skip=6), 3)
        rec["state_bytes_per_sequence_bf16"] = sr.state_bytes()
        out["%dx%d->%dx%d" % (H, W, 4 * H, 4 * W)] = rec
    del m
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)       # (two: the caching allocator still grows in the second step of a process)
    ap.add_argument("--batch", type=int, default=4, help="sequences per GPU")
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--width", type=int, default=240)
    ap.add_argument("--seql", type=int, default=9)
    ap.add_argument("--n_c", type=int, default=128)
    ap.add_argument("--n_b", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the whole step in a HIP graph and replay it (single GPU; pays off when the step is launch-bound, i.e. small frames)")
    ap.add_argument("--recompute", action="store_true", help="per-window activation recompute (long sequences / big batches)")
    ap.add_argument("--math", default=os.environ.get("BMC_MATH", "fp32"), choices=["fp32", "bf16x6", "bf16"],
                    help="arithmetic of the MFMA kernels for the headline measurement (default fp32 = native fp32 MFMA)")
    ap.add_argument("--no-bf16x6", action="store_true", help="skip the additional bf16x6-mode measurement")
    ap.add_argument("--also", default="config3,config4,infer,dist1",
                    help="comma list of extra blocks appended to the JSON line at --gpus 1 (config3, config4, infer, dist1; 'none' = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet (importing
        # torch does not), and it never will: the N ranks are CHILD processes of torch.distributed.run, one per GPU,
        # rank 0 prints the JSON line to the shared stdout, and this process exits with the launcher's code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    # stdout carries exactly ONE line, the JSON: everything else that writes to file descriptor 1 -- RCCL prints a version banner
    # there when its first communicator is created -- is sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    pinned = pin_rank_to_cores(local_rank, world)          # (before the first GPU call of this process)
    # Dry run of the multi-rank path on a box with ONE GPU (tests/test_gpu_r6.py): every rank on cuda:0, collectives over gloo
    # (RCCL refuses two ranks on one device).  Never the measured configuration: the JSON line says which backend ran.
    one_gpu = os.environ.get("BMC_BENCH_ONE_GPU") == "1"
    backend = os.environ.get("BMC_BENCH_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the BMCNet HIP path has no CPU fallback")
    if torch.cuda.device_count() < world and not one_gpu:
        raise SystemExit("bench.py: %d ranks requested but only %d GPUs visible" % (world, torch.cuda.device_count()))
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or bool(os.environ.get("BMC_FORCE_DIST"))      # BMC_FORCE_DIST: exercise RCCL with 1 rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus or os.environ.get("BMC_FORCE_DIST")
    if args.graph and use_dist:
        raise SystemExit("--graph is single-GPU only")
    # the `dist1` side run (the step with the gradient reducer and a 1-rank RCCL all-reduce) needs a process group; RCCL
    # initialised AFTER a 120 GiB workload has run in the process costs the steps that follow it 50-80 ms each for a while
    # (tools/second_workload.py), so the group of one is created here, before anything else; the headline run does not use it
    dist1_group = False
    if (world == 1 and not use_dist and "dist1" in args.also.split(",") and not args.graph
            and (args.batch, args.height, args.width, args.seql, args.n_c, args.n_b) == (4, 180, 240, 9, 128, 5)):
        import socket
        with socket.socket() as sk:              # (a free port: two bench runs on one box must not meet on a fixed one)
            sk.bind(("127.0.0.1", 0))
            free_port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port))
        try:
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
            dist.barrier()
            dist1_group = True
        except Exception:
            dist1_group = False

    from bmc_hip import ops
    n_c, n_b = args.n_c, args.n_b
    B, H, W, L = args.batch, args.height, args.width, args.seql
    wl = Workload(dev, B, H, W, L, n_c, n_b, args.math, recompute=args.recompute, graph=args.graph, use_dist=use_dist, rank=rank)

    dt, loss = timed(wl.step, args.warmup, args.steps, use_dist, dev)
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2**30
    lockstep = None
    if use_dist:
        # the ranks must be in lock step: identical parameters on every rank after the timed steps (a rank that missed an
        # all-reduce, or reduced a stale bucket, shows up here and FAILS the run instead of producing a number)
        with torch.no_grad():
            ck = torch.stack([torch.cat([p.detach().double().reshape(-1) for p in wl.model.parameters()]).sum(),
                              torch.cat([p.detach().double().abs().reshape(-1) for p in wl.model.parameters()]).sum()])
        if backend != "nccl":
            ck = ck.cpu()                        # (gloo gathers host tensors)
        allck = [torch.zeros_like(ck) for _ in range(dist.get_world_size())]
        dist.all_gather(allck, ck)
        lockstep = all(torch.equal(allck[0], c) for c in allck)
        if not lockstep:
            raise SystemExit("bench.py: parameters differ between ranks after %d steps: %s" % (args.steps + args.warmup, [c.tolist() for c in allck]))

    # the instrumented extra step contains the gradient all-reduce: every rank has to take part in it
    iso = isolated_conv(dev, B, H, W, n_c) if rank == 0 else None
    roof = dominant_kernel_roofline(wl.eager_step, iso, args.math, wl.shape_key(), step_ms=dt / args.steps * 1e3)
    # second arithmetic mode of the same step (every rank takes part): the fp32-equivalent bf16x6 split -- reported
    # beside the headline number, never as it
    split = None
    if args.math == "fp32" and not args.no_bf16x6 and not args.graph:
        ops.set_math("bf16x6")
        dt6, loss6 = timed(wl.eager_step, 1, args.steps, use_dist, dev)
        roof6 = dominant_kernel_roofline(wl.eager_step, None, "bf16x6", wl.shape_key(), step_ms=dt6 / args.steps * 1e3)
        ops.set_math("fp32")
        split = (dt6, float(loss6), roof6)
    step_flops = wl.step_flops()
    del wl
    torch.cuda.empty_cache()
    if rank == 0:
        windows = L - 1
        frames_per_step = world * B * windows
        value = frames_per_step * args.steps / dt
        if step_flops:
            roof["step_algorithmic_tflops_per_gpu"] = round(step_flops * args.steps / dt / 1e12, 2)
            roof["step_frac_algorithmic"] = round(step_flops * args.steps / dt / 1e12 / KERNEL_PEAK[args.math], 4)
        out = {
            "metric": "LR-voxel-frames/sec x4 SR train step, %s, %s" % (shape_name(H, W), DTYPE[args.math]),
            "value": round(value, 3), "unit": "LR-voxel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.math], "data": "synthetic",
            "config": {"workload": workload_string(n_c, n_b, H, W, B, L, args.math, dist_on=use_dist, recompute=args.recompute, graph=args.graph),
                       "global_batch": world * B, "frames_per_step": frames_per_step,
                       "parallelism": "dp%d" % world, "rccl_ranks": dist.get_world_size() if use_dist else 1,
                       "collective_backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend + " [dry run, not a measurement]") if use_dist else None,
                       "ranks_in_lock_step": lockstep, "rank_cpu_pinning": pinned,
                       "rccl_initialised_before_headline_run": bool(dist1_group) or (use_dist and backend == "nccl"),
                       "peak_mem_GiB": round(peak_mem, 1),
                       "final_loss": round(float(loss), 6)},
            "roofline": roof,
        }
        if split is not None:
            dt6, loss6, roof6 = split
            roof6.pop("isolated_2B_128to128", None)
            out["bf16x6_mode"] = {
                "roofline": roof6,
                "value": round(frames_per_step * args.steps / dt6, 3), "unit": "LR-voxel-frames/s",
                "ms_per_step": round(dt6 / args.steps * 1e3, 2), "steps": args.steps, "warmup": 1,
                "final_loss": round(loss6, 6),
                "arithmetic": "every fp32 MFMA operand split exactly into 3 bf16 planes, 6 plane products on "
                              "v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-equivalent (error vs float64 <= the "
                              "native fp32 MFMA path's, tests/test_gpu_parity.py::test_math_modes_vs_float64); same "
                              "workload, same step, `--math bf16x6` makes it the headline run"}
            if step_flops:
                out["bf16x6_mode"]["step_fp32_equivalent_tflops_per_gpu"] = round(step_flops * args.steps / dt6 / 1e12, 2)
        also = [] if args.also in ("", "none") or world != 1 or use_dist else [a.strip() for a in args.also.split(",") if a.strip()]
        if also and (H, W, B, L, n_c, n_b) == (180, 240, 4, 9, 128, 5) and not args.graph:
            extra = {}
            if "config3" in also:
                # ~5 900 launches of ~10 us: GPU time and host issue time are within 10 % of each other, so the eager step follows
                # the host's load (66 ... 83 ms on one box); replayed from a HIP graph (the same kernels, captured once) it does not
                c3 = extra_train(dev, "c3", 4, 31, 56, 9, "bf16", 10, 4)
                c3g = extra_train(dev, "c3g", 4, 31, 56, 9, "bf16", 10, 4, graph=True)
                c3["hip_graph_replay"] = {k: c3g[k] for k in ("workload", "ms_per_step", "value", "unit", "steps", "warmup")}
                extra["configs[3] EventZoom 31x56 bs4, bf16 (the config's arithmetic)"] = c3
                extra["configs[3] shape in fp32"] = extra_train(dev, "c3f", 4, 31, 56, 9, "fp32", 10, 4)
                extra["reference NFS LR shape 45x80 bs2 (config/train_nfs.yml:71), fp32"] = extra_train(dev, "nfs", 2, 45, 80, 9, "fp32", 10, 4)
            if "config4" in also:
                extra["configs[4] per-GPU shape RGB 180x190 T=16 bs8, fp32, per-window recompute"] = \
                    extra_train(dev, "c4", 8, 180, 190, 17, "fp32", 2, 1, recompute=True)
            if "infer" in also:
                extra["streaming inference latency"] = extra_infer(dev)
            if "dist1" in also:
                # the multi-GPU step's own code on ONE rank: GradAllReducer (hooks, finish(), bucket staging) + an RCCL all-reduce
                # over a world of 1 inside the timed region, next to the plain step (`ms_per_step` of this line) -- what the
                # reducer costs before an 8-GPU node ever runs it
                try:
                    if not dist1_group:
                        raise RuntimeError("no RCCL process group of one rank could be created")
                    # (at least 4 warm-up steps: the first steps after RCCL's initialisation are slower by 50-80 ms, tools/second_workload.py)
                    r = extra_train(dev, "dist1", 4, 180, 240, 9, "fp32", min(args.steps, 10), max(args.warmup, 4), use_dist=True)
                    r["plain_step_ms"] = out["ms_per_step"]
                    extra["C2 step with the gradient reducer + 1-rank RCCL all-reduce in the timed region"] = r
                except Exception as e:      # (an RCCL that refuses a world of one must not cost the headline line)
                    extra["C2 step with the gradient reducer + 1-rank RCCL all-reduce in the timed region"] = {"error": repr(e)[:300]}
                finally:
                    if dist.is_initialized():
                        dist.destroy_process_group()
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dev)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
