#!/usr/bin/env python
"""Throughput benchmark of the reverse-SDE sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch per GPU: 16 synthetic 4-s utterances
(wav already resident in HBM) -> STFT -> 30-step PC sampler (reverse_diffusion predictor +
1 annealed-Langevin corrector step = 60 score evaluations of NCSN++ 27.8 M, bf16 MFMA operands)
-> iSTFT.  This is BASELINE.json configs[1] (configs[2] = the same per GPU on 8 GPUs: utterances are
sharded over ranks, no data-path collective -> "scaling": "weak").

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     — the dominant kernel (implicit-GEMM 3x3 conv, bf16 MFMA): algorithmic FLOPs per
                 launch / average launch duration, measured with HIP events on the launch stream
                 (storm_program_run_timed) over profiled score evaluations of the same workload;
  cpu_baseline — the CPU oracle (oracle/, "port") timed on this host on a bounded sample
                 (one score evaluation of one utterance), extrapolated to utterances/s.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "enhanced utterances/sec @ 30 PC steps, NCSN++ 27.8M, 4 s@16 kHz"
# What the matrix pipes sustain at the socket's 1400 W cap on RANDOM bf16 operands when every 8 MFMAs re-read their six fragments from LDS,
# which every convolution must (tools/ubench/mfma_rate.hip, profiles/r02_ubench.txt / r03_ubench.txt: 1.55 - 1.57 PF; registers only: 1.82 PF;
# the 2.5 PF datasheet figure needs quiet operands and no operand traffic).  `roofline.power_bound` divides by THIS: the honest second denominator.
POWER_BOUND_TFLOPS = {"bf16": 1550.0, "fp16": 1440.0, "fp32": None}     # (fp16: the same streams run 7 % slower, profiles/r04_power_probe.txt)
POWER_CAP_W = 1400.0
HBM_PEAK_GBS = 8000.0            # HBM3E specification (MI355X_MICROARCH.md); measured float4 copy: 6290
BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak
F32_MFMA_PEAK_TFLOPS = 157.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=16, help="utterances per GPU per step")
    p.add_argument("--seconds", type=float, default=4.0)
    p.add_argument("--N", type=int, default=30, help="reverse steps of the PC sampler")
    p.add_argument("--corrector", default="ald", choices=["ald", "langevin", "none"])
    p.add_argument("--corrector-steps", type=int, default=1)
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--backbone", default="ncsnpp")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--profile-nfe", type=int, default=2, help="score evaluations profiled with per-op HIP events")
    p.add_argument("--ops-json", default=None, help="write the per-op timing table of the profiled pass here")
    p.add_argument("--sampler", default="pc", choices=["pc", "ode"], help="pc: predictor-corrector (configs[1-3]); ode: probability-flow RK45 (configs[4])")
    p.add_argument("--stream", type=int, default=0, metavar="N",
                   help="configs[4]: a step = N synthetic utterances of 2-10 s (seeded lengths), micro-batched by padded frame count "
                        "(<= --batch per launch, ragged rows) instead of one equal-length batch")
    p.add_argument("--no-group", action="store_true", help="--stream: one micro-batch after the other instead of grouped score evaluations (A/B)")
    p.add_argument("--width", type=int, default=None, help="--stream: at most this many micro-batches in flight (a finished one is replaced by the next); default: all of them")
    p.add_argument("--ode-idle", action="store_true", help="--sampler ode: rows that reached eps idle in their micro-batch instead of leaving it (A/B of the row compaction)")
    p.add_argument("--dist-world1", action="store_true",
                   help="with ONE rank: initialise the RCCL process group anyway and run the barrier / gather lines through it (dry run of the "
                        "N-rank path on one GPU; the driver launches the real one)")
    p.add_argument("--no-h2d", action="store_true",
                   help="skip the second timed pass from HOST wavs (pinned: H2D of the batch, the sampler, D2H of the result - SURVEY 8(d)), "
                        "which is reported as `from_host` beside `value`, never as it")
    p.add_argument("--include-h2d", action="store_true", help="(accepted for old command lines: the from-host pass is now the default)")
    p.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                   help="HIP-graph replay of the score evaluations (storm_ncsnpp_set_graph): auto = the library's rule (today: eager at every batch size, profiles/r05a_*)")
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip the `other_configs` object: BASELINE.json configs[3], a configs[4]-style ragged stream (PC and a shortened ODE leg) and "
                        "one utterance per call, each timed AFTER the headline region on rank 0 of a one-GPU run (never inside it, never part of `value`)")
    p.add_argument("--no-traffic", action="store_true",
                   help="do not re-measure `roofline.traffic` (two rocprofv3 --pmc passes, FETCH_SIZE and WRITE_SIZE, over a short run of this same command, "
                        "about a minute): keep the figure of profiles/conv_traffic.json")
    p.add_argument("--power", action="store_true",
                   help="sample socket power / clock with rocm-smi DURING the timed steps (perturbs rank 0: off by default; the default samples an "
                        "untimed repetition of the same step instead)")
    p.add_argument("--selftest-cpu", action="store_true",
                   help="(tests) run the launch / sharding / timing skeleton with a stand-in step on CPU ranks (gloo)")
    return p.parse_args()


def randomize(model, seed):
    """Seeded non-degenerate weights (the reference's init_scale=0 layers would zero the output; throughput
    and clocks must be measured on realistic values — zero tensors clock ~20 % higher on this chip)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                fan = p.shape[1] * (p[0][0].numel() if p.dim() > 2 else 1)
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5)
            elif name.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith(".W") and p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 16)
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))


def cpu_baseline(backbone, seconds, reps=3):
    """Score evaluations of one utterance with the CPU oracle (PyTorch fp32, all host cores): one warm-up, then the
    median of `reps`."""
    from oracle import ncsnpp_ref as NR
    cfg = NR.NCSNppConfig(**NR.NAMED_CONFIGS[backbone], input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=0)
    T = (1 + int(seconds * 16000) // 128 + 63) // 64 * 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 2, 256, T, dtype=torch.complex64, generator=g) * 0.3
    t = torch.tensor([0.5])
    times = []
    with torch.no_grad():
        NR.ncsnpp_forward(sd, cfg, x, t)                   # warm-up (allocator, thread pool, page-in)
        for _ in range(reps):
            t0 = time.perf_counter()
            NR.ncsnpp_forward(sd, cfg, x, t)
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], times, torch.get_num_threads()


def profile_ops(net, Y, nfe_count):
    """Per-op HIP-event timing of `nfe_count` score evaluations (same batch / shapes as the timed steps): the op program
    the C-ABI network object planned (storm_ncsnpp_program) run through storm_program_run_timed."""
    from storm_amd import _lib as L
    from storm_amd.backbones.plan import BUF_IN0, BUF_OUT, BUF_PARAMS, BUF_T, BUF_WS, N_BUFS
    B, _, F, T = Y.shape
    dev = Y.device
    code = L.dt(net.compute_dtype)
    h = net._get_handle(code, dev)
    ops, n, _ = net.program(B, F, T)
    ws = net._get_workspace(h, B, F, T, code, dev)
    x = torch.randn_like(Y)
    out = torch.empty_like(Y)
    tvec = torch.full((B,), 0.5, device=dev)
    bufs = (C.c_void_p * N_BUFS)()
    bufs[BUF_WS], bufs[BUF_PARAMS] = ws.data_ptr(), L.lib().storm_ncsnpp_arena(h)
    xin, yin = x[:, 0].contiguous(), Y[:, 0].contiguous()
    bufs[BUF_IN0], bufs[BUF_IN0 + 1] = torch.view_as_real(xin).data_ptr(), torch.view_as_real(yin).data_ptr()
    bufs[BUF_T], bufs[BUF_OUT] = tvec.data_ptr(), torch.view_as_real(out).data_ptr()
    acc = [0.0] * n
    ms = (C.c_float * n)()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(nfe_count):
        L.check(L.lib().storm_program_run_timed(ops, n, bufs, N_BUFS, code, st, ms), "storm_program_run_timed")
        for k in range(n):
            acc[k] += ms[k] / nfe_count
    rows = []
    tname = {0: "float", 1: "storm::bf16_t", 2: "storm::half_t"}.get(code, "storm::bf16_t")
    for k in range(n):
        op = ops[k]
        row = dict(idx=k, code=int(op.code), ms=acc[k])
        if op.code == 4:
            nseg, Bq, H, W, outC, Cout = [int(op.i[j]) for j in range(6)]
            flops, taps, esz = 0, [], 4 if code == 0 else 2
            abytes = Bq * H * W * outC * esz                # algorithmic HBM bytes: every operand once, the output once
            for gseg in range(nseg):
                q = 8 + 7 * gseg
                cin, nt = int(op.i[q]) + int(op.i[q + 1]), int(op.i[q + 4])
                flops += 2 * Bq * H * W * Cout * cin * nt
                abytes += (Bq * H * W * cin + Cout * cin * nt) * esz
                taps.append(nt)
            if int(op.p[9].buf) >= 0:
                abytes += Bq * H * W * outC * esz           # the residual / input-skip tensor added in the epilogue is read once too
            row.update(flops=flops, algorithmic_bytes=abytes, H=H, W=W, Cout=Cout, taps=taps, big=outC > 32, cin=[int(op.i[8]) + int(op.i[9])],
                       kernel=L.lib().storm_program_kernel_name(ops, k, code).decode())
        elif op.code == 6:                                  # GroupNorm-apply (+ SiLU) (+ FIR x2 of the activated AND the raw tensor)
            Ca, Cb, Bq, H, W = [int(op.i[j]) for j in range(5)]
            rs = int(op.i[7])
            esz, Cc = (4 if code == 0 else 2), Ca + Cb
            n_in = Bq * H * W * Cc
            n_out = n_in if rs == 0 else (2 * 4 * n_in if rs == 1 else 2 * n_in // 4)     # up / down: two output tensors
            row.update(algorithmic_bytes=(n_in + n_out) * esz, H=H, W=W, C=Cc, resample=rs,
                       kernel=L.lib().storm_gn_apply_kernel_name(Cc, Bq, H, W, int(op.i[6]), rs, code).decode())   # (what the launcher picks)
        elif op.code in (7, 8):                             # FIR x2 of the 8-channel pyramids
            Bq, H, W, Cc = [int(op.i[j]) for j in range(4)]
            esz = 4 if code == 0 else 2
            n_in = Bq * H * W * Cc
            row.update(algorithmic_bytes=(n_in + (4 * n_in if op.code == 7 else n_in // 4) + (4 * n_in if op.code == 7 else 0)) * esz,
                       kernel=f"storm::fir_kernel<{tname}, {1 if op.code == 7 else 2}>")
        elif op.code in (13, 14):                           # the 8-channel pyramids, one launch each (csrc/pyramid.hip)
            Bq, H, W, nl = int(op.i[1]), int(op.i[2]), int(op.i[3]), int(op.i[4] if op.code == 13 else op.i[5])
            esz = 4 if code == 0 else 2
            lv = sum(Bq * (H >> k) * (W >> k) * 8 * esz for k in range(nl))        # every level once
            # input: the complex inputs read (8 B each) + every level written; output: every ph read + the complex score written
            row.update(algorithmic_bytes=lv + Bq * H * W * 8 * (int(op.i[0]) if op.code == 13 else 1), H=H, W=W, levels=nl,
                       kernel=f"storm::{'input_pyramid_kernel' if op.code == 13 else 'output_pyramid_kernel'}<{tname}" + (", true>" if op.code == 13 else ">"))
        rows.append(row)
    net.release_program(ops)                               # (the rows above hold plain numbers: the op list may go)
    return rows


def ragged_stream(n, batch, seed, gen, dev):
    """configs[4]'s input: n synthetic utterances of 2-10 s (seeded lengths), micro-batched by padded frame count (<= batch rows per
    launch, rows keep their own lengths), resident in HBM.  Returns [(wav [b, Lmax], lengths or None)], lengths of all utterances."""
    from storm_amd import distributed as D
    lens = [int(v) for v in torch.randint(32000, 160001, (n,), generator=torch.Generator().manual_seed(seed))]
    batches = []
    for ids in D.bucket_by_frames(lens, batch):
        bl = [lens[i] for i in ids]
        yb = torch.zeros(len(ids), max(bl))
        for k, n_ in enumerate(bl):
            yb[k, :n_] = 0.1 * torch.randn(n_, generator=gen)
        batches.append((yb.to(dev), None if len(set(bl)) == 1 else bl))
    return batches, lens


def other_configs(model, dev, sync):
    """The configurations of BASELINE.json that are not the headline, each timed on its own AFTER the headline region (rank 0, one GPU):
    same engine, same wav -> wav step, synthetic inputs resident in HBM, one untimed pass first (plans / workspaces of the new shapes).
    Every entry carries its own workload, dtype, steps and ms_per_step; none of them enters `value`."""
    from storm_amd import distributed as D
    from storm_amd.model import ScoreModel
    out = {}

    def run(step, steps, warm_step=None):
        (warm_step or step)(0)                              # untimed: planning, workspace allocation, first-touch
        el, _, res = D.timed_steps(step, steps, 0, sync=sync)
        return el, res

    # ---- one utterance per call (the reference's own operating point, enhancement.py:66-72): latency and real-time factor
    g = torch.Generator().manual_seed(4321)
    wav1 = (0.1 * torch.randn(1, 64000, generator=g)).to(dev)
    model.set_precision("bf16")
    el, (_, nfe) = run(lambda i: model.enhance_batch(wav1, seed=7000 + i, return_nfe=True, N=30, corrector="ald", corrector_steps=1, snr=0.5), 3)
    out["one_utterance_per_call"] = {"workload": "configs[0]'s shape on the GPU: ncsnpp, ONE 4-s utterance per call, 30-step PC (reverse_diffusion + ald x1), bf16, wav->wav",
                                     "value": 3 / el, "unit": "utterances/s", "steps": 3, "ms_per_step": 1e3 * el / 3, "dtype": "bf16",
                                     "nfe_per_utterance": nfe, "rtf": el / 3 / 4.0}
    # ---- configs[4]-style: ragged stream, fp16, PC sampler and a shortened ODE leg
    model.set_precision("fp16")
    batches, lens = ragged_stream(32, 16, 77, g, dev)

    def stream_step(kw, sel, grouped=True):
        def step(i):
            outs, n_ = model.enhance_stream(sel, grouped=grouped, seed=9000 + 100 * i, return_nfe=True, **kw)
            return outs[-1], n_
        return step
    pc = dict(sampler_type="pc", predictor="reverse_diffusion", corrector="ald", N=30, corrector_steps=1, snr=0.5)
    el, (_, nfe) = run(stream_step(pc, batches), 1, warm_step=stream_step(dict(pc, N=1), batches))
    audio_s = sum(lens) / 16000.0
    out["configs4_stream_pc"] = {"workload": f"configs[4]-style on one GPU: ncsnpp, stream of 32 utterances of 2-10 s ({audio_s:.0f} s of audio) in {len(batches)} ragged "
                                             "micro-batches (<= 16 rows, bucketed by padded frame count), 30-step PC (reverse_diffusion + ald x1), fp16, wav->wav",
                                 "value": 32 / el, "unit": "utterances/s", "steps": 1, "ms_per_step": 1e3 * el, "dtype": "fp16", "nfe_per_utterance": nfe,
                                 "micro_batches": len(batches), "rtf_per_audio_second": el / audio_s,
                                 "grouped": "the micro-batches' samplers in lockstep, one grouped network call per step (storm_ncsnpp_forward_group)",
                                 "grouped_calls_rows": list(model.last_group_calls or ())}
    el_seq, _ = D.timed_steps(stream_step(pc, batches, grouped=False), 1, 0, sync=sync)[::2]
    out["configs4_stream_pc"]["value_one_micro_batch_after_the_other"] = 32 / el_seq
    # the ODE leg on the micro-batches that hold the first 8 utterances' worth of rows (shapes planned by the PC pass above)
    sel, n_sel = [], 0
    for b in sorted(batches, key=lambda b: -b[0].shape[0]):
        if n_sel >= 8:
            break
        sel.append(b)
        n_sel += b[0].shape[0]
    el, (_, nfe) = D.timed_steps(stream_step(dict(sampler_type="ode", N=30), sel), 1, 0, sync=sync)[::2]
    out["configs4_stream_ode"] = {"workload": f"configs[4] as configured, shortened: ncsnpp, {n_sel} utterances of 2-10 s in {len(sel)} ragged micro-batches of the same stream, "
                                              "probability-flow ODE sampler (RK45, rtol = atol = 1e-5, one step controller per row), fp16, wav->wav",
                                  "value": n_sel / el, "unit": "utterances/s", "steps": 1, "ms_per_step": 1e3 * el, "dtype": "fp16",
                                  "nfe_per_utterance": nfe, "micro_batches": len(sel), "grouped": True,
                                  "grouped_calls_rows": list(model.last_group_calls or ()), "rows_leave_their_micro_batch_at_eps": True}
    model.set_precision("bf16")
    # ---- configs[3]: ncsnpplarge (65.6 M), 8 utterances of 8 s per GPU, 50-step PC + 1 corrector step = 100 evaluations
    large = ScoreModel(backbone="ncsnpplarge", sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, spec_factor=0.15, spec_abs_exponent=0.5)
    randomize(large, seed=1)
    large._error_loading_ema = True
    large = large.eval().to(dev)
    large.set_precision("bf16")
    wav8 = (0.1 * torch.randn(8, 128000, generator=g)).to(dev)
    kw3 = dict(predictor="reverse_diffusion", corrector="ald", corrector_steps=1, snr=0.5, return_nfe=True)
    el, (_, nfe) = run(lambda i: large.enhance_batch(wav8, seed=8000 + i, N=50, **kw3), 1, warm_step=lambda i: large.enhance_batch(wav8, seed=1, N=1, **kw3))
    out["configs3"] = {"workload": "configs[3] per GPU: ncsnpplarge (65.6 M) batch=8x8 s@16 kHz, 50-step PC + 1 corrector (reverse_diffusion + ald x1 = 100 evaluations), bf16, wav->wav",
                       "value": 8 / el, "unit": "utterances/s", "steps": 1, "ms_per_step": 1e3 * el, "dtype": "bf16", "nfe_per_utterance": nfe,
                       "evaluation_tflops": 8 * 2131.0e9 * nfe / el / 1e12, "mfma_frac": 8 * 2131.0e9 * nfe / el / 1e12 / BF16_MFMA_PEAK_TFLOPS}
    del large
    return out


def measure_traffic(timeout_s=150):
    """HBM bytes per launch of every kernel of this bench command, measured NOW: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
    passes (MI355X_MICROARCH.md: separate passes, KiB units, FETCH_SIZE x2 on gfx950) over a short run of the same workload (one step of
    two reverse steps = four score evaluations of the batch).  Returns (table, note) or (None, why)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on this box"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_traffic import per_kernel
    tmp = tempfile.mkdtemp(prefix="storm_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", os.path.join(tmp, c), "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--N", "2", "--no-cpu-baseline", "--no-roofline", "--no-other-configs", "--no-h2d"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(tmp, c)) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {c} failed (rc {r.returncode})"
            got[c] = per_kernel(files[0], c)
    except Exception as e:  # noqa: BLE001
        return None, f"rocprofv3 pass failed: {type(e).__name__}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    table = {}
    for name, (fetch_kib, n) in got["FETCH_SIZE"].items():
        write_kib = got["WRITE_SIZE"].get(name, (0.0, 0))[0]
        table[name.split("(")[0].replace("void ", "").replace(" ", "")] = {"launches": n, "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0}
    return table, "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2 per the gfx950 correction) over a four-evaluation " \
                  "run of this bench command, average per launch of the kernel"


class PowerSampler:
    """Socket power / shader clock beside the timed steps (rocm-smi, one sample per ~0.1 s in a host thread; nothing when rocm-smi is
    missing): the convolutions of this workload run AT the 1400 W cap (profiles/r04_power_probe.txt), so throughput per watt is the
    quantity a kernel change can move."""

    def __init__(self, enabled=True):
        import shutil
        import threading
        self.samples, self.stop = [], False
        self.thread = threading.Thread(target=self._run, daemon=True) if enabled and shutil.which("rocm-smi") else None

    def _run(self):
        import re
        import subprocess
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                w = re.search(r"Power \(W\):\s*([\d.]+)", out)
                c = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
                if w:
                    self.samples.append((float(w.group(1)), int(c.group(1)) if c else -1))
            except Exception:  # noqa: BLE001
                pass
            time.sleep(0.2)

    def __enter__(self):
        if self.thread:
            self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.thread:
            self.thread.join(timeout=10)

    def summary(self):
        hot = sorted(s for s in self.samples if s[0] > 0.6 * POWER_CAP_W)      # (samples taken while the GPU was loaded)
        if not hot:
            return None
        return {"median_w": hot[len(hot) // 2][0], "max_w": hot[-1][0], "median_sclk_mhz": sorted(h[1] for h in hot)[len(hot) // 2],
                "samples": len(hot), "cap_w": POWER_CAP_W}


def selftest_cpu(args, rank, world):
    """Launch / sharding / timing skeleton on CPU ranks (gloo) with a stand-in step: what tests/test_distributed.py runs."""
    import torch.distributed as dist
    from storm_amd import distributed as D
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    mine = D.shard_indices(args.batch * world, rank, world)

    def step(i):
        time.sleep(0.02 * (1 + rank))                      # ranks differ: the reported time must be the slowest rank's
        return len(mine)

    elapsed, per_rank, n = D.timed_steps(step, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": args.batch * world * args.steps / elapsed, "unit": "utterances/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                          "selftest": True, "per_rank_s": per_rank, "utterances_per_rank": n}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher (`python bench.py --gpus N`): become N ranks of this node, one per GPU
        from storm_amd.distributed import self_launch
        self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} rank(s)"
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from storm_amd import distributed as D
    pinned = D.pin_to_gpu_numa(local) if world > 1 else None    # (one rank: the whole host is ours)
    dist = None
    if world > 1 or args.dist_world1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus

    from storm_amd.model import ScoreModel
    model = ScoreModel(backbone=args.backbone, sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5,
                       spec_factor=0.15, spec_abs_exponent=0.5)
    randomize(model, seed=0)
    model._error_loading_ema = True
    model.eval()
    model = model.to(dev)
    model.set_precision(args.precision)
    model.dnn.set_graph({"auto": -1, "on": 1, "off": 0}[args.graph])

    L = int(args.seconds * 16000)
    g = torch.Generator().manual_seed(1234 + rank)
    wav = (0.1 * torch.randn(args.batch, L, generator=g)).to(dev)          # inputs resident in HBM

    skw = dict(sampler_type="pc", predictor="reverse_diffusion", corrector=args.corrector, N=args.N, corrector_steps=args.corrector_steps,
               snr=0.5) if args.sampler == "pc" else dict(sampler_type="ode", N=args.N, compact=not args.ode_idle)
    units = args.batch
    if args.stream:                                         # configs[4]: ragged micro-batches (2-10 s), resident in HBM
        batches, _ = ragged_stream(args.stream, args.batch, 77 + rank, g, dev)
        units = args.stream

    def step(i):
        if not args.stream:
            return model.enhance_batch(wav, seed=1000 * rank + i, return_nfe=True, **skw)
        # one stream (kernels of concurrent queues corrupt each other on this platform, profiles/r06_concurrent_repro.txt); the micro-batches
        # share LAUNCHES instead: ScoreModel.enhance_stream runs their samplers in lockstep around one grouped network call per step
        outs, n_ = model.enhance_stream(batches, grouped=not args.no_group, seed=1000 * rank + 100 * i, return_nfe=True, width=args.width, **skw)
        return outs[-1], n_                                 # mean score evaluations per utterance

    # socket power: by default NOT sampled inside the timed region (the sampler spawns rocm-smi every 0.2 s on rank 0's host cores and
    # would perturb that rank only); one more repetition of the same step is sampled after it, on a one-rank run, where rocm-smi's
    # first GPU is this rank's device.  --power samples the timed steps themselves.
    with PowerSampler(enabled=args.power and rank == 0 and world == 1) as power:
        elapsed, per_rank, (out, nfe) = D.timed_steps(step, args.steps, args.warmup, sync=torch.cuda.synchronize, single_rank_group=args.dist_world1)
    power = power.summary() if rank == 0 else None
    assert torch.isfinite(out).all(), "non-finite output"
    value = units * world * args.steps / elapsed
    h2d = None
    if not args.no_h2d and not args.stream:                 # the same step from host memory: H2D + sampler + D2H inside the timed region
        wav_host = wav.cpu().pin_memory()
        out_host = torch.empty(out.shape, dtype=out.dtype).pin_memory()

        def step_host(i):
            o, n_ = model.enhance_batch(wav_host.to(dev, non_blocking=True), seed=1000 * rank + i, return_nfe=True, **skw)
            out_host.copy_(o, non_blocking=True)
            return o, n_
        el_h, _, _ = D.timed_steps(step_host, args.steps, 0, sync=torch.cuda.synchronize, single_rank_group=args.dist_world1)
        h2d = {"value_from_host_wavs": units * world * args.steps / el_h, "ms_per_step": 1e3 * el_h / args.steps,
               "bytes_per_step": 2 * args.batch * L * 4, "note": "pinned host wavs -> H2D -> sampler -> D2H of the enhanced wavs, all inside the timed region"}

    cfg_name = {("ncsnpp", 4.0, 30): "configs[1]" if world == 1 else "configs[2]", ("ncsnpplarge", 8.0, 50): "configs[3]"}.get(
        (args.backbone, float(args.seconds), args.N), "custom")
    result = {
        "metric": METRIC, "value": value, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": (f"configs[4]-style: {args.backbone} stream of {args.stream} utterances of 2-10 s@16 kHz per GPU in "
                                f"{len(batches)} ragged micro-batches (<= {args.batch}, bucketed by padded frame count; " + ("one after the other" if args.no_group else "grouped score evaluations") + "), " if args.stream else
                                f"{cfg_name}: {args.backbone} batch={args.batch}x{args.seconds:g} s@16 kHz per GPU, ") +
                               (f"{args.N}-step PC sampler (reverse_diffusion + {args.corrector} x{args.corrector_steps}), " if args.sampler == "pc"
                                else "probability-flow ODE sampler (RK45, rtol = atol = 1e-5), ") +
                               f"{args.precision} operands, wav->wav incl. STFT/iSTFT, inputs resident in HBM "
                               f"(H2D + D2H of {2 * args.batch * L * 4 / 1e6:.1f} MB per step would add < 0.01 %)",
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world, "seconds": args.seconds,
                   "pc_steps": args.N, "nfe_per_utterance": nfe, "parallelism": f"utterance-sharded x{world}"},
        "nfe_per_s": value * nfe, "ms_per_nfe_batch": None if args.stream else 1e3 * elapsed / args.steps / nfe,
        "per_rank_s": [round(t, 4) for t in per_rank],
        # the reference's own performance figure (model.py:304-308): processing time / audio duration.  `rtf` = latency of a call over the
        # duration of ONE of its utterances (the whole batch returns together); `rtf_per_audio_second` = GPU-seconds per second of audio
        "rtf": None if args.stream else elapsed / args.steps / args.seconds,
        "rtf_per_audio_second": None if args.stream else elapsed / args.steps / (args.seconds * args.batch),
        "graph": {"mode": args.graph, "hip_graph_launches": model.dnn.graph_launches()},
        # --stream: (grouped network calls, rows they evaluated) of the last step - with the ODE sampler's row compaction the rows are
        # the evaluations the utterances needed, not micro-batch size x the slowest row's count
        "grouped_calls_rows": list(getattr(model, "last_group_calls", None) or ()) if args.stream else None,
        # what the barrier / gather lines of the timed region ran through: the RCCL group the launcher's ranks formed, or nothing (one rank)
        "process_group": ({"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist is not None
                          else {"backend": None, "world_size": 1}),
        "from_host": h2d,
    }
    if pinned is not None:
        result["cpu_affinity"] = {"numa_cpus": len(pinned), "first": pinned[0], "last": pinned[-1]}
    if dist is not None:
        # every rank leaves the group HERE, right behind the timed region's closing fence: ranks 1.. exit, rank 0 profiles alone
        # (parked in an RCCL barrier they would spin on their GPUs for as long as rank 0's per-op profile takes)
        dist.barrier()
        dist.destroy_process_group()
        dist = None

    if rank == 0 and world == 1 and not args.power and not args.no_roofline and not args.stream:
        with PowerSampler(enabled=True) as ps:               # untimed repetitions of the same step, sampled
            for i in range(2):
                step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
        power = ps.summary()
        if power is not None:
            power["sampled"] = "two untimed repetitions of the step after the timed region"
    if rank == 0 and not args.no_roofline and not args.stream:
        Y, _, _ = model._prepare(wav)
        rows = profile_ops(model.dnn, Y, args.profile_nfe)
        groups = {}                                        # 3x3 convolutions on the matrix cores, by the kernel the launcher picked
        for r in rows:
            if r["code"] == 4 and r["big"] and 9 in r["taps"]:
                groups.setdefault(r["kernel"], []).append(r)
        kname, big = max(groups.items(), key=lambda kv: sum(r["ms"] for r in kv[1]))
        flops, ms = sum(r["flops"] for r in big), sum(r["ms"] for r in big)
        total_ms = sum(r["ms"] for r in rows)
        all_conv = [r for r in rows if r["code"] == 4]
        all3 = [r for g in groups.values() for r in g]
        peak = F32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16_MFMA_PEAK_TFLOPS     # (fp16 MFMA peak = bf16 peak)
        ach = flops / (ms * 1e-3) / 1e12
        names = {0: "memset", 1: "pack_input", 2: "temb", 3: "dense", 4: "conv", 5: "gn_stats", 6: "gn_apply",
                 7: "fir_up", 8: "fir_down", 9: "softmax", 10: "output_head", 11: "gn_finalize", 12: "attention", 13: "input_pyramid", 14: "output_pyramid"}
        by_kind = {}
        for r in rows:
            by_kind[names.get(r["code"], str(r["code"]))] = by_kind.get(names.get(r["code"], str(r["code"])), 0.0) + r["ms"]
        traffic, tsrc, live = None, None, None
        tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")     # PMC passes of scripts/pmc_bench.sh, per launch
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        if world == 1 and not args.no_traffic and cfg_name == "configs[1]":
            t0 = time.perf_counter()
            live, why = measure_traffic()
            why += f" ({time.perf_counter() - t0:.0f} s)"
            if live is not None:
                traffic = live.get(kname.replace(" ", ""), {}).get("hbm_bytes_per_launch")
                tsrc = why
        if traffic is None and tj:
            traffic = tj.get(kname.replace("storm::", "").replace(" ", ""), {}).get("hbm_bytes_per_launch")
            tsrc = "profiles/conv_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2 per the " \
                   "gfx950 correction) over this bench command, average per launch of this kernel; not re-measured in this run" + \
                   (f" ({why})" if not args.no_traffic and world == 1 and cfg_name == "configs[1]" else "")
        # socket power of each kernel family running sustained (tools/power_probe.py, profiles/r04_power_probe.txt: one kernel back to back for 4 s)
        probe_w = {"conv_pipe_kernel": 1397.0, "conv_pipe128_kernel": 1328.0, "conv_igemm_kernel": 1371.0, "conv_pipe_splitk_kernel": None}
        by_kernel = {}
        for k, v in groups.items():
            tf = sum(r["flops"] for r in v) / (sum(r["ms"] for r in v) * 1e-3) / 1e12
            w = next((pw for nm, pw in probe_w.items() if nm in k), None)
            by_kernel[k] = {"launches_per_nfe": len(v), "ms_per_nfe": round(sum(r["ms"] for r in v), 3), "tflops": round(tf, 1),
                            "probe_w": w, "tflops_per_kw": None if not w else round(tf / (w / 1e3), 1),
                            "pj_per_flop": None if not w else round(w / tf, 3)}
        result["roofline"] = {
            "bound": "mfma", "kernel": kname,
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": tsrc,
            "launches_per_nfe": len(big), "avg_launch_ms": ms / len(big), "avg_launch_gflop": flops / len(big) / 1e9,
            "avg_launch_algorithmic_bytes": sum(r["algorithmic_bytes"] for r in big) / len(big),
            "nfe_ms_profiled": total_ms, "ms_by_op_kind": {k: round(v, 3) for k, v in by_kind.items()},
            "conv3x3_by_kernel": by_kernel,
            # the dominant kernel's launches by layer resolution: the few-tile levels (< 512 pixel tiles of 8 x 32: one workgroup's serial
            # K loop bounds them, not the matrix pipe) pull its launch average down - `achieved` / `frac` above are over ALL its launches
            "dominant_by_resolution": {f"{h}x{w}": {"launches_per_nfe": len(v), "ms_per_nfe": round(sum(r["ms"] for r in v), 3),
                                                    "tflops": round(sum(r["flops"] for r in v) / (sum(r["ms"] for r in v) * 1e-3) / 1e12, 1)}
                                       for (h, w), v in sorted({(r["H"], r["W"]): [q for q in big if (q["H"], q["W"]) == (r["H"], r["W"])] for r in big}.items(),
                                                               reverse=True)},
            # the second denominator: what the part sustains at its 1400 W cap for this operand mix (POWER_BOUND_TFLOPS above)
            "power_bound": None if POWER_BOUND_TFLOPS[args.precision] is None else {
                "tflops_at_cap": POWER_BOUND_TFLOPS[args.precision], "cap_w": POWER_CAP_W, "frac": ach / POWER_BOUND_TFLOPS[args.precision],
                "source": "tools/ubench/mfma_rate.hip on random operands with the convolutions' 6 LDS fragment reads per 8 MFMAs, sustained at the "
                          "socket cap (profiles/r02_ubench.txt, r03_ubench.txt); the kernels of this evaluation run at 1328 - 1400 W "
                          "(profiles/r04_power_probe.txt)"},
            "power": power,
            "evaluation_tflops_per_kw": None if not power else round(sum(r.get("flops", 0) for r in rows) / (total_ms * 1e-3) / 1e12 / (power["median_w"] / 1e3), 1),
            "all_3x3_tflops": sum(r["flops"] for r in all3) / (sum(r["ms"] for r in all3) * 1e-3) / 1e12,
            "all_conv_tflops": sum(r["flops"] for r in all_conv) / (sum(r["ms"] for r in all_conv) * 1e-3) / 1e12,
            "method": f"HIP events per op on the launch stream (storm_program_run_timed) over {args.profile_nfe} score "
                      f"evaluations at batch {args.batch}; algorithmic FLOPs = 2*B*H*W*Cout*Cin*taps per launch; kernel names from "
                      f"the launcher (storm_program_kernel_name).  An event bracket around a single launch includes the dispatch / drain gap "
                      f"of the packets around it (about 20-25 us here): rocprofv3 --kernel-trace of the same command (profiles/) reports "
                      f"durations about 4 % shorter, so `achieved` is the conservative one of the two",
        }
        # the HBM-bound family (north_star: "fraction of the HBM/MFMA roofline"): GroupNorm-apply / resample kernels and the
        # narrow convolutions (<= 8 output or input channels: padded MFMA tiles whose time is operand traffic)
        hbm_rows = {}
        for r in rows:
            if "algorithmic_bytes" not in r:
                continue
            if r["code"] == 4 and r["big"] and min(r["cin"]) > 8:
                continue                                      # matrix-core bound convolutions: reported above
            key = r["kernel"] if r["code"] != 4 else ("narrow conv: " + ("3x3" if 9 in r["taps"] else "1x1") + f" {min(r['cin'])}->{r['Cout']}")
            hbm_rows.setdefault(key, []).append(r)
        if hbm_rows:
            tbl = {k: {"launches_per_nfe": len(v), "ms_per_nfe": round(sum(r["ms"] for r in v), 3),
                       "algorithmic_mb": round(sum(r["algorithmic_bytes"] for r in v) / 1e6, 1),
                       "tb_per_s": round(sum(r["algorithmic_bytes"] for r in v) / (sum(r["ms"] for r in v) * 1e-3) / 1e12, 2)}
                   for k, v in hbm_rows.items()}
            hk, hv = max(hbm_rows.items(), key=lambda kv: sum(r["ms"] for r in kv[1]))
            hb, hms = sum(r["algorithmic_bytes"] for r in hv), sum(r["ms"] for r in hv)
            htraffic = None
            if live is not None:
                htraffic = live.get(hk.replace(" ", ""), {}).get("hbm_bytes_per_launch")
            if htraffic is None and tj:
                tkey = "void" + hk.replace(" ", "")                       # (rocprofv3 prints template kernels with their return type)
                htraffic = next((v.get("hbm_bytes_per_launch") for k, v in tj.items() if k.replace(" ", "") in (tkey, tkey[4:])), None)
            result["roofline_hbm"] = {
                "bound": "hbm", "kernel": hk, "achieved": hb / (hms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hb / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": htraffic, "launches_per_nfe": len(hv),
                "avg_launch_ms": hms / len(hv), "avg_launch_algorithmic_bytes": hb / len(hv),
                "by_kernel": tbl,
                "family_ms_per_nfe": round(sum(r["ms"] for v in hbm_rows.values() for r in v), 3),
                "method": "the dominant HBM-bound kernel of the evaluation by time; algorithmic bytes = every input element read once + "
                          "every output element written once (op level, as BASELINE.md section 3 counts them) / HIP-event time of its "
                          "launches; peak = the 8 TB/s HBM3E specification (6.29 TB/s is the measured copy ceiling)",
            }
        if args.ops_json:
            with open(args.ops_json, "w") as f:
                json.dump(rows, f, indent=0)

    if rank == 0 and world == 1 and not args.no_other_configs and not args.stream and cfg_name == "configs[1]":
        t0 = time.perf_counter()
        result["other_configs"] = other_configs(model, dev, torch.cuda.synchronize)
        result["other_configs"]["seconds_spent"] = round(time.perf_counter() - t0, 1)
        model.set_precision(args.precision)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        dt, times, cores = cpu_baseline(args.backbone, args.seconds)
        result["cpu_baseline"] = {
            "value": 1.0 / (dt * nfe), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": f"score evaluations (1 of the {nfe} per utterance each) of one {args.seconds:g}-s utterance with the CPU oracle "
                      f"(PyTorch fp32): 1 warm-up + median of {len(times)} = {dt:.1f} s "
                      f"(runs: {', '.join(f'{t:.1f}' for t in times)} s), extrapolated x{nfe}",
            "s_per_nfe": dt,
        }

    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
