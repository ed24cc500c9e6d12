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
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--backbone", default="ncsnpp")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--profile-nfe", type=int, default=2, help="score evaluations profiled with per-op HIP events")
    p.add_argument("--ops-json", default=None, help="write the per-op timing table of the profiled pass here")
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


def cpu_baseline(backbone, seconds):
    """One score evaluation of one utterance with the CPU oracle (PyTorch fp32, all host cores)."""
    from oracle import ncsnpp_ref as NR
    cfg = NR.NCSNppConfig(**NR.NAMED_CONFIGS[backbone], input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=0)
    T = (1 + int(seconds * 16000) // 128 + 63) // 64 * 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 2, 256, T, dtype=torch.complex64, generator=g) * 0.3
    t = torch.tensor([0.5])
    with torch.no_grad():
        t0 = time.perf_counter()
        NR.ncsnpp_forward(sd, cfg, x, t)
        dt = time.perf_counter() - t0
    return dt, torch.get_num_threads()


def profile_ops(net, Y, nfe_count):
    """Per-op HIP-event timing of `nfe_count` score evaluations (same batch / shapes as the timed steps)."""
    from storm_amd import _lib as L
    from storm_amd.backbones.plan import BUF_IN0, BUF_OUT, BUF_PARAMS, BUF_T, BUF_WS, N_BUFS
    B, _, F, T = Y.shape
    dev = Y.device
    code = L.dt(net.compute_dtype)
    _, arena = net._get_arena(code, dev)
    prog, ws = net._get_program(B, F, T, code, dev)
    x = torch.randn_like(Y)
    out = torch.empty_like(Y)
    tvec = torch.full((B,), 0.5, device=dev)
    bufs = (C.c_void_p * N_BUFS)()
    bufs[BUF_WS], bufs[BUF_PARAMS] = ws.data_ptr(), arena.data_ptr()
    bufs[BUF_IN0], bufs[BUF_IN0 + 1] = torch.view_as_real(x[:, 0].contiguous()).data_ptr(), torch.view_as_real(Y[:, 0].contiguous()).data_ptr()
    xin, yin = x[:, 0].contiguous(), Y[:, 0].contiguous()
    bufs[BUF_IN0], bufs[BUF_IN0 + 1] = torch.view_as_real(xin).data_ptr(), torch.view_as_real(yin).data_ptr()
    bufs[BUF_T], bufs[BUF_OUT] = tvec.data_ptr(), torch.view_as_real(out).data_ptr()
    n = len(prog.ops)
    acc = [0.0] * n
    ms = (C.c_float * n)()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(nfe_count):
        L.check(L.lib().storm_program_run_timed(prog.op_array, n, bufs, N_BUFS, code, st, ms), "storm_program_run_timed")
        for k in range(n):
            acc[k] += ms[k] / nfe_count
    rows = []
    for k, op in enumerate(prog.ops):
        row = dict(idx=k, code=int(op.code), ms=acc[k])
        if op.code == 4:
            nseg, Bq, H, W, outC, Cout = [int(op.i[j]) for j in range(6)]
            flops, taps = 0, []
            for gseg in range(nseg):
                q = 8 + 7 * gseg
                flops += 2 * Bq * H * W * Cout * (int(op.i[q]) + int(op.i[q + 1])) * int(op.i[q + 4])
                taps.append(int(op.i[q + 4]))
            row.update(flops=flops, H=H, W=W, Cout=Cout, taps=taps, big=outC > 32, cin=[int(op.i[8]) + int(op.i[9])])
        rows.append(row)
    return rows, prog


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from storm_amd.model import ScoreModel
    model = ScoreModel(backbone=args.backbone, sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5,
                       spec_factor=0.15, spec_abs_exponent=0.5)
    randomize(model, seed=0)
    model._error_loading_ema = True
    model.eval()
    model = model.to(dev)
    model.set_precision(args.precision)

    L = int(args.seconds * 16000)
    g = torch.Generator().manual_seed(1234 + rank)
    wav = (0.1 * torch.randn(args.batch, L, generator=g)).to(dev)          # inputs resident in HBM

    def step(i):
        return model.enhance_batch(wav, predictor="reverse_diffusion", corrector=args.corrector, N=args.N,
                                   corrector_steps=args.corrector_steps, snr=0.5, seed=1000 * rank + i, return_nfe=True)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    nfe = None
    for i in range(args.warmup):
        _, nfe = step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out, nfe = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all(), "non-finite output"
    value = args.batch * world * args.steps / elapsed

    result = {
        "metric": METRIC, "value": value, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"configs[1]: {args.backbone} batch={args.batch}x{args.seconds:g} s@16 kHz per GPU, "
                               f"{args.N}-step PC sampler (reverse_diffusion + {args.corrector} x{args.corrector_steps}), "
                               f"{args.precision} operands, wav->wav incl. STFT/iSTFT",
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world, "seconds": args.seconds,
                   "pc_steps": args.N, "nfe_per_utterance": nfe, "parallelism": f"utterance-sharded x{world}"},
        "nfe_per_s": value * nfe, "ms_per_nfe_batch": 1e3 * elapsed / args.steps / nfe,
    }

    if rank == 0 and not args.no_roofline:
        Y, _, _ = model._prepare(wav)
        rows, prog = profile_ops(model.dnn, Y, args.profile_nfe)
        # the 3x3 implicit-GEMM convolution has two kernels (dispatch rule of conv_igemm.hip): the pipelined
        #   256 cout x 256 px kernel of conv_pipe.hip when outC > 128 and >= 512 pixel tiles, else 128 cout x 256 px
        tname = "storm::bf16_t" if args.precision == "bf16" else "float"
        groups = {}
        for r in rows:
            if r["code"] == 4 and r["big"] and 9 in r["taps"]:
                v2 = r["Cout"] > 128 and args.batch * ((r["H"] * r["W"] + 255) // 256) >= 512
                key = ("storm::conv_pipe_kernel<256, 128, 4, 2, 0>" if v2 and args.precision == "bf16"
                       else f"storm::conv_igemm_kernel<{tname}, 9, 2, 4, 2, true, false, 0>" if v2
                       else f"storm::conv_igemm_kernel<{tname}, 9, 2, 2, 2, false, false, 0>")
                groups.setdefault(key, []).append(r)
        kname, big = max(groups.items(), key=lambda kv: sum(r["ms"] for r in kv[1]))
        flops, ms = sum(r["flops"] for r in big), sum(r["ms"] for r in big)
        total_ms = sum(r["ms"] for r in rows)
        all_conv = [r for r in rows if r["code"] == 4]
        all3 = [r for g in groups.values() for r in g]
        peak = BF16_MFMA_PEAK_TFLOPS if args.precision == "bf16" else F32_MFMA_PEAK_TFLOPS
        ach = flops / (ms * 1e-3) / 1e12
        names = {0: "memset", 1: "pack_input", 2: "temb", 3: "dense", 4: "conv", 5: "gn_stats", 6: "gn_apply",
                 7: "fir_up", 8: "fir_down", 9: "softmax", 10: "output_head", 11: "gn_finalize"}
        by_kind = {}
        for r in rows:
            by_kind[names[r["code"]]] = by_kind.get(names[r["code"]], 0.0) + r["ms"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")     # PMC pass (scripts/pmc_round.sh), per launch
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(kname.replace("storm::", "").replace(" ", ""), {}).get("hbm_bytes_per_launch")
        result["roofline"] = {
            "bound": "mfma", "kernel": kname,
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
            "launches_per_nfe": len(big), "avg_launch_ms": ms / len(big), "avg_launch_gflop": flops / len(big) / 1e9,
            "nfe_ms_profiled": total_ms, "ms_by_op_kind": {k: round(v, 3) for k, v in by_kind.items()},
            "all_3x3_tflops": sum(r["flops"] for r in all3) / (sum(r["ms"] for r in all3) * 1e-3) / 1e12,
            "all_conv_tflops": sum(r["flops"] for r in all_conv) / (sum(r["ms"] for r in all_conv) * 1e-3) / 1e12,
            "method": f"HIP events per op on the launch stream (storm_program_run_timed) over {args.profile_nfe} score "
                      f"evaluations at batch {args.batch}; algorithmic FLOPs = 2*B*H*W*Cout*Cin*taps per launch",
        }
        if args.ops_json:
            with open(args.ops_json, "w") as f:
                json.dump(rows, f, indent=0)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        dt, cores = cpu_baseline(args.backbone, args.seconds)
        result["cpu_baseline"] = {
            "value": 1.0 / (dt * nfe), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": f"1 score evaluation (of {nfe}) of one {args.seconds:g}-s utterance with the CPU oracle "
                      f"(PyTorch fp32): {dt:.1f} s, extrapolated x{nfe}",
            "s_per_nfe": dt,
        }

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()                     # rank 0 profiles after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
