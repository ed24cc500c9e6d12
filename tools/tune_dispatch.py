#!/usr/bin/env python
"""Re-tune the convolution dispatcher in ONE GPU visit: measure every 16-bit 3x3 layer of the planned networks
under every kernel family that can run it, with the alternating >= 20-repetition protocol (profiles/r04_duo_fair_ab.txt: a kernel
timed cold over five repetitions runs up to 20 % faster than sustained), and write the measured EXCEPTIONS to the rule ladder of
conv_igemm.hip:choose_variant as storm_amd/csrc/conv_dispatch_table.h.

    gpurun -- 'python tools/tune_dispatch.py --write > gpurun_out/tune_dispatch.log'      then rebuild (python -m storm_amd.build)

How: for every configuration (network, batch, frames) the C planner's op list is run op by op through storm_program_run on a
workspace filled with N(0, 1) values in the operand type (zeros would clock 20 % higher: LAB_NOTES 2.1), each 3x3 convolution with > 32
output channels under STORM_CONV_VARIANT = candidate, the candidates of a layer interleaved round-robin, `--reps` timed launches each
after a sustained warm-up of the whole candidate set.  Split-K layers are skipped (conv_splitk_slices decides them; the planner sizes
their scratch).  A candidate replaces the ladder's choice when it is faster by more than `--margin` (default 3 %) in BOTH halves of
the repetitions (guards against drift).  Layers that share (K, shortcut K, output channels, tiles per image) across batch sizes are
merged into batch ranges.  The full measurement table goes to stdout (kept under profiles/)."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import _lib as L  # noqa: E402
from storm_amd.backbones.ncsnpp import NCSNpp, NCSNppLarge  # noqa: E402
from storm_amd.backbones.plan import BUF_IN0, BUF_OUT, BUF_PARAMS, BUF_T, BUF_WS, N_BUFS  # noqa: E402

NAMES = {0: "conv_igemm 128-cout tile", 2: "conv_igemm 256-cout tile", 3: "conv_pipe<256,8>", 4: "conv_pipe128", 7: "conv_igemm 64-cout tile",
         9: "conv_pipe<128,8>"}


def candidates(outC):
    return [0, 4, 7, 9] if outC <= 128 else [3, 9, 2]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--reps", type=int, default=24)
    p.add_argument("--margin", type=float, default=0.03)
    p.add_argument("--keep", type=float, default=0.05, help="only exceptions of at least this gain go into the table (3 - 5 % ties between the two "
                   "<= 128-cout structures flip from box to box and from pass to pass)")
    p.add_argument("--write", action="store_true", help="write storm_amd/csrc/conv_dispatch_table.h")
    p.add_argument("--configs", default="ncsnpp:16:512,ncsnpp:8:512,ncsnpp:4:512,ncsnpp:2:512,ncsnpp:1:512,ncsnpplarge:8:1024,ncsnpp:3:1280,ncsnpp:2:768")
    p.add_argument("--json", default="", help="also dump the measurements here")
    args = p.parse_args()
    dev = torch.device("cuda:0")
    lib = L.lib()
    code = L.BF16
    L.check(lib.storm_set_switch(b"STORM_CONV_TABLE", 0), "table off")        # the baseline is the ladder
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    results = {}                                                              # key -> {B: {variant: ms, "ladder": v}}
    nets = {}
    for cfg in args.configs.split(","):
        name, B, T = cfg.split(":")
        B, T, F = int(B), int(T), 256
        if name not in nets:
            net = (NCSNppLarge if name == "ncsnpplarge" else NCSNpp)(input_channels=4).to(dev)
            net.set_compute_dtype(torch.bfloat16)
            nets[name] = net
        net = nets[name]
        h = net._get_handle(code, dev)
        ops, n, _ = net.program(B, F, T)
        ws = net._get_workspace(h, B, F, T, code, dev)
        g = torch.Generator(device=dev).manual_seed(1)
        chunk = 1 << 28                                                       # N(0, 1) 16-bit operands everywhere (read as fp32 tables: finite, ~N(0, 1) too)
        wsv = ws[:ws.numel() // 2 * 2].view(torch.bfloat16)
        for o in range(0, wsv.numel(), chunk):
            m = min(chunk, wsv.numel() - o)
            wsv[o:o + m] = torch.randn(m, generator=g, device=dev).to(torch.bfloat16)
        x = torch.randn(B, F, T, 2, generator=g, device=dev)
        x = torch.view_as_complex(x.contiguous())
        out = torch.empty_like(x)
        tv = torch.full((B,), 0.5, device=dev)
        bufs = (C.c_void_p * N_BUFS)()
        bufs[BUF_WS], bufs[BUF_PARAMS] = ws.data_ptr(), lib.storm_ncsnpp_arena(h)
        bufs[BUF_IN0], bufs[BUF_IN0 + 1] = torch.view_as_real(x).data_ptr(), torch.view_as_real(x).data_ptr()
        bufs[BUF_T], bufs[BUF_OUT] = tv.data_ptr(), torch.view_as_real(out).data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        op_size = C.sizeof(L.Op)
        base = C.addressof(ops.contents)
        for k in range(n):
            op = ops[k]
            if op.code != 4:
                continue
            nseg, Bq, H, W, outC = [int(op.i[j]) for j in range(5)]
            t0, t1 = int(op.i[12]), int(op.i[19]) if nseg == 2 else 0
            if t0 != 9 or outC <= 32 or (nseg == 2 and t1 != 1):
                continue
            k9 = int(op.i[8]) + int(op.i[9])
            k1 = int(op.i[15]) + int(op.i[16]) if nseg == 2 else 0
            if k9 < 32:
                continue                                                      # (stem-like layers: conv_thin's)
            ladder_name = lib.storm_program_kernel_name(ops, k, code).decode()
            if "splitk" in ladder_name or "thin" in ladder_name or "narrow" in ladder_name:
                continue
            key = (k9, k1, outC, ((H + 7) // 8) * ((W + 31) // 32))
            if key in results and Bq in results[key]:
                continue
            one = C.cast(base + k * op_size, C.POINTER(L.Op))
            runs = {}
            for v in candidates(outC):
                L.check(lib.storm_set_switch(b"STORM_CONV_VARIANT", v), "variant")
                nm = lib.storm_program_kernel_name(ops, k, code).decode()
                if nm in [r[0] for r in runs.values()]:
                    continue                                                  # this variant falls back to one already in the set
                runs[v] = [nm, [], []]
            ladder = next((v for v, r in runs.items() if r[0] == ladder_name), None)

            def launch(v, reps):
                L.check(lib.storm_set_switch(b"STORM_CONV_VARIANT", v), "variant")
                for _ in range(reps):
                    L.check(lib.storm_program_run(one, 1, bufs, N_BUFS, code, st), "storm_program_run")
            for v in runs:                                                    # sustained warm-up of the whole set
                launch(v, 6)
            for half in (1, 2):
                for _ in range(args.reps // 2):
                    for v in runs:
                        e0, e1 = ev(), ev()
                        e0.record()
                        launch(v, 1)
                        e1.record()
                        runs[v][half].append((e0, e1))
            torch.cuda.synchronize()
            L.check(lib.storm_set_switch(b"STORM_CONV_VARIANT", -1), "variant")
            med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
            ms = {v: (med([a.elapsed_time(b) for a, b in r[1]]), med([a.elapsed_time(b) for a, b in r[2]])) for v, r in runs.items()}
            results.setdefault(key, {})[Bq] = {"ms": ms, "ladder": ladder, "names": {v: r[0] for v, r in runs.items()}, "net": name, "hw": (H, W)}
            flops = 2.0 * Bq * H * W * int(op.i[5]) * (k9 * 9 + k1)
            line = f"{name:11s} B={Bq:2d} {H:3d}x{W:<4d} K9={k9:3d} K1={k1:3d} outC={outC:3d} tiles/img={key[3]:5d} ladder={ladder}: "
            line += " | ".join(f"v{v} {0.5 * (a + b):.3f} ms ({flops / (0.5 * (a + b)) / 1e9:5.0f} TF)" for v, (a, b) in ms.items())
            print(line, flush=True)
        net.release_program(ops)
    # ---- exceptions to the ladder --------------------------------------------------------------------------------------------
    entries = []
    for key, perB in sorted(results.items()):
        picks = {}
        for Bq, r in sorted(perB.items()):
            if r["ladder"] is None:
                continue
            base1, base2 = r["ms"][r["ladder"]]
            best = min(r["ms"], key=lambda v: sum(r["ms"][v]))
            b1, b2 = r["ms"][best]
            if best != r["ladder"] and b1 < base1 * (1 - args.margin) and b2 < base2 * (1 - args.margin):
                picks[Bq] = (best, 1 - (b1 + b2) / (base1 + base2))
        # merge neighbouring batch sizes with the same pick into ranges (only measured batch sizes: no extrapolation across an unmeasured ladder decision)
        for Bq, (v, gain) in sorted(picks.items()):
            if gain >= args.keep:
                entries.append((key, Bq, Bq, v, gain))
            else:
                print(f"  (below --keep: K9={key[0]} K1={key[1]} outC={key[2]} tiles/img={key[3]} B={Bq}: variant {v}, -{100 * gain:.1f} %)")
    print("\nexceptions to the ladder (margin %.0f %% in both halves of the repetitions):" % (100 * args.margin))
    for (k9, k1, outC, tiles), lo, hi, v, gain in entries:
        print(f"  K9={k9} K1={k1} outC={outC} tiles/img={tiles} B={lo}..{hi}: variant {v} ({NAMES.get(v, v)}), -{100 * gain:.1f} % time")
    if args.json:
        json.dump({f"{k}": {str(b): {"ms": {str(v): m for v, m in r["ms"].items()}, "ladder": r["ladder"], "net": r["net"]} for b, r in pb.items()}
                   for k, pb in results.items()}, open(args.json, "w"), indent=0)
    if args.write:
        src = os.path.join(ROOT, "storm_amd", "csrc", "conv_dispatch_table.h")
        text = open(src).read()
        head = text[:text.index("#pragma once")]
        body = "#pragma once\nnamespace storm {\nstruct DispatchEntry { int k9, k1, outC, tiles_img, images_lo, images_hi, variant; };\n"
        body += (f"// measured on {torch.cuda.get_device_name(0)}: {args.reps} interleaved repetitions per candidate, margin {100 * args.margin:.0f} %, "
                 f"configurations {args.configs}: {len(entries)} exceptions\n")
        body += "static const DispatchEntry kDispatchTable[] = {\n"
        for (k9, k1, outC, tiles), lo, hi, v, gain in entries:
            body += f"    {{{k9}, {k1}, {outC}, {tiles}, {lo}, {hi}, {v}}},      // {NAMES.get(v, v)}: -{100 * gain:.1f} % against the ladder\n"
        body += "    {0, 0, 0, 0, 0, 0, -1},\n};\n}  // namespace storm\n"
        for path in (src, os.path.join(ROOT, "gpurun_out", "conv_dispatch_table.h")):      # (gpurun_out/ is what travels back from the GPU box)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            open(path, "w").write(head + body)
            print("wrote", path)


if __name__ == "__main__":
    main()
