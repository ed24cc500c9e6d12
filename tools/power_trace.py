#!/usr/bin/env python
"""Sample socket power / shader clock / junction temperature (rocm-smi) every ~0.1 s into a CSV until terminated (SIGTERM from the
script that started it: `python tools/power_trace.py out.csv & PT=$!; <command>; kill $PT`); `--summary out.csv` prints the medians of the
samples above 60 % of the power cap - what the part does WHILE the bench runs (round 4: the convolutions run at the 1400 W cap)."""
import re
import signal
import subprocess
import sys
import time


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
    w = re.search(r"Power \(W\):\s*([\d.]+)", out)
    c = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    t = re.search(r"Temperature \(Sensor junction\) \(C\):\s*([\d.]+)", out)
    return (float(w.group(1)) if w else float("nan"), int(c.group(1)) if c else -1, float(t.group(1)) if t else float("nan"))


if sys.argv[1] == "--summary":
    rows = [l.strip().split(",") for l in open(sys.argv[2]) if l[0].isdigit()]
    hot = [(float(r[1]), int(r[2]), float(r[3])) for r in rows if float(r[1]) > 0.6 * 1400]
    med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")  # noqa: E731
    print(f"{len(rows)} samples, {len(hot)} above 840 W: median power {med([h[0] for h in hot]):.0f} W (max {max([h[0] for h in hot], default=0):.0f}), "
          f"median sclk {med([h[1] for h in hot])} MHz, junction {med([h[2] for h in hot]):.0f} C")
    sys.exit(0)
stop = False
signal.signal(signal.SIGTERM, lambda *a: globals().__setitem__("stop", True))
with open(sys.argv[1], "w") as f:
    f.write("t_s,power_w,sclk_mhz,junction_c\n")
    t0 = time.time()
    while not stop:
        try:
            w, c, t = sample()
            f.write(f"{time.time() - t0:.2f},{w},{c},{t}\n")
            f.flush()
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.05)
