#!/usr/bin/env python3
"""Board power against instruction mix: runs tools/bin/micro_power once per variant for a few seconds and samples rocm-smi beside it
(read-only).  power_mix.py [seconds]   -> profiles/r04_power_probe.txt, second table"""
import os, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
SECS = sys.argv[1] if len(sys.argv) > 1 else "5"
LABEL = {"V": "v_fma_f64, vector-register operands", "S": "v_fma_f64, one operand from a streamed scalar register", "L": "V + one ds_read_b64 per four FMAs",
         "W": "V + a float64 store per lane every 128 FMAs", "M": "v_mfma_f64_16x16x4_f64 from registers"}


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    w = [l.split(":")[-1].strip() for l in out.splitlines() if "Power (W)" in l]
    c = [l.split("(")[-1].split(")")[0] for l in out.splitlines() if "sclk" in l]
    return (w[0] if w else "?", c[0] if c else "?")


print("idle:", smi(), flush=True)
for mode in "VSLWM":
    proc = subprocess.Popen([os.path.join(HERE, "bin", "micro_power"), mode, SECS], stdout=subprocess.PIPE, text=True)
    time.sleep(1.5)
    samples = []
    while proc.poll() is None:
        samples.append(smi())
        time.sleep(0.6)
    line = proc.stdout.read().strip()
    ws = [float(w) for w, _ in samples[:-1] if w != "?"] or [float("nan")]
    print(f"{LABEL[mode]:58s} {min(ws):6.0f}-{max(ws):4.0f} W   sclk {samples[0][1]:>8s}   {line}", flush=True)
    time.sleep(2.0)
