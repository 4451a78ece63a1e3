"""Measurement aid: run a command and sample sclk / mclk / fclk / power with rocm-smi while it runs.
usage: clock_watch.py <seconds-between-samples> -- cmd args..."""
import subprocess, sys, time, re
i = sys.argv.index("--")
dt = float(sys.argv[1])
p = subprocess.Popen(sys.argv[i + 1:])
while p.poll() is None:
    time.sleep(dt)
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    vals = {}
    for k in ("sclk", "mclk", "fclk", "socclk"):
        m = re.search(k + r" clock level: \d+: \((\d+)Mhz\)", out)
        if m: vals[k] = int(m.group(1))
    m = re.search(r"Power \(W\): ([\d.]+)", out)
    if m: vals["W"] = float(m.group(1))
    print("  [clocks]", vals, flush=True)
sys.exit(p.returncode)
