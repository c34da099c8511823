"""Clock / power trace next to a command (DESIGN.md "Status & measurements": are the kernels running at the nominal clock?).

usage: python tools/power_trace.py <out.json> -- <command ...>
Samples the GPU while <command> runs and writes a summary + the raw samples.  Sources, in the order tried:
  1. sysfs of the first amdgpu card: hwmon freq1_input (sclk, Hz), power1_average / power1_input (uW), pp_dpm_sclk / pp_dpm_mclk
     (the line marked '*'), gpu_busy_percent -- a few hundred samples per second are possible;
  2. `rocm-smi --showclocks --showpower --json` (slow: a few samples per second) when sysfs is not readable.
Nothing here is on the product path; it is a measurement helper for the GPU box."""
import glob
import json
import os
import re
import subprocess
import sys
import time


def _read(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


def visible_bdf():
    """PCI address of the GPU this container can use (rocm-smi lists only that one), e.g. 0000:75:00.0"""
    try:
        out = subprocess.run(["rocm-smi", "--showbus", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        for v in j[sorted(j)[0]].values():
            m = re.search(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", str(v))
            if m:
                return m.group(0).lower()
    except Exception:  # noqa: BLE001
        pass
    return None


def find_card():
    """sysfs device directory of the visible GPU (the box's sysfs also shows the other tenants' cards)"""
    cards = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if _read(os.path.join(d, "vendor")) == "0x1002"]
    bdf = visible_bdf()
    for d in cards:
        if bdf and os.path.realpath(d).lower().endswith(bdf):
            return d
    return cards[0] if len(cards) == 1 else None


def dpm_current(txt):
    if not txt:
        return None
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            m = re.search(r"(\d+)\s*Mhz", line, re.I)
            if m:
                return int(m.group(1))
    return None


def sample_sysfs(dev, hw):
    s = {"t": time.time()}
    v = _read(os.path.join(hw, "freq1_input")) if hw else None
    if v:
        s["sclk_mhz"] = int(v) / 1e6
    for name in ("power1_average", "power1_input"):
        v = _read(os.path.join(hw, name)) if hw else None
        if v:
            s["power_w"] = int(v) / 1e6
            break
    c = dpm_current(_read(os.path.join(dev, "pp_dpm_sclk")))
    if c is not None:
        s.setdefault("sclk_mhz", float(c))
        s["dpm_sclk_mhz"] = c
    c = dpm_current(_read(os.path.join(dev, "pp_dpm_mclk")))
    if c is not None:
        s["mclk_mhz"] = c
    v = _read(os.path.join(dev, "gpu_busy_percent"))
    if v:
        s["busy"] = int(v)
    return s


def sample_smi():
    s = {"t": time.time()}
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        j = json.loads(out)
        card = j[sorted(j)[0]]
        for k, v in card.items():
            m = re.search(r"\((\d+)Mhz\)", str(v))
            if "sclk" in k.lower() and m:
                s["sclk_mhz"] = float(m.group(1))
            if "mclk" in k.lower() and m:
                s["mclk_mhz"] = float(m.group(1))
            if "power" in k.lower():
                try:
                    s["power_w"] = float(v)
                except (TypeError, ValueError):
                    pass
    except Exception as e:  # noqa: BLE001 -- a measurement helper: report and go on
        s["error"] = str(e)[:100]
    return s


def stats(xs):
    if not xs:
        return None
    xs = sorted(xs)
    n = len(xs)
    return {"n": n, "min": xs[0], "p10": xs[n // 10], "median": xs[n // 2], "p90": xs[(9 * n) // 10], "max": xs[-1], "mean": sum(xs) / n}


def main():
    if "--" not in sys.argv or len(sys.argv) < 4:
        sys.exit(__doc__)
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    dev = find_card()
    hw = None
    if dev:
        hws = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        hw = hws[0] if hws else None
    use_sysfs = bool(dev) and bool(sample_sysfs(dev, hw).keys() - {"t"})
    period = 0.02 if use_sysfs else 0.0
    t0 = time.time()
    proc = subprocess.Popen(cmd)
    samples = []
    while proc.poll() is None:
        samples.append(sample_sysfs(dev, hw) if use_sysfs else sample_smi())
        if period:
            time.sleep(period)
    rc = proc.returncode
    for s in samples:
        s["t"] = round(s["t"] - t0, 3)
    busy = [s for s in samples if s.get("busy", 100) >= 50 and s.get("power_w", 1e9) > 300]  # the loaded part of the run
    summ = {"command": " ".join(cmd), "rc": rc, "source": ("sysfs " + os.path.realpath(dev)) if use_sysfs else "rocm-smi", "samples": len(samples),
            "all": {k: stats([s[k] for s in samples if k in s]) for k in ("sclk_mhz", "mclk_mhz", "power_w", "busy")},
            "loaded (busy >= 50 %, power > 300 W)": {k: stats([s[k] for s in busy if k in s]) for k in ("sclk_mhz", "mclk_mhz", "power_w")}}
    json.dump({"summary": summ, "samples": samples[:: max(1, len(samples) // 2000)]}, open(out, "w"), indent=0)
    print(json.dumps(summ))
    sys.exit(rc)


if __name__ == "__main__":
    main()
