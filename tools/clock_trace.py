"""Shader clock / package power trace of the GPU while a command runs (the "power envelope" claim needs a
committed trace).  Samples the amdgpu hwmon files of every GPU the box exposes at --hz (default 20) from a background thread:

    python tools/clock_trace.py --out profiles/r03_clock_power.csv -- python bench.py --mode f16 --no-exact --no-secondary --no-cpu-baseline
    python tools/clock_trace.py --probe          # list the sysfs files found and one sample of each

CSV columns: t_s (since the command started), label, sclk_mhz, power_w, temp_c, mclk_mhz (empty where a file is missing).
`label` comes from markers the command prints on stderr as `##clock_trace <label>` (bench.py does around its timed region
when GRIP_CLOCK_MARKERS=1); without markers it stays "run".  Falls back to `rocm-smi --showclocks --showpower --json` (~3 Hz)
when no hwmon file is readable.  The sampler itself is importable: `ClockSampler().start() ... .stop() -> summary`.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def find_gpu_hwmon():
    """[{card, freq (Hz file), power (uW file), temp (mC file), mclk}] for every amdgpu card with a hwmon directory."""
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if _read(os.path.join(card, "device", "vendor")) != "0x1002":
            continue
        for hw in glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")):
            ent = {"card": os.path.basename(card), "dir": hw, "bdf": os.path.basename(os.path.realpath(os.path.join(card, "device")))}
            for key, names in (("freq", ["freq1_input"]), ("mclk", ["freq2_input"]), ("power", ["power1_average", "power1_input"]),
                               ("temp", ["temp1_input", "temp2_input"])):
                ent[key] = next((os.path.join(hw, n) for n in names if _read(os.path.join(hw, n)) not in (None, "")), None)
            out.append(ent)
    return out


def visible_gpu_bdf(index=0):
    """PCI address ("0000:05:00.0") of the GPU this process computes on.  A box may expose the hwmon files of EVERY GPU of the
    node while only one is visible to HIP: the trace must follow that one, not card0.  $GRIP_TRACE_BDF overrides."""
    if os.environ.get("GRIP_TRACE_BDF"):
        return os.environ["GRIP_TRACE_BDF"].lower()
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


class ClockSampler:
    """Background sampler of (sclk MHz, power W, temp C, mclk MHz) of the GPU at PCI address `bdf` (default: HIP device 0)."""

    def __init__(self, hz=20.0, bdf=None):
        self.hz = hz
        hw = find_gpu_hwmon()
        bdf = bdf or visible_gpu_bdf()
        self.bdf = bdf
        match = [e for e in hw if bdf is not None and e["bdf"].lower() == bdf]
        self.ent = match[0] if match else (hw[0] if len(hw) == 1 else None)     # never guess among several cards
        self.rows = []          # (t, label, sclk, power, temp, mclk)
        self.label = "run"
        self._stop = threading.Event()
        self._thread = None
        self.t0 = None

    def available(self):
        return self.ent is not None and self.ent.get("freq") is not None

    def sample(self):
        e = self.ent

        def num(path, scale):
            v = _read(path) if path else None
            try:
                return float(v) / scale
            except (TypeError, ValueError):
                return None
        return num(e["freq"], 1e6), num(e["power"], 1e6), num(e["temp"], 1e3), num(e["mclk"], 1e6)

    def _smi_sample(self):
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = d[sorted(d)[0]]
            sclk = next((float(v.strip("()Mhz")) for k, v in c.items() if "sclk" in k.lower() and "(" in str(v)), None)
            pw = next((float(v) for k, v in c.items() if "power" in k.lower() and str(v).replace(".", "", 1).isdigit()), None)
            return sclk, pw, None, None
        except Exception:
            return None, None, None, None

    def _loop(self):
        period = 1.0 / self.hz
        while not self._stop.is_set():
            t = time.perf_counter()
            s = self.sample() if self.available() else self._smi_sample()
            self.rows.append((t - self.t0, self.label) + tuple(s))
            dt = period - (time.perf_counter() - t)
            if dt > 0:
                self._stop.wait(dt)

    def start(self):
        self.t0 = time.perf_counter()
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2)
        return self.summary()

    def summary(self, label=None):
        rows = [r for r in self.rows if (label is None or r[1] == label) and r[2] is not None]
        if not rows:
            return None
        sclk = sorted(r[2] for r in rows)
        pw = [r[3] for r in rows if r[3] is not None]
        return {"samples": len(rows), "sclk_mhz_mean": sum(sclk) / len(sclk), "sclk_mhz_p10": sclk[len(sclk) // 10], "sclk_mhz_max": sclk[-1],
                "power_w_mean": sum(pw) / len(pw) if pw else None, "power_w_max": max(pw) if pw else None,
                "source": f"hwmon {self.ent['card']} {self.ent['bdf']}" if self.available() else "rocm-smi"}

    def write_csv(self, path):
        with open(path, "w") as f:
            f.write("t_s,label,sclk_mhz,power_w,temp_c,mclk_mhz\n")
            for r in self.rows:
                f.write(",".join("" if v is None else (f"{v:.3f}" if isinstance(v, float) else str(v)) for v in r) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/clock_power.csv")
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    if a.probe:
        print("HIP device 0 is at", visible_gpu_bdf())
        for e in find_gpu_hwmon():
            print(e["card"], e["bdf"], {k: _read(v) for k, v in e.items() if k not in ("card", "dir", "bdf") and v})
        if not find_gpu_hwmon():
            print("no amdgpu hwmon directory visible; rocm-smi fallback:", ClockSampler()._smi_sample())
        return
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        ap.error("give a command after --")
    s = ClockSampler(a.hz).start()
    env = dict(os.environ, GRIP_CLOCK_MARKERS="1")
    p = subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True, env=env)
    for line in p.stderr:
        if line.startswith("##clock_trace "):
            s.label = line.split(None, 1)[1].strip()
        else:
            sys.stderr.write(line)
    rc = p.wait()
    s.stop()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    s.write_csv(a.out)
    labels = []
    for r in s.rows:
        if r[1] not in labels:
            labels.append(r[1])
    print(json.dumps({"csv": a.out, "rc": rc, "by_label": {l: s.summary(l) for l in labels}}, indent=1))
    sys.exit(rc)


if __name__ == "__main__":
    main()
