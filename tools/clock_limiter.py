#!/usr/bin/env python3
"""What holds the clock while the bucket accumulation runs?  

On an MI355X box:   python3 tools/clock_limiter.py [--log-len 26] [--seconds 20] > profiles/r05_clock_limiter.txt

1. `amd-smi metric` (all sections, JSON) BEFORE and AFTER a loop of resident table-mode MSMs (k_accumulate is ~90 % of it): the
   accumulated throttler / violation counters (package power PPT, socket / VR / HBM thermal, PROCHOT, and the per-XCD "gfx clock
   below host limit because of power / thermal / low utilisation" accumulators where the driver exposes them) -- their DELTAS over
   the loop name the limiter, instead of inferring it from a power reading 9 % under the cap.
2. A ~10 ms-resolution trace of sclk and socket power from the hwmon / pp_dpm sysfs files of the card (ordinary user, no root)
   while the loop runs: minimum / median / maximum, and the first 40 samples.
3. The raw gpu_metrics table's header (format / content revision) for whoever wants to decode it.
Nothing here changes a GPU setting."""
import argparse, glob, json, os, statistics, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--log-len", type=int, default=26)
ap.add_argument("--seconds", type=float, default=20.0)
a = ap.parse_args()


def smi(*args):
    try:
        r = subprocess.run(["amd-smi"] + list(args), capture_output=True, text=True, timeout=60)
        return r.stdout if r.returncode == 0 else "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])
    except Exception as e:      # noqa: BLE001
        return "failed: %r" % e


def flatten(obj, prefix=""):
    out = {}
    if isinstance(obj, dict):
        for k, v in obj.items():
            out.update(flatten(v, prefix + str(k) + "."))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            out.update(flatten(v, prefix + str(i) + "."))
    else:
        out[prefix[:-1]] = obj
    return out


def metric_snapshot():
    txt = smi("metric", "--json")
    try:
        return flatten(json.loads(txt)), None
    except Exception:           # noqa: BLE001
        return {}, txt[:2000]


def first(paths):
    for p in paths:
        for f in glob.glob(p):
            return f
    return None


hw = "/sys/class/drm/card*/device/hwmon/hwmon*/"
f_power = first([hw + "power1_input", hw + "power1_average"])
f_sclk = first([hw + "freq1_input"])
f_dpm = first(["/sys/class/drm/card*/device/pp_dpm_sclk"])
f_metrics = first(["/sys/class/drm/card*/device/gpu_metrics"])


def read_num(path, scale):
    try:
        return float(open(path).read().split()[0]) / scale
    except Exception:           # noqa: BLE001
        return None


def read_dpm():
    try:
        for line in open(f_dpm):
            if "*" in line:
                return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
    except Exception:           # noqa: BLE001
        pass
    return None


print("# tools/clock_limiter.py  --log-len %d  --seconds %.0f" % (a.log_len, a.seconds))
print("# sysfs: power %s | sclk %s | dpm %s | gpu_metrics %s" % (f_power, f_sclk, f_dpm, f_metrics))
print("== amd-smi version / limits")
print(smi("version").strip()[:300])
print(smi("static", "--limit").strip()[:1500])
if f_metrics:
    try:
        raw = open(f_metrics, "rb").read()
        print("== gpu_metrics: %d bytes, structure_size %d, format_revision %d, content_revision %d" % (len(raw), raw[0] | raw[1] << 8, raw[2], raw[3]))
    except Exception as e:      # noqa: BLE001
        print("== gpu_metrics unreadable: %r" % e)

before, err = metric_snapshot()
if err:
    print("== amd-smi metric --json did not parse:", err)
t0 = time.time()
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "msm_bench.py"), "--tables", "--log-len", str(a.log_len), "--reps", "100000"],
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
# the bases and their window tables are built first (tens of seconds at 2^26): wait until the socket draws > 1 kW for a second
t_wait, hot = time.time(), 0
while time.time() - t_wait < 240 and hot < 20:
    pw = read_num(f_power, 1e6) if f_power else None
    hot = hot + 1 if (pw or 0) > 1000 else 0
    time.sleep(0.05)
print("== load reached after %.1f s (socket power > 1 kW for a second: %s)" % (time.time() - t_wait, hot >= 20))
mid0, _ = metric_snapshot()
samples = []
t_loop = time.time()
while time.time() - t_loop < a.seconds:
    ts = time.time()
    samples.append((ts - t_loop, read_num(f_power, 1e6) if f_power else None, (read_num(f_sclk, 1e6) if f_sclk else None) or read_dpm()))
    dt = 0.010 - (time.time() - ts)
    if dt > 0:
        time.sleep(dt)
mid, _ = metric_snapshot()          # still under load: `mid0` -> `mid` brackets exactly the traced seconds
child.terminate()
try:
    child.wait(20)
except Exception:               # noqa: BLE001
    child.kill()

print("== trace: %d samples over %.1f s (%.1f ms apart)" % (len(samples), a.seconds, 1e3 * a.seconds / max(1, len(samples))))
for name, idx, unit in (("socket power", 1, "W"), ("sclk", 2, "MHz")):
    vals = [s[idx] for s in samples if s[idx] is not None]
    if vals:
        print("%-13s min %.0f  median %.0f  max %.0f %s   (n = %d)" % (name, min(vals), statistics.median(vals), max(vals), unit, len(vals)))
    else:
        print("%-13s not readable from sysfs on this box" % name)
print("first 40 samples (s, W, MHz):", [(round(t, 3), p and round(p), c and round(c)) for t, p, c in samples[:40]])

time.sleep(3.0)
after, _ = metric_snapshot()        # idle again
print("== amd-smi metric under load: per-XCD gfx clocks, power, temperatures, throttle status")
for k in sorted(mid):
    kl = k.lower()
    if (".clock.gfx_" in kl and kl.endswith("clk.value")) or "power" in kl or "temperature" in kl or ("throttle" in kl and "xcp" not in kl):
        print("  %-70s %s" % (k, mid[k]))
print("== accumulators that MOVED during the %.0f traced seconds under load (the deltas name the limiter)" % a.seconds)
moved = 0
for k in sorted(mid):
    x, y = mid0.get(k), mid[k]
    if isinstance(x, (int, float)) and isinstance(y, (int, float)) and x != y:
        kl = k.lower()
        if any(w in kl for w in ("acc", "thrott", "violation", "residency", "below_host", "count", "energy")):
            print("  %-70s %s -> %s   (delta %s)" % (k, x, y, y - x))
            moved += 1
if not moved:
    print("  none of the accumulator-like fields moved (or this amd-smi exposes none)")
print("== the same accumulators over 3 idle seconds afterwards (for scale: which of them count idleness, not throttling)")
for k in sorted(after):
    x, y = mid.get(k), after[k]
    if isinstance(x, (int, float)) and isinstance(y, (int, float)) and x != y and any(w in k.lower() for w in ("acc", "residency", "below_host")):
        print("  %-70s %s -> %s   (delta %s)" % (k, x, y, y - x))
print("== amd-smi metric --throttle (text, after the loop)")
print(smi("metric", "--throttle").strip()[:3000])
