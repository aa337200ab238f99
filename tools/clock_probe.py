#!/usr/bin/env python3
"""What clock and board power does the chip hold under K5?  Loops one K5 kernel for a few seconds while a thread samples
`rocm-smi --showclocks --showpower --json` (and hwmon power if readable): the 64-row and the 32-row kernel, sparse R2 and dense."""
import ctypes, glob, json, os, subprocess, sys, threading, time
os.environ["RSA_TUNING"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectified_spaattn_amd import _lib
from bench import REGIMES, WORKLOADS, gen_inputs, make_neighbors, make_spec
from rectified_spaattn_amd import _core

def sample():
    out = {}
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
        j = json.loads(r.stdout)
        c = j.get("card0", {})
        for k, v in c.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "power" in kl or "fclk" in kl:
                out[k] = v
    except Exception as e:
        out["err"] = str(e)[:80]
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")[:1]:
        try: out["hwmon_W"] = int(open(f).read()) / 1e6
        except Exception: pass
    return out

def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    wl = WORKLOADS["hunyuan_720p_128f"]; spec = make_spec(wl)
    cent, nbr_kind, p = REGIMES["r2"]
    q, k, v = gen_inputs(wl, 24, 0, dev, cent)
    call = _core.StagedCall(q, k, v, spec, wl["top_k"], p, make_neighbors(wl, spec, nbr_kind))
    call.select(); torch.cuda.synchronize()
    print("idle:", sample(), flush=True)
    Sd = 16384
    qd = torch.randn(1, 24, Sd, 128, device=dev).to(torch.bfloat16)
    def run_for(fn, secs, tag):
        samples, stop = [], False
        def th():
            while not stop:
                samples.append(sample()); time.sleep(0.3)
        t = threading.Thread(target=th); t.start()
        t0 = time.time(); n = 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        while time.time() - t0 < secs:
            for _ in range(10): fn()
            n += 10
            torch.cuda.synchronize()
        b.record(); torch.cuda.synchronize()
        stop = True; t.join()
        ms = a.elapsed_time(b) / n
        print(f"{tag}: {ms:.3f} ms/call over {n} calls; samples (last 4 of {len(samples)}):", flush=True)
        for s in samples[-4:]: print("    ", s, flush=True)
    for w64 in (1, 0, 1):   # the 64-rows-per-wave K5 (the product at head dim 128) and the 32-row kernel
        assert L.rsa_set_tuning(b"k5_w64", w64) == 0
        run_for(call.attend, float(os.environ.get("SECS", "5")), f"sparse R2, {'64' if w64 else '32'}-row K5")
        run_for(lambda: _core.dense_attention(qd, qd, qd), 3, f"dense 16k, {'64' if w64 else '32'}-row K5")
    L.rsa_set_tuning(b"k5_w64", 3)

if __name__ == "__main__":
    main()
