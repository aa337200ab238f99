#!/usr/bin/env python3
"""Interleaved A/B of K5 across several BUILDS of librsa_hip.so loaded side by side in ONE process on ONE device
(cdna_hip_programming.md rule 24: never rank builds by timings taken on different devices).

    python tools/ab_libs.py name=path[:nbuf] ...   [--rounds 12] [--pmc]

Each library gets the same q, k, v and the same statistics / kept lists (computed once by the in-tree library);
`nbuf[:flags]` = how many rsa_buffers members that build's header declares (round 1: 14, round 2: 18, now: 15) -- the first 14
members never moved, the split-KV partial buffer `tpart` is the last member of the later ones.  Prints every round's time
per library, medians, minima and the max |difference| of the outputs against the first library.  --pmc: two launches per
library and nothing else (for rocprofv3 --pmc passes; kernels of different builds differ by template arguments).
"""
import ctypes
import os
import sys

os.environ.setdefault("RSA_TUNING", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import REGIMES, WORKLOADS, gen_inputs, make_neighbors, make_spec, regime_top_k  # noqa: E402
from rectified_spaattn_amd import _core, _lib  # noqa: E402
from rectified_spaattn_amd._lib import RsaLayout, RsaOut4, RsaTensor4  # noqa: E402

FIRST14 = ("qbar", "aq", "kbar", "ak", "vbar", "scores", "unrel", "probs", "w", "R", "comp", "bitmask", "cols", "counts")


def load(path, nbuf):
    L = ctypes.CDLL(os.path.abspath(path))
    # (builds since 0.5.0 read the capacity of tpart behind the 15 pointers; older builds never look there)
    fields = [(n, ctypes.c_void_p) for n in FIRST14] + [(f"x{i}", ctypes.c_void_p) for i in range(nbuf - 14)] + [("tpart_bytes", ctypes.c_size_t)]
    Buf = type(f"Buf{nbuf}", (ctypes.Structure,), {"_fields_": fields})
    P = ctypes.POINTER
    L.rsa_block_sparse_fwd.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, RsaTensor4, P(Buf), RsaOut4, ctypes.c_void_p]
    L.rsa_block_sparse_fwd.restype = ctypes.c_int
    L.rsa_dense_fwd.argtypes = [ctypes.c_int] * 6 + [RsaTensor4, RsaTensor4, RsaTensor4, ctypes.c_int, ctypes.c_int,
                                                     RsaOut4, ctypes.c_void_p]
    L.rsa_dense_fwd.restype = ctypes.c_int
    return L, Buf


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = 12
    if "--rounds" in sys.argv:
        rounds = int(sys.argv[sys.argv.index("--rounds") + 1])
        args = [a for a in args if a != str(rounds)]
    pmc = "--pmc" in sys.argv
    fp8 = "--fp8" in sys.argv      # K5 on the e4m3 images (rsa_block_sparse_fwd_fp8); builds with the current rsa_fp8_operands only
    dev = torch.device("cuda:0")
    H = int(os.environ.get("RSA_PERF_H", "24"))
    wl = WORKLOADS[os.environ.get("RSA_PERF_WORKLOAD", "hunyuan_720p_128f")]
    if "RSA_PERF_H" not in os.environ:
        H = wl["H"]
    spec = make_spec(wl)
    regime = os.environ.get("RSA_PERF_REGIME", "r2")
    cent, nbr_kind, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, H, 0, dev, cent)
    call = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, make_neighbors(wl, spec, nbr_kind), qkv_fp8=fp8)
    call.select()
    torch.cuda.synchronize()
    pairs = call.bufs["counts"].sum().item()
    flops = 4.0 * 128 * 128 * 128 * pairs + 4.0 * 128 * spec.q_text_valid * spec.kv_text_valid * H
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    libs = []
    for a in args:
        name, rest = a.split("=", 1)
        parts = rest.split(":")
        path, nbuf = parts[0], int(parts[1]) if len(parts) > 1 and parts[1] else 15
        flags = parts[2].split(",") if len(parts) > 2 else []   # o8: output rows 8-byte aligned only; nots: no split-KV; key=value: rsa_set_tuning before every call
        L, Buf = load(path, nbuf)
        ptrs = [call.bufs[n].data_ptr() for n in FIRST14]
        extra = [None] * (nbuf - 14)
        if nbuf > 14 and "nots" not in flags:
            extra[-1] = call.bufs["tpart"].data_ptr()
        cb = Buf(*(ptrs + extra + [call.bufs["tpart"].numel() * 4 if extra and extra[-1] else 0]))
        if "o8" in flags:
            raw = torch.empty(call.out.numel() + 4, dtype=call.out.dtype, device=dev)
            out = raw[4:].view(call.out.shape)
        else:
            out = torch.empty_like(call.out)
        o4 = RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
        tune = [(f.split("=")[0].encode(), int(f.split("=")[1])) for f in flags if "=" in f]   # e.g. k5_blk=2
        if tune:
            L.rsa_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        libs.append(dict(name=name, L=L, cb=cb, out=out, o4=o4, ts=[], td=[], tune=tune))

    def apply_tuning(lib):
        for key, val in lib["tune"]:
            assert lib["L"].rsa_set_tuning(key, val) == 0, (lib["name"], key)

    def run(lib):
        apply_tuning(lib)
        if fp8:
            fn = lib["L"].rsa_block_sparse_fwd_fp8
            fn.restype = ctypes.c_int
            rc = fn(ctypes.byref(call.lay), ctypes.byref(call.cf), ctypes.byref(lib["cb"]), lib["o4"], st)
            assert rc == 0, (lib["name"], rc)
            return
        rc = lib["L"].rsa_block_sparse_fwd(ctypes.byref(call.lay), *call.t, ctypes.byref(lib["cb"]), lib["o4"], st)
        assert rc == 0, (lib["name"], rc)

    Sd = 16384
    qd, kd, vd = (torch.randn(1, H, Sd, 128, device=dev).to(torch.bfloat16) for _ in range(3))
    od = torch.empty((1, Sd, H, 128), dtype=torch.bfloat16, device=dev)
    od4 = RsaOut4(od.data_ptr(), od.stride(0), od.stride(2), od.stride(1))
    fld = 4.0 * Sd * Sd * 128 * H

    def run_dense(lib):
        apply_tuning(lib)
        if fp8:
            return
        rc = lib["L"].rsa_dense_fwd(1, H, Sd, Sd, 128, 0, _core._t4(qd), _core._t4(kd), _core._t4(vd), Sd, Sd, od4, st)
        assert rc == 0, (lib["name"], rc)

    if pmc:
        for lib in libs:
            for _ in range(2):
                run(lib)
            for _ in range(2):
                run_dense(lib)
        torch.cuda.synchronize()
        return

    def timed(fn, n):
        evs = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return sorted(x.elapsed_time(y) for x, y in evs)[n // 2]

    for lib in libs:  # warm-up + output check
        run(lib); run(lib); run_dense(lib)
    torch.cuda.synchronize()
    ref = libs[0]["out"].float()
    for lib in libs[1:]:
        d = (lib["out"].float() - ref).abs()
        print(f"{lib['name']}: max|out - {libs[0]['name']}| = {d.max().item():.3e}", flush=True)
    for r in range(rounds):
        order = libs if r % 2 == 0 else libs[::-1]
        for lib in order:
            lib["ts"].append(timed(lambda: run(lib), 3))
        for lib in order:
            lib["td"].append(timed(lambda: run_dense(lib), 3))
        print(f"round {r:2d}: " + " | ".join(f"{lib['name']} {lib['ts'][-1]:7.3f} (dense16k {lib['td'][-1]:6.3f})"
                                             for lib in libs), flush=True)
    for lib in libs:
        ts, td = sorted(lib["ts"]), sorted(lib["td"])
        print(f"{lib['name']:>10s}: sparse median {ts[len(ts)//2]:7.3f} ms min {ts[0]:7.3f} max {ts[-1]:7.3f} "
              f"({flops/ts[len(ts)//2]/1e9:6.1f} TFLOP/s) | dense16k median {td[len(td)//2]:6.3f} min {td[0]:6.3f} "
              f"({fld/td[len(td)//2]/1e9:6.1f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
