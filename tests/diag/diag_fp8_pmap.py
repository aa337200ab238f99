"""Not a test (no test_ prefix): numpy study of the e4m3 kernel's P map.

P must reach e4m3 to feed the fp8 MFMA.  Exact form: P = 2^x by v_exp_f32, then round-to-nearest e4m3 (relative step 2^-3,
so +-2^-4 rounding).  Code map (the product, rsa_attn_fp8_kernel.hip PMap<true>): the byte is rint(8 x + 56), i.e. the
format's own piecewise-linear log2 -- P = 2^e (1 + m3/8) where e + m3/8 ~ x; against 2^x that is a smooth factor in
[1, 1.0615] (constant factors cancel between numerator and row sum) on top of the same +-1/16 mantissa rounding.  This script
measures the attention-output error of both against exact P on random scores: the code map costs 1.18-1.26x the exact form's RMS
error; run on CPU: python tests/diag/diag_fp8_pmap.py"""
import numpy as np

rng = np.random.default_rng(0)


def e4m3_val(c):
    c = np.asarray(c)
    e, m = (c >> 3) & 15, c & 7
    return np.where(e == 0, m * 2.0 ** -9, (1 + m / 8.0) * 2.0 ** (e.astype(float) - 7))


VALS = e4m3_val(np.arange(127))


def rne_e4m3(p):
    idx = np.clip(np.searchsorted(VALS, p), 1, 126)
    lo, hi = VALS[idx - 1], VALS[idx]
    return VALS[np.where(p - lo <= hi - p, idx - 1, idx)]


def code_map(x):
    return VALS[np.clip(np.rint(8 * x + 56), 0, 126).astype(int)]


def run(scores_std, nkeys, nrow=256, D=64):
    S = rng.normal(0, scores_std, (nrow, nkeys))
    V = rng.normal(0, 1, (nkeys, D))
    m = S.max(1, keepdims=True)
    P = np.exp2(S - m)
    O = (P @ V) / P.sum(1, keepdims=True)
    res = {}
    for name, off, thr in (("exponential", 4.0, 4.0), ("code map", 6.5, 2.0)):
        u = rng.uniform(0, thr, (nrow, 1))            # the deferred reference lags the true maximum by up to the threshold
        x = S - m + off + u
        Pq = rne_e4m3(np.exp2(x)) if name == "exponential" else code_map(x)
        d = np.abs((Pq @ V) / Pq.sum(1, keepdims=True) - O)
        res[name] = (d.max(), d.mean(), np.sqrt((d ** 2).mean() / (O ** 2).mean()))
    return res


if __name__ == "__main__":
    print("score std, keys | form: max |dO|, mean |dO|, RMS relative")
    for std, n in ((0.5, 11776), (1.5, 11776), (3.0, 11776), (6.0, 11776), (1.5, 1024), (3.0, 128)):
        r = run(std, n)
        print(f"{std:4.1f} {n:6d} | " + " | ".join(f"{k}: {a:.2e} {b:.2e} {c:.4f}" for k, (a, b, c) in r.items())
              + f" | ratio {r['code map'][2] / r['exponential'][2]:.2f}")
