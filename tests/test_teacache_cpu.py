"""TeaCache controller on CPU tensors: its decisions, accumulator and residual handling must equal a
step-by-step re-enactment of the reference forwards' bookkeeping (scripts/main_hunyuan.py:110-157 single stream;
main_wan21t2v.py:101-164 even/odd streams) on the same input sequence."""
import numpy as np
import torch

from rectified_spaattn_amd.teacache import COEFFICIENTS, TeaCache, rel_l1_distance


def _inputs(n, seed, drift):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 64, 32, generator=g)
    seq = []
    for i in range(n):
        x = x + drift * (1.0 + 0.5 * np.sin(i)) * torch.randn(1, 64, 32, generator=g)
        seq.append(x.clone())
    return seq


def _enact_single(seq, num_steps, thresh, coeffs):
    """The bookkeeping of the reference's single-stream forward, written out."""
    cnt, acc, prev, out = 0, 0.0, None, []
    for x in seq:
        if cnt == 0 or cnt == num_steps - 1:
            calc, acc = True, 0.0
        else:
            acc += np.poly1d(coeffs)(((x - prev).abs().mean() / prev.abs().mean()).item())
            if acc < thresh:
                calc = False
            else:
                calc, acc = True, 0.0
        prev = x
        cnt += 1
        if cnt == num_steps:
            cnt = 0
        out.append(calc)
    return out


def _enact_two_streams(seq, total, thresh, coeffs, ret, cutoff):
    cnt, acc, prev, out = 0, [0.0, 0.0], [None, None], []
    for x in seq:
        s = cnt % 2
        if cnt < ret or cnt >= cutoff:
            calc, acc[s] = True, 0.0
        else:
            acc[s] += np.poly1d(coeffs)(((x - prev[s]).abs().mean() / prev[s].abs().mean()).item())
            if acc[s] < thresh:
                calc = False
            else:
                calc, acc[s] = True, 0.0
        prev[s] = x.clone()
        cnt += 1
        if cnt == total:
            cnt = 0
        out.append(calc)
    return out


def test_single_stream_matches_reference_bookkeeping():
    for seed, drift, thresh in ((1, 0.02, 0.15), (2, 0.05, 0.1), (3, 0.2, 0.3)):
        seq = _inputs(2 * 20 + 3, seed, drift)           # runs over two generations: the counter wraps
        tc = TeaCache.hunyuan(20, thresh)
        got = []
        for x in seq:
            c = tc.should_compute(x)
            got.append(c)
            if c:
                tc.store_residual(x * 2, x)
        want = _enact_single(seq, 20, thresh, COEFFICIENTS["hunyuan"])
        assert got == want
        assert got[0] and got[19] and got[20] and got[39] and not all(got)


def test_two_streams_match_reference_bookkeeping():
    for use_ret in (True, False):
        steps = 12
        seq = _inputs(2 * steps + 5, 7, 0.004 if use_ret else 0.05)
        tc = TeaCache.wan(steps, 0.2, use_ret_steps=use_ret)
        got = []
        for x in seq:
            c = tc.should_compute(x)
            got.append(c)
            if c:
                tc.store_residual(x + 1, x)
        key = "wan21_14b_ret" if use_ret else "wan21_14b"
        ret, cutoff = (10, 2 * steps) if use_ret else (2, 2 * steps - 2)
        assert got == _enact_two_streams(seq, 2 * steps, 0.2, COEFFICIENTS[key], ret, cutoff)
        assert all(got[:ret])


def test_residual_is_kept_per_stream_and_applied_in_place():
    tc = TeaCache(8, 1e9, [1.0, 0.0], streams=2, ret_calls=2, cutoff_calls=8)   # huge threshold: skip whenever allowed
    h = [torch.full((2, 3), float(i + 1)) for i in range(6)]
    assert tc.should_compute(h[0]) and tc.stream == 0
    tc.store_residual(h[0] + 10, h[0])
    assert tc.should_compute(h[1]) and tc.stream == 1
    tc.store_residual(h[1] + 20, h[1])
    assert not tc.should_compute(h[2]) and tc.stream == 0
    x = torch.zeros(2, 3)
    assert tc.apply_residual(x) is x and torch.equal(x, torch.full((2, 3), 10.0))
    assert not tc.should_compute(h[3]) and tc.stream == 1
    assert torch.equal(tc.apply_residual(torch.zeros(2, 3)), torch.full((2, 3), 20.0))


def test_rel_l1_distance_cpu_expression():
    a, b = torch.tensor([1.0, -2.0, 3.0]), torch.tensor([2.0, -2.0, 1.0])
    assert abs(rel_l1_distance(a, b) - (3.0 / 3) / (5.0 / 3)) < 1e-7
