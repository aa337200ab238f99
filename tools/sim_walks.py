#!/usr/bin/env python3
"""Monte-Carlo of what an XCD's L2 sees under K5's sparse walks (CPU only, no GPU needed): 64 workgroups walk ascending lists of
K kept key blocks out of NB (independent random lists = regime R2), one block per time step (+- jitter), the L2 = an LRU of C
key blocks (4 MiB / 64 KiB per K + V block = 64 nominal; ~48 effective reproduces the measured hit rates).

Scheduling policies compared:
  free      the workgroups start at unrelated times (round 3)
  aligned   they start together and run freely (the product since round 4: rsa_attn.h, aligned starts)
  sorted    aligned + the 64 walks of a generation sorted by their first kept block
  window W  aligned + a walk may not begin a key block more than W block positions ahead of the slowest walk of its XCD
            (a position-synchronised sweep: what "walks that stay together" would have to mean for independent lists)

Prints hit rate, time per step relative to an unconstrained walk (1.0 = no stall) and the stall share.  The result the round-5
write-up quotes (profiles/r05_k5_walk_sim.md): every window that raises the hit rate noticeably costs more in stalls than the
aligned-start change gained per hit-rate point, because the work per key range is uneven across independent walks (a window of
w positions holds w K / NB +- sqrt(..) blocks of a walk)."""
import heapq
import sys
from collections import OrderedDict

import numpy as np

NB, K, NW = 900, 90, 64


def run(rng, C, W, jitter=0.03, gens=6, aligned=True, sort_first=False, nb=NB, k=K):
    hits = acc = 0
    cache = OrderedDict()
    tot_time, stall = 0.0, 0.0
    for _ in range(gens):
        lists = [np.sort(rng.choice(nb, k, replace=False)) for _ in range(NW)]
        if sort_first:
            lists.sort(key=lambda l: l[0])
        idx = [0] * NW
        t0 = tot_time
        ev = [((t0 if aligned else t0 + rng.uniform(0, k)), i) for i in range(NW)]
        heapq.heapify(ev)
        pos = [lists[i][0] for i in range(NW)]
        waiting = []
        tend = t0

        def release(t):
            nonlocal waiting, stall
            mn = min(pos)
            keep = []
            for tw, j in waiting:
                if lists[j][idx[j]] <= mn + W:
                    stall += t - tw
                    heapq.heappush(ev, (t, j))
                else:
                    keep.append((tw, j))
            waiting = keep

        while ev or waiting:
            t, i = heapq.heappop(ev)
            if idx[i] >= k:
                pos[i] = 10 ** 9
                tend = max(tend, t)
                if W is not None:
                    release(t)
                continue
            b = lists[i][idx[i]]
            pos[i] = b
            if W is not None and b > min(pos) + W:
                waiting.append((t, i))
                release(t)
                continue
            acc += 1
            if b in cache:
                hits += 1
                cache.move_to_end(b)
            else:
                cache[b] = 1
                if len(cache) > C:
                    cache.popitem(last=False)
            idx[i] += 1
            heapq.heappush(ev, (t + max(1.0 + rng.normal(0, jitter), 0.5), i))
            if W is not None and waiting:
                release(t)
        tot_time = tend
    return hits / acc, tot_time / (gens * k), stall / (gens * k * NW)


def main():
    rng = np.random.default_rng(0)
    dens = [(900, 90, "R2: 10 % of 900 key blocks"), (900, 180, "script: 20 %"), (591, 147, "Wan2.1 script: 25 % of 591")]
    for nb, k, name in dens:
        print(f"== {name}")
        for C in (48, 64):
            f = run(rng, C, None, aligned=False, nb=nb, k=k)
            a = run(rng, C, None, nb=nb, k=k)
            s = run(rng, C, None, sort_first=True, nb=nb, k=k)
            print(f"  L2 = {C} blocks: free {f[0]:.3f} | aligned {a[0]:.3f} | aligned + sorted by first block {s[0]:.3f}")
            for W in (32, 64, 96, 128, 192):
                h, tt, st = run(rng, C, W, nb=nb, k=k)
                print(f"     window {W:4d}: hits {h:.3f}  time per step {tt:.3f}  stalled {st:.3f}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
