#!/bin/bash
# round 4: aligned starts of the 64-row K5, the record: the new tests, interleaved A/B per regime (free-running / aligned /
# the 32-row kernel), L2 hit / miss counters of both settings, in-kernel stamps of both, the default bench line
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4k}
R=$PWD
L=rectified_spaattn_amd/librsa_hip.so
( timeout 600 python -m pytest tests/test_gpu_gsync.py -q -m gpu 2>&1 | tail -3 ) > gpurun_out/${T}_tests.txt
for RG in r2 r1 locality; do
  ( RSA_PERF_REGIME=$RG timeout 600 python tools/ab_libs.py free=$L::k5_w64=1,k5_gsync=0 aligned=$L::k5_w64=1,k5_gsync=1 row32=$L::k5_w64=0,k5_gsync=0 --rounds 8 ) > gpurun_out/${T}_ab_$RG.txt 2>&1
done
( timeout 300 python tools/diag_k5w.py ) > gpurun_out/${T}_diag.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cd /tmp; export TMPDIR=/tmp
for G in 0 1; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${T}_pmc$G -- python3 $R/tools/ab_libs.py x=$R/$L::k5_w64=1,k5_gsync=$G --pmc > $R/gpurun_out/${T}_pmc$G.log 2>&1
done
cd $R
python3 - > gpurun_out/${T}_l2.txt <<PY
import csv, glob
for G in (0, 1):
    for f in glob.glob(f"gpurun_out/${T}_pmc{G}/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "bsfwd64" in r["Kernel_Name"]:
                key = (r["Dispatch_Id"], r["Counter_Name"])
                acc[key] = acc.get(key, 0) + float(r["Counter_Value"])
        for d in sorted({k[0] for k in acc}, key=int):
            h, m = acc.get((d, "TCC_HIT_sum"), 0), acc.get((d, "TCC_MISS_sum"), 0)
            kind = "sparse R2" if h + m > 5e8 else "dense 16k"
            print(f"k5_gsync={G} {kind} launch: L2 hits {h/1e6:.1f} M, misses {m/1e6:.1f} M, hit rate {h/(h+m+1e-9):.3f}, fabric reads ~{m*128/1e9:.1f} GB")
PY
cat gpurun_out/${T}_tests.txt
for RG in r2 r1 locality; do echo $RG; tail -3 gpurun_out/${T}_ab_$RG.txt | cut -c1-200; done
cat gpurun_out/${T}_diag.txt | tail -2; cat gpurun_out/${T}_l2.txt; tail -c 2500 gpurun_out/${T}_bench.json; tail -2 gpurun_out/${T}_bench.err
