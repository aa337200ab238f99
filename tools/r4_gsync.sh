#!/bin/bash
# round 4: aligned starts of the 64-row K5 (tuning key k5_gsync): interleaved A/B in one process, then L2 hit / miss counters
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4g}
L=rectified_spaattn_amd/librsa_hip.so
( timeout 600 python tools/ab_libs.py free=$L::k5_gsync=0 aligned=$L::k5_gsync=1 --rounds ${ROUNDS:-8} ) > gpurun_out/${T}_ab.txt 2>&1
echo "rc=$?" >> gpurun_out/${T}_ab.txt
tail -5 gpurun_out/${T}_ab.txt | cut -c1-200
R=$PWD
cd /tmp; export TMPDIR=/tmp
for G in 0 1; do
  export RSA_K5_GSYNC=$G
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${T}_pmc$G -- python3 $R/tools/ab_libs.py x=$R/$L::k5_gsync=$G --pmc > $R/gpurun_out/${T}_pmc$G.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, sys, os
T = os.environ.get("T", "r4g")
for G in (0, 1):
    for f in glob.glob(f"gpurun_out/{sys.argv[1] if len(sys.argv) > 1 else T}_pmc{G}/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "bsfwd64" in r["Kernel_Name"]:
                key = (r["Dispatch_Id"], r["Counter_Name"])
                acc[key] = acc.get(key, 0) + float(r["Counter_Value"])
        ids = sorted({k[0] for k in acc}, key=int)
        for d in ids:
            h, m = acc.get((d, "TCC_HIT_sum"), 0), acc.get((d, "TCC_MISS_sum"), 0)
            print(f"gsync={G} dispatch {d}: L2 hits {h/1e6:.1f} M misses {m/1e6:.1f} M hit rate {h/(h+m+1e-9):.3f} fabric reads ~{m*128/1e9:.1f} GB")
PY
