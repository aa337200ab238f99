# usage: bash tools/prof_quick.sh <tag> <bench args...> : rocprofv3 kernel stats of a short bench run, summary to stdout
TAG=$1; shift
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG.log 2>&1
cd $R; python3 tools/summarize_prof.py $(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1) | cut -c1-140
