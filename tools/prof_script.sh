# usage: bash tools/prof_script.sh <tag> <script.py> : rocprofv3 kernel stats of a python script, our kernels only
TAG=$1; R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/$2 > $R/gpurun_out/$TAG.log 2>&1
cd $R; python3 tools/summarize_prof.py $(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1) | cut -c1-130
