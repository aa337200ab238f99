#!/bin/bash
# round 6: K4 with two chunks in flight, text_combine with its loads up front: whole suite, per-kernel times, kernel stats of a short bench run
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4 ) > gpurun_out/r6m_suite.txt 2>&1; cat gpurun_out/r6m_suite.txt
( python tools/perf_select.py k4_split=1,0; RSA_PERF_WORKLOAD=wan21_720p_81f python tools/perf_select.py k4_split=1,0 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6m_select.txt; cat gpurun_out/r6m_select.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6m_prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r6m_prof.log 2>&1
cd $R
F=$(find gpurun_out/r6m_prof -name "*kernel_stats.csv" | head -1); cp $F gpurun_out/r6m_prof_kernel_stats.csv; python3 tools/summarize_prof.py $F > gpurun_out/r6m_prof_kernel_stats.md; find gpurun_out/r6m_prof -name "*kernel_trace.csv" -delete
cat gpurun_out/r6m_prof_kernel_stats.md | tail -9; grep "compensation" gpurun_out/r6m_prof_kernel_stats.csv | cut -c1-140; tail -c 300 gpurun_out/r6m_prof.log
