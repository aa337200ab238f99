#!/bin/bash
# round 6: what makes the default line slower after the GPU suite ran in the same lease (tools/r6_x.sh: 14.60 / 14.60 -> 14.81 / 15.29)?
# the same line six times with the device's temperature / power / clocks / memory use sampled every 3 s:
#   a b (fresh) | 120 s idle | c | GPU test suite | d e | 240 s idle | f ;  leftover processes listed after the suite
OUT=gpurun_out
smi_loop() { while true; do echo "t=$(date +%s)"; rocm-smi --showtemp --showpower --showclocks --showmemuse 2>/dev/null | grep -E "Temperature|Power|sclk|mclk|fclk|socclk|Memory|VRAM" ; sleep 3; done; }
smi_loop > $OUT/r6y_smi.txt 2>&1 &
SMI=$!
run() { echo "bench $1 start $(date +%s)" >> $OUT/r6y_phases.txt; python bench.py --no-cpu-baseline --no-extras > $OUT/r6y_bench_$1.json 2> $OUT/r6y_bench_$1.err; echo "bench $1 end $(date +%s)" >> $OUT/r6y_phases.txt; }
rm -f $OUT/r6y_phases.txt
run a; run b
echo "idle 120 start $(date +%s)" >> $OUT/r6y_phases.txt; sleep 120
run c
echo "suite start $(date +%s)" >> $OUT/r6y_phases.txt
python -m pytest tests -x -q -m gpu > $OUT/r6y_suite.txt 2>&1
echo "suite end $(date +%s)" >> $OUT/r6y_phases.txt
rocm-smi --showpids > $OUT/r6y_pids_after_suite.txt 2>&1
ps -eo pid,ppid,etime,stat,cmd | grep -v "ps -eo" | grep -E "python|rsa|pytest" > $OUT/r6y_ps_after_suite.txt 2>&1
run d; run e
echo "idle 240 start $(date +%s)" >> $OUT/r6y_phases.txt; sleep 240
run f
kill $SMI
