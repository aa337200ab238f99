set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.txt 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/bench4.json 2> gpurun_out/bench4.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof4.log 2>&1
cd $R; ls gpurun_out/prof4 | head; find gpurun_out/prof4 -name "*kernel_stats*" | head
