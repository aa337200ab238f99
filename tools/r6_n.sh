#!/bin/bash
# round 6: PMC passes (counters only) over the CogVideoX K5 launch (head dim 64, the 64-row kernel) and its 32-row partner
mkdir -p gpurun_out
export RSA_TUNING=1 RSA_PERF_WORKLOAD=cogvideox_768p_81f RSA_PERF_NODENSE=1
bash tools/pmc_passes.sh r6n_pmc_d64 > /dev/null 2>&1; cp gpurun_out/r6n_pmc_d64/summary.txt gpurun_out/r6n_pmc_summary_d64.txt; rm -rf gpurun_out/r6n_pmc_d64
RSA_K5_W64=1 bash tools/pmc_passes.sh r6n_pmc_d64_32 > /dev/null 2>&1; cp gpurun_out/r6n_pmc_d64_32/summary.txt gpurun_out/r6n_pmc_summary_d64_32row.txt; rm -rf gpurun_out/r6n_pmc_d64_32
grep -E "bsfwd" gpurun_out/r6n_pmc_summary_d64.txt | grep -E "MFMA|GRBM|INSTS_VALU|WAVE_CYCLES" | cut -c1-150
grep -E "bsfwd" gpurun_out/r6n_pmc_summary_d64_32row.txt | grep -E "MFMA|GRBM|INSTS_VALU|WAVE_CYCLES" | cut -c1-150
