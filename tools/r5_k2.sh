#!/bin/bash
export RSA_TUNING=1
python tools/perf_select.py k2_split=1 k2_form=0,1,2,3,4,6,7 2>&1 | grep -v amdgpu.ids | sed 's/| K3.*//' > gpurun_out/r5b_forms_split.txt
cat gpurun_out/r5b_forms_split.txt
