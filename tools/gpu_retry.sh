#!/bin/bash
# usage: tools/gpu_retry.sh <timeout-seconds> <log> <command...> : gpurun with retries while no GPU slot is free (exit 3)
T=$1; LOG=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $LOG; then echo "exit=$rc" >> $LOG; exit $rc; fi
  sleep 45
done
echo "exit=gave-up" >> $LOG
