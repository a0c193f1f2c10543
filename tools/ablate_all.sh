#!/bin/bash
# steer-kernel ablations on the GPU box: which part of a rollout step costs what
cd /root/repo
for v in "" ABL_NOFEAS ABL_NORUDDER ABL_NOTRIG "ABL_NOTRIG,ABL_NOFEAS"; do
  timeout 200 python tools/ablate_steer.py "$v" 2>&1 | grep "steer avg"
done
