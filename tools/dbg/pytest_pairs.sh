#!/bin/bash
# debug: which earlier test makes the unsampled capture in test_one_capture_... crash?
cd "$(dirname "$0")/../.."
for k in "one_capture and hybrid" "unsampled_partitions or (one_capture and hybrid)" "(sampled_replay and hybrid) or (one_capture and hybrid)" "(sampled_replay and straight) or (one_capture and hybrid)" "(sampled_replay and two_pass) or (one_capture and hybrid)" "one_capture"; do
  echo "=== -k $k"
  timeout -k 10 200 python -X faulthandler -m pytest tests/test_gpu_stepgraph.py -m gpu -x -q -k "$k" 2>&1 | tail -4
done
