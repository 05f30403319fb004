#!/bin/bash
# round 6, third session, call 5: tools/micro/store_policy.hip — cost of a dependent kernel boundary behind X MB of fresh stores, by store cache policy
set -u
mkdir -p gpurun_out
timeout 300 tools/micro/store_policy 50 2>&1 | tee gpurun_out/r06c5_store_policy.txt
