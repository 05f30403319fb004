#!/bin/bash
# round 6, session M: does the bf16-split kernel lose time to its partial last round of blocks? 64 -> 64 @32x32 at N = 192 (1536 blocks =
# 2 whole rounds of 3 blocks x 256 CUs), 200 (1600), 96, 100, 288, 300
set -u
mkdir -p gpurun_out
for n in 96 100 192 200 288 300 384 400; do timeout 120 python tools/bs_layer.py $n 64 64 32 2>&1 | tail -1; done | tee gpurun_out/r06_m_bs_rounds.txt
