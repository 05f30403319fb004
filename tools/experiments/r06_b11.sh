#!/bin/bash
# round 6, second session, call 11: which activation is first non-finite in the sweep's failing step (main build)
timeout 900 python tools/experiments/nan_diag.py 4 2>&1 | grep -v amdgpu.ids | grep -E "LOSS_STEP|input of plan|sample|grad of param|sweep|first non" | cut -c1-260 | head -40
