#!/bin/bash
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py tests/test_gpu_framework.py -q -p no:cacheprovider -x -k "hat or HAT" 2>&1 | tail -4
timeout 120 python tools/method_steps.py hat 64 10 2>&1 | tail -1
timeout 120 python tools/method_steps.py packnet 64 10 2>&1 | tail -1
timeout 200 python tools/two_stream_probe.py 2>&1 | tail -5
timeout 200 python tools/two_stream_probe.py base_VGG9_cl_512_512 2>&1 | tail -5
