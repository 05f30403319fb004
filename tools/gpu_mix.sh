#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv or pool or full_size or net or determin" 2>&1 | tail -3
python tools/l1_dump.py /tmp/new.pt; CLHIP_LIB=clsurvey_amd/libclhip_nomix.so python tools/l1_dump.py /tmp/old.pt
for v in "" _nomix; do echo "== $v"; CLHIP_LIB=clsurvey_amd/libclhip$v.so timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep "64x64  @32\|ALL"; done
timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-configs --no-sweep 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
