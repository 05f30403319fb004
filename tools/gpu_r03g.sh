#!/bin/bash
export TMPDIR=/tmp
echo base; timeout 100 python tools/wino_bench.py 10 2>&1 | grep "64x64  @32\|256x256\|512x512"
for v in notf nolds nostage mfma; do echo $v; CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so timeout 100 python tools/wino_bench.py 10 2>&1 | grep "64x64  @32\|256x256\|512x512" | sed 's/fwd+pool.*//' ; done
timeout 200 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "full_size and small" 2>&1 | grep -v "^$" | tail -25
