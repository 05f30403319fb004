#!/bin/bash
# tuning session: tests (quick), conv micro-bench per lib variant, PMC counters for the conv kernels
set -u
mkdir -p gpurun_out
TAG=${1:-t1}
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "Loss:\|^Epoch\|lr is\|Accuracy\|Training complete\|ATTEMPT\|FINETUNE\|USING SI\|^$" | tail -40 | tee gpurun_out/${TAG}_tests.log
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.log
echo "== bench"; timeout 600 python bench.py --steps 100 --warmup 10 2> gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json | cut -c1-400
for v in ""; do
  if [ -z "$v" ]; then unset CLHIP_LIB; else export CLHIP_LIB=$PWD/clsurvey_amd/libclhip_$v.so; fi
  echo "== conv_bench ${v:-default}"; timeout 300 python tools/conv_bench.py small 200 20 2>&1 | tail -12 | tee gpurun_out/${TAG}_convbench_${v:-default}.log
done
unset CLHIP_LIB
echo "== counters available"
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|GRBM|TCC|TCP|TA)_[A-Z0-9_]+" | sort -u > gpurun_out/${TAG}_counters.txt; wc -l gpurun_out/${TAG}_counters.txt
grep -E "MFMA|LDS|WAIT|BUSY_CY|WAVE_CYCLES|ACTIVE_INST|GUI_ACTIVE" gpurun_out/${TAG}_counters.txt | tr '\n' ' '
echo
echo "== pmc"
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/${TAG}_pmc -- python $OLDPWD/tools/conv_bench.py small 200 2 > $OLDPWD/gpurun_out/${TAG}_pmc.log 2>&1 )
tail -3 gpurun_out/${TAG}_pmc.log
find gpurun_out/${TAG}_pmc -name "*.csv" | head
