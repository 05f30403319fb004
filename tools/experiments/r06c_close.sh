#!/bin/bash
# round-6 closing run of the THIRD session on the final build: GPU suite, smoke, the driver's bench command, 2-rank dry run of the N > 1 bench path (gloo, both
# ranks on cuda:0, HIP_VISIBLE_DEVICES given); then the profile set of the same build: rocprofv3 kernel trace of the plan-executor step,
# MFMA-pipe busy (PMC) per kernel for the three VGG9 widths, per-layer conv timings, HBM traffic of the layer-2 launches, AlexNet step
set -u
mkdir -p gpurun_out/r06w; export TMPDIR=/tmp
O=gpurun_out/r06w; P=$PWD
SECONDS=0
python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -4 | cut -c1-300
echo "suite: $SECONDS s"
python __graft_entry__.py --smoke 2>&1 | tail -1
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err > $O/bench.json; echo "bench: $SECONDS s, $(wc -c < $O/bench.json) bytes, $(wc -l < $O/bench.json) line(s)"
cp gpurun_out/bench_details.json $O/bench_details.json
tail -1 $O/bench.json | cut -c1-3200
SECONDS=0
HIP_VISIBLE_DEVICES=0 CLHIP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 20 --warmup 5 2> $O/bench2.err > $O/bench2.json; echo "bench --gpus 2 (gloo dry run): $SECONDS s"
tail -1 $O/bench2.json | cut -c1-1800
SECONDS=0
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$O/prof -- python $P/tools/one_step.py 20 small_VGG9_cl_128_128 > $P/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_by_grid.py "$f" > $O/kernel_stats_by_grid.csv
rm -rf $O/prof
head -14 $O/kernel_stats_by_grid.csv | cut -c1-200
echo "trace: $SECONDS s"; SECONDS=0
for m in small base wide; do
  timeout 120 python tools/conv_bench.py $m 200 20 2>&1 | grep -v amdgpu.ids > $O/conv_layers_$m.txt
  bash tools/gpu_mfma_util.sh ${m}_VGG9_cl_$([ $m = small ] && echo 128_128 || echo 512_512) r06w/mfma_util_$m > /dev/null 2>&1
done
tail -4 $O/conv_layers_small.txt; tail -1 $O/conv_layers_base.txt; tail -1 $O/conv_layers_wide.txt
head -10 $O/mfma_util_small.csv | cut -c1-150
for k in bs_fwdpool bs_dgrad_unpool bs_wgrad_unpool; do bash tools/gpu_traffic.sh r06w/t_$k $k 200 64 64 32 3 2>&1 | grep -v amdgpu.ids | tee -a $O/traffic.txt; done
timeout 200 python tools/alexnet_step.py 128 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/alexnet_step.txt
echo "profiles: $SECONDS s"
SECONDS=0
CLHIP_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29566 bench.py --gpus 1 --steps 20 --warmup 5 --no-sweep 2> $O/bench_rccl1.err > $O/bench_rccl1.json; echo "bench, one-rank RCCL communicator (dry run of the N > 1 path): $SECONDS s"
tail -1 $O/bench_rccl1.json | cut -c1-900
