#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== tests"; timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
echo "== bench"; ( time timeout 900 python bench.py 2> gpurun_out/r03c_bench.err > gpurun_out/r03c_bench.json ) 2>&1 | tail -4
tail -3 gpurun_out/r03c_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03c_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
c = d.get("configs") or {}
for k, v in c.items():
    if isinstance(v, dict) and "ms_per_step" in v: print(k, round(v["ms_per_step"], 3), round(v["frac_of_f32_mfma_peak"], 3))
for r in c.get("hbm_kernels", []): print(r["kernel"], round(r["us"], 1), "us", round(r["achieved_TBps"], 2), "TB/s", round(r["frac_of_hbm_peak"], 3))
for k, v in (c.get("conv_backward") or {}).items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
s = d.get("sweep") or {}
print({k: v for k, v in s.items() if k not in ("what", "pair")})
print(s.get("pair"))
PY
echo "== N=2 dry run (gloo, both ranks on cuda:0)"
( time CLHIP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 2> gpurun_out/r03c_n2.err > gpurun_out/r03c_n2.json ) 2>&1 | tail -3
tail -3 gpurun_out/r03c_n2.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03c_n2.json").read().strip().splitlines()[-1])
    print(d["value"], d["n_gpus"], json.dumps(d.get("grid"))[:1500])
except Exception as e:
    print("n2 parse failed", e)
PY
