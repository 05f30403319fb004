"""round 6, third session: how much of the 10-task sweep is torch.save / torch.load of model files (the sweep alone: bench.full_sweep without CPU legs)?"""
import sys, time, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--sweep-only", "--no-cpu-baseline"]
import torch
from clsurvey_amd.methods import train_common as TC
acc = {"n": 0, "s": 0.0}
orig = TC.save_model
def timed(model, path):
    torch.cuda.synchronize()
    t = time.perf_counter()
    orig(model, path)
    acc["n"] += 1; acc["s"] += time.perf_counter() - t
TC.save_model = timed
orig_save = torch.save
acc2 = {"n": 0, "s": 0.0}
def tsave(*a, **k):
    t = time.perf_counter(); r = orig_save(*a, **k); acc2["n"] += 1; acc2["s"] += time.perf_counter() - t; return r
torch.save = tsave
orig_load = torch.load
acc3 = {"n": 0, "s": 0.0}
def tload(*a, **k):
    t = time.perf_counter(); r = orig_load(*a, **k); acc3["n"] += 1; acc3["s"] += time.perf_counter() - t; return r
torch.load = tload
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
res = b.full_sweep(0, cpu_threads=0, forced=False, chain=False)
print("sweep gpu_s %.1f first %.1f | save_model calls %d %.2f s | torch.save calls %d %.2f s | torch.load calls %d %.2f s" % (res["gpu_s"], res["gpu_first_task_s"], acc["n"], acc["s"], acc2["n"], acc2["s"], acc3["n"], acc3["s"]))
