#!/usr/bin/env python
"""Wall-clock of a full 10-task EWC sweep at Tiny-ImageNet shapes (BASELINE.json's second metric) on one MI355X.

Runs what `framework/main.py` runs for `small_VGG9_cl_128_128 --method_name EWC --test` with the reference's defaults
(5-value LR grid, 70 epochs with the count-based LR drop / early stop, batch 200, two-phase hyper-parameter framework,
final evaluation of every model on every task): SI first-task model dump, then the EWC task sequence.  Data: synthetic
tensors of Tiny-ImageNet's shape (10 tasks x 20 classes, 8000 / 2000 / 1000 images of 3x64x64; there is no dataset in
the container).  Prints ONE JSON line: seconds per stage, image passes by phase, and the CPU-equivalent time of the same
image passes at the host-oracle rate given with --cpu-train-rate (images / s of bench.py's cpu_baseline; eval passes are
priced at 3x that rate = forward only).

usage: sweep.py [--tasks 10] [--epochs 70] [--root /tmp/sweep] [--cpu-train-rate 527]
"""
import argparse
import json
import os
import shutil
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import data as D  # noqa: E402
from clsurvey_amd.framework import driver  # noqa: E402
from clsurvey_amd.framework.tasks import SyntheticTaskSequence  # noqa: E402
from clsurvey_amd.methods import method as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tasks", type=int, default=10)
ap.add_argument("--epochs", type=int, default=70)
ap.add_argument("--root", default="/tmp/clhip_sweep")
ap.add_argument("--method", default="EWC")
ap.add_argument("--model", default="small_VGG9_cl_128_128")
ap.add_argument("--cpu-train-rate", type=float, default=0.0)
ap.add_argument("--hw", type=int, default=64, help="image side (224 = the iNaturalist geometry of BASELINE configs[4], data/dataset.py:97)")
ap.add_argument("--sizes", type=str, default="8000,2000,1000", help="train,val,test images per task")
ap.add_argument("--batch", type=int, default=200)
ap.add_argument("--friendly-init", action="store_true", help="base-model file with a Kaiming-normal classifier (see bench._base_model_file)")
a = ap.parse_args()

SIZES = tuple(int(v) for v in a.sizes.split(","))
counts = {"train": 0, "eval": 0}
_iter = D.DeviceLoader.__iter__


def counting_iter(self):
    for x, y in _iter(self):
        counts["train" if self.n >= SIZES[0] else "eval"] += x.shape[0]
        yield x, y


D.DeviceLoader.__iter__ = counting_iter
shutil.rmtree(a.root, ignore_errors=True)
ds = SyntheticTaskSequence(os.path.join(a.root, "data"), task_count=a.tasks, classes_per_task=20, sizes=SIZES, hw=a.hw,
                           name="synthetic_tiny_imagenet")
if a.friendly_init:
    from clsurvey_amd import models as _models
    torch.manual_seed(0)
    _m = _models.parse_model_name(a.model, (a.hw, a.hw), 20)
    for _mod in _m.modules():
        if isinstance(_mod, torch.nn.Linear):
            torch.nn.init.kaiming_normal_(_mod.weight, nonlinearity="relu")
    os.makedirs(os.path.join(a.root, "models"), exist_ok=True)
    torch.save(_m, os.path.join(a.root, "models", a.model + ".pth.tar"))
t0 = time.time()
for i in range(1, a.tasks + 1):
    ds.get_task_dataset_path(str(i))
t_data = time.time() - t0
common = [a.model, "--num_epochs", str(a.epochs), "--results_root", a.root, "--batch_size", str(a.batch)]
devnull = open(os.devnull, "w")
stdout = sys.stdout
sys.stdout = devnull          # the trainers print per-epoch lines like the reference
try:
    t0 = time.time()
    driver.main(common + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"), dataset=ds)
    torch.cuda.synchronize()
    t_first = time.time() - t0
    c_first = dict(counts)
    t0 = time.time()
    out = driver.main(common + ["--method_name", a.method, "--test"], method=M.parse(a.method), dataset=ds)
    torch.cuda.synchronize()
    t_seq = time.time() - t0
finally:
    sys.stdout = stdout
res = out["results"]
last = max(res)
accs = res[last]["seq_res"]
final_acc = [float(accs[t][-1]) for t in sorted(accs)] if isinstance(accs, dict) else []
line = {"metric": "full-sweep wall-clock, %s %s, %d tasks, %dx%d images (%s per task), batch %d" % (a.method, a.model, a.tasks, a.hw, a.hw, a.sizes, a.batch), "unit": "s",
        "final_accuracies": final_acc,
        "value": round(t_first + t_seq, 2), "first_task_SI_s": round(t_first, 2), "task_sequence_s": round(t_seq, 2),
        "synthetic_data_generation_s (not counted)": round(t_data, 2), "train_image_passes": counts["train"],
        "eval_image_passes": counts["eval"], "epochs_cap": a.epochs, "lr_grid": 5, "data": "synthetic",
        "images_per_s_overall": round((counts["train"] + counts["eval"]) / (t_first + t_seq), 1),
        "n_models_evaluated": len(res)}
if a.cpu_train_rate > 0:
    cpu_s = counts["train"] / a.cpu_train_rate + counts["eval"] / (3.0 * a.cpu_train_rate)
    line["cpu_equivalent_s (estimate at the host-oracle rate)"] = round(cpu_s, 0)
    line["speedup_vs_cpu_estimate"] = round(cpu_s / (t_first + t_seq), 1)
print(json.dumps(line))
