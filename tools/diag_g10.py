#!/usr/bin/env python
"""Diagnostic: EWC task-3 training of the G10 flow, build (GPU) vs CPU oracle from the SAME task-2 model."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from g10_weights import det_weights, SMALL  # noqa: E402
from clsurvey_amd import models  # noqa: E402
from clsurvey_amd.data import DeviceLoader  # noqa: E402
from clsurvey_amd.framework import driver  # noqa: E402
from clsurvey_amd.framework.tasks import SyntheticTaskSequence  # noqa: E402
from clsurvey_amd.methods import ewc as EW, method as M, train_common as tc  # noqa: E402
from oracle import regularizers_ref as R, vgg_ref  # noqa: E402

root = tempfile.mkdtemp()
COMMON = ["small_VGG9_cl_128_128", "--lr_grid", "1e-2,3e-3", "--num_epochs", "8", "--batch_size", "40", "--saving_freq",
          "100", "--drop_margin", "0.05"]
ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=3, classes_per_task=4, sizes=(160, 40, 40), hw=32,
                           noise=0.4, name="tiny3")
m = models.parse_model_name("small_VGG9_cl_128_128", (32, 32), 4)
with torch.no_grad():
    for p, w in zip(m.parameters(), det_weights()):
        p.copy_(torch.from_numpy(w))
os.makedirs(os.path.join(root, "models"))
torch.save(m, os.path.join(root, "models", "small_VGG9_cl_128_128.pth.tar"))
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    driver.main(COMMON + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    ewc = M.parse("EWC"); ewc.hyperparams["lambda"] = 40.0
    out = driver.main(COMMON + ["--method_name", "EWC", "--results_root", root, "--test", "--max_task_count", "2"],
                      method=ewc, dataset=ds)
print("seq_res after 2 tasks:", {i: out["results"][i]["seq_res"][i] for i in out["results"]})
m2_path = out["model_paths"][1]
d2 = torch.load(ds.get_task_dataset_path("2"), weights_only=False)
d3 = torch.load(ds.get_task_dataset_path("3"), weights_only=False)


def test_acc_cpu(params, head, dset):
    ps = list(params[:-2]) + list(head)
    with torch.no_grad():
        return float((vgg_ref.forward(ps, SMALL, dset.x).argmax(1) == dset.y).float().mean())


# ---------------- build: EWC task 3 from the task-2 model
driver.set_random(0)
with contextlib.redirect_stdout(buf):
    model3, acc3 = EW.fine_tune_EWC_acuumelation(d3, m2_path, os.path.join(root, "t3_build"), None, [d2], reg_lambda=40.0,
                                                 num_epochs=8, lr=1e-2, batch_size=40)
best3 = torch.load(os.path.join(root, "t3_build", "best_model.pth.tar"), weights_only=False)
m2 = torch.load(m2_path, weights_only=False)
head2 = [p.detach().cpu() for p in list(m2.parameters())[-2:]]
p3 = [p.detach().cpu() for p in best3.parameters()]
print("BUILD : best val acc task3 %.3f ; test acc on task2 with head2 = %.3f ; on task3 = %.3f" %
      (acc3, test_acc_cpu(p3, head2, d2["test"]), test_acc_cpu(p3, p3[-2:], d3["test"])))
rp = best3.reg_params
plist = list(best3.parameters())
print("BUILD : omega max per param:", ["%.3g" % float(rp[p]["omega"].max()) if p in rp else "-" for p in plist])
print("BUILD : |theta-init| max   :", ["%.3g" % float((p.data - rp[p]["init_val"]).abs().max()) if p in rp else "-" for p in plist])

# ---------------- oracle: same thing on the CPU from the same model, same seeds / loader order
driver.set_random(0)
theta = [p.detach().cpu().clone() for p in m2.parameters()]
rp2 = m2.reg_params
prev = [rp2[p]["omega"].detach().cpu().clone() if p in rp2 else None for p in m2.parameters()]
fl = DeviceLoader(d2["train"], 40, False, "cpu")
new = R.diag_fisher(theta, SMALL, list(fl), len(d2["train"]))
omega = [(pv + nw) if pv is not None else None for pv, nw in zip(prev, new)]
init = [t.clone() if o is not None else None for t, o in zip(theta, omega)]
head = torch.nn.Linear(128, 4)
theta[-2], theta[-1] = head.weight.detach().clone(), head.bias.detach().clone()
omega[-2] = omega[-1] = init[-2] = init[-1] = None
loaders = {x: DeviceLoader(d3[x], 40, True, "cpu") for x in ("train", "val")}
bufs = [None] * len(theta)
best, best_theta, count, lr, first = 0.0, None, 0, 1e-2, True
for ep in range(8):
    if count > 10:
        break
    if count == 5:
        lr *= 0.1
    for x, y in loaders["train"]:
        _, _, g, _ = vgg_ref.loss_and_grads(theta, SMALL, x, y, "ce_mean")
        nn_ = [R.reg_sgd_step(t, gi, o, iv, b, 40.0, lr, 0.9, 0.0, first) for t, gi, o, iv, b in zip(theta, g, omega, init, bufs)]
        theta, bufs = [a[0] for a in nn_], [a[1] for a in nn_]
        first = False
    corr = 0
    for x, y in loaders["val"]:
        with torch.no_grad():
            corr += int((vgg_ref.forward(theta, SMALL, x).argmax(1) == y).sum())
    acc = corr / len(d3["val"])
    if acc > best:
        best, best_theta, count = acc, [t.clone() for t in theta], 0
    else:
        count += 1
print("ORACLE: best val acc task3 %.3f ; test acc on task2 with head2 = %.3f ; on task3 = %.3f" %
      (best, test_acc_cpu(best_theta, head2, d2["test"]), test_acc_cpu(best_theta, best_theta[-2:], d3["test"])))
print("ORACLE: omega max per param:", ["%.3g" % float(o.max()) if o is not None else "-" for o in omega])
print("ORACLE: |theta-init| max   :", ["%.3g" % float((t - iv).abs().max()) if iv is not None else "-" for t, iv in zip(best_theta, init)])
