#!/usr/bin/env python
"""Where does a non-finite value first appear in the bench's EWC sweep?  Runs the first TASKS tasks of the sweep through the driver with
finiteness checks hooked into diag_fisher (gradient arena after every batch, omega at the end) and into the penalised step (loss,
gradient arena, parameters after every batch of the first epoch).  usage: nan_diag.py [tasks]   (CLHIP_LIB selects the build)"""
import contextlib, io, os, sys, tempfile, shutil
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from clsurvey_amd import ops
from clsurvey_amd.framework import driver
from clsurvey_amd.framework.tasks import SyntheticTaskSequence
from clsurvey_amd.methods import ewc, method as M
from clsurvey_amd import net

tasks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
first = {}

def note(tag, t):
    if t is None:
        return
    bad = ~torch.isfinite(t)
    if bool(bad.any()) and tag not in first:
        idx = int(bad.nonzero()[0])
        first[tag] = (idx, float(t.reshape(-1)[idx]), int(bad.sum()))
        sys.stderr.write("NONFINITE %s first at flat index %d value %r count %d\n" % (tag, idx, float(t.reshape(-1)[idx]), int(bad.sum())))

orig_fisher = ops.fisher_accum
calls = {"fisher": 0, "reg": 0}
def fisher_accum(omega, grad, n):
    calls["fisher"] += 1
    note("fisher_grad_call%d" % calls["fisher"] if not torch.isfinite(grad).all() else "fisher_grad", grad)
    r = orig_fisher(omega, grad, n)
    note("fisher_omega", omega)
    return r
ops.fisher_accum = fisher_accum
ewc.ops.fisher_accum = fisher_accum
orig_reg = ops.reg_sgd_step
def reg_sgd_step(theta, grad, omega, init_val, buf, lam, lr, mom, wd, firststep):
    calls["reg"] += 1
    if not torch.isfinite(grad).all():
        note("reg_grad_call%d" % calls["reg"], grad)
    note("reg_omega_in", omega)
    note("reg_init_in", init_val)
    r = orig_reg(theta, grad, omega, init_val, buf, lam, lr, mom, wd, firststep)
    if not torch.isfinite(theta).all():
        note("reg_theta_after_call%d" % calls["reg"], theta)
    return r
ops.reg_sgd_step = reg_sgd_step
import clsurvey_amd.optim as optim
if hasattr(optim, "ops"):
    optim.ops.reg_sgd_step = reg_sgd_step

# the first loss_step whose gradient arena is not finite: which activation, which parameters
orig_loss_step = net.NetEngine.loss_step
seen = {"done": False, "n": 0}
def loss_step(self, x, y, kind="ce_mean", backward=True, stats=None, **kw):
    r = orig_loss_step(self, x, y, kind, backward, stats, **kw)
    seen["n"] += 1
    if backward and not seen["done"] and not bool(torch.isfinite(self.arena.grad).all()):
        seen["done"] = True
        n = x.shape[0]
        sys.stderr.write("LOSS_STEP %d (%s, N=%d): non-finite gradient; loss %r; input finite %s max|x| %.3g; theta finite %s max|theta| %.3g\n" % (
            seen["n"], kind, n, float(r[0]), bool(torch.isfinite(x).all()), float(x.abs().max()), bool(torch.isfinite(self.arena.theta).all()),
            float(self.arena.theta.abs().max())))
        l = 1
        while True:
            try:
                a = self.layer_input(l, n)
            except Exception:
                break
            bad = ~torch.isfinite(a)
            fin = a[torch.isfinite(a)]
            sys.stderr.write("  input of plan layer %d: %d values/sample, non-finite %d (samples %s), max finite |a| %.4g, +inf %d -inf %d nan %d\n" % (
                l, a.shape[1], int(bad.sum()), bad.any(1).nonzero().reshape(-1)[:6].tolist(), float(fin.abs().max()) if fin.numel() else -1.0,
                int((a == float("inf")).sum()), int((a == float("-inf")).sum()), int(torch.isnan(a).sum())))
            if bool(bad.any()):
                smp = int(bad.any(1).nonzero()[0]); cols = bad[smp].nonzero().reshape(-1)
                sys.stderr.write("    sample %d: %d bad features, first %s, values %s\n" % (smp, cols.numel(), cols[:8].tolist(), a[smp, cols[:8]].tolist()))
            l += 1
        for pi, prm in enumerate(self.arena.params):
            gv = self.arena.view("grad", prm)
            if not bool(torch.isfinite(gv).all()):
                sys.stderr.write("  grad of parameter %d %s: %d non-finite of %d\n" % (pi, tuple(prm.shape), int((~torch.isfinite(gv)).sum()), gv.numel()))
    return r
net.NetEngine.loss_step = loss_step

root = tempfile.mkdtemp(prefix="clhip_nandiag_")
try:
    groot = os.path.join(root, "gpu")
    ds = SyntheticTaskSequence(os.path.join(groot, "data"), task_count=tasks, classes_per_task=20, sizes=(8000, 2000, 1000), hw=64,
                               name="synthetic_tiny_imagenet", noise=bench.SWEEP_DATA["noise"], kind=bench.SWEEP_DATA["kind"],
                               blobs=bench.SWEEP_DATA["blobs"], seed=7)
    common = ["small_VGG9_cl_128_128", "--num_epochs", "70", "--results_root", groot, "--device", "cuda:0"]
    quiet = io.StringIO()
    try:
        with contextlib.redirect_stdout(quiet):
            driver.main(common + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"), dataset=ds)
            out = driver.main(common + ["--method_name", "EWC", "--test"], method=M.parse("EWC"), dataset=ds)
        print("sweep finished; accepted lambdas:", [float(hf.trace[-1][0]["lambda"]) for hf in out["frameworks"] if hf is not None and hf.trace])
    except BaseException as e:
        print("sweep failed:", type(e).__name__, str(e)[:200])
    lines = quiet.getvalue().splitlines()
    for ln in [l for l in lines if "ATTEMPT" in l or "FINETUNE DONE" in l or "nan" in l.lower()][-30:]:
        print("  ", ln[:160])
    print("first non-finite sightings:", first)
    print("calls:", calls)
finally:
    shutil.rmtree(root, ignore_errors=True)
