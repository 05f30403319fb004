"""Oracle: an EWC method object for the framework driver that does ALL its arithmetic with the torch-CPU restatements
of this package (vgg_ref / regularizers_ref) — TEST / BENCH INFRASTRUCTURE, never imported by the product.

bench.py's cpu_baseline leg hands it to clsurvey_amd.framework.driver.main to run, on the host cores, the SAME bounded
sweep it has just run on the GPU (same task files, LR grid, epoch cap, batch size, decay rule): phase-1 finetune per LR
(Finetune/train_SGD.py:41-189), phase 2 = Fisher diagonal on the previous task (EWC/main_EWC.py:138-157), omega
accumulation (:205-232), penalised momentum SGD (train_EWC.py:23-86) with the count-based LR drop / early stop (:89-101),
best-validation model saved, and the final top-1 evaluation of every model on every task (framework/inference.py:8-87).
The pickled artefacts are plain nn.Modules of standard torch layers (the wire format between the framework's layers), so
the first-task model written by the GPU run is a valid starting point here.
"""
import os
import time
from collections import OrderedDict

import torch

from . import regularizers_ref as R
from . import vgg_ref


def _params_of(model):
    return [p.detach().cpu().clone() for p in model.parameters()]


def _store(model, params):
    with torch.no_grad():
        for p, v in zip(model.parameters(), params):
            p.copy_(v)
    return model


def _omega_of_reg_params(model):
    """Omega of a model file written by the reference's own EWC code (or by the build's HIP path, which writes the same
    artefact): `model.reg_params[param]["omega"]` per parameter (EWC/main_EWC.py:160-232), in model.parameters() order; None
    for a parameter without an entry (the fresh head of the last training).  Lets a training of this oracle start from such a
    file — bench.py's teacher-forced chain hands every task's CPU leg the GPU run's model of the task before."""
    rp = getattr(model, "reg_params", None)
    if not rp:
        return None
    out = []
    for p in model.parameters():
        e = rp.get(p)
        out.append(e["omega"].detach().cpu().clone() if isinstance(e, dict) and "omega" in e else None)
    return out if any(o is not None for o in out) else None


def _order(n, shuffle):
    """Sample order of one pass of torch.utils.data.DataLoader(dataset, batch_size, shuffle) — what the reference's loops
    iterate (Finetune/main_SGD.py, EWC/main_EWC.py:138-157): every iterator first draws its worker base seed from the global
    generator (_BaseDataLoaderIter.__init__), a shuffling one then seeds a private generator from the global one
    (RandomSampler.__iter__) and draws the permutation from that.  Following the protocol draw for draw keeps this oracle on
    the same batches as any other restatement of those loops that starts from the same seed."""
    torch.empty((), dtype=torch.int64).random_()
    if not shuffle:
        return torch.arange(n)
    g = torch.Generator()
    g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    return torch.randperm(n, generator=g)


def _batches(dset, batch_size, shuffle):
    n = len(dset)
    order = _order(n, shuffle)
    x, y = dset.x.cpu(), dset.y.cpu()
    for i in range(0, n, batch_size):
        idx = order[i:i + batch_size]
        yield x[idx], y[idx]


class OracleEWC:
    name = eval_name = "EWC"
    category = None
    extra_hyperparams_count = 1

    def __init__(self, cfg_name="small_VGG9"):
        self.hyperparams = OrderedDict([("lambda", 400)])
        self.cfg = vgg_ref.CFGS[cfg_name]
        self.image_passes = {"train": 0, "eval": 0}
        self.seconds = {"train": 0.0, "eval": 0.0}           # host seconds spent in forward+backward+update / forward-only loops

    # ------------------------------------------------------------------ shared epoch loop
    def _fit(self, model, params, omega, init, lam, lr, dsets, args, exp_dir):
        os.makedirs(exp_dir, exist_ok=True)
        bufs = [None] * len(params)
        best_acc, count, first = 0.0, 0, True
        for _ in range(args.num_epochs):
            if count > 10:
                break
            if count == 5:
                lr *= 0.1
            t0 = time.perf_counter()
            running_loss = 0.0
            for x, y in _batches(dsets["train"], args.batch_size, True):
                _, loss, grads, _ = vgg_ref.loss_and_grads(params, self.cfg, x, y, "ce_mean")
                running_loss += float(loss)
                stepped = [R.reg_sgd_step(t, g, o, iv, b, lam, lr, 0.9, args.weight_decay, first)
                           for t, g, o, iv, b in zip(params, grads, omega, init, bufs)]
                params, bufs = [s[0] for s in stepped], [s[1] for s in stepped]
                first = False
                self.image_passes["train"] += x.shape[0]
            t1 = time.perf_counter()
            epoch_loss = running_loss / len(dsets["train"])
            if epoch_loss > 1e4 or epoch_loss != epoch_loss:          # train_EWC.py:204-205: a diverged run returns what it has
                self.seconds["train"] += t1 - t0
                return best_acc
            hits = 0
            with torch.no_grad():
                for x, y in _batches(dsets["val"], args.batch_size, True):
                    hits += int((vgg_ref.forward(params, self.cfg, x).argmax(1) == y).sum())
                    self.image_passes["eval"] += x.shape[0]
            self.seconds["train"] += t1 - t0
            self.seconds["eval"] += time.perf_counter() - t1
            acc = hits / float(len(dsets["val"]))
            if acc > best_acc:
                best_acc, count = acc, 0
                torch.save(_store(model, params), os.path.join(exp_dir, "best_model.pth.tar"))
            else:
                count += 1
        return best_acc

    def _fresh_head(self, model, params, n_out):
        last = str(len(model.classifier._modules) - 1)
        head = torch.nn.Linear(model.classifier._modules[last].in_features, n_out)
        model.classifier._modules[last] = head
        return params[:-2] + [head.weight.detach().clone(), head.bias.detach().clone()]

    # ------------------------------------------------------------------ plugin surface
    def grid_train(self, args, manager, lr):
        dsets = torch.load(manager.current_task_dataset_path, weights_only=False)
        model = torch.load(manager.previous_task_model_path, map_location="cpu", weights_only=False)
        model.reg_params = {}
        params = self._fresh_head(model, _params_of(model), len(dsets["train"].classes))
        none = [None] * len(params)
        return None, self._fit(model, params, none, none, 0.0, lr, dsets, args, manager.gridsearch_exp_dir)

    def train(self, args, manager, hyperparams):
        dsets = torch.load(manager.current_task_dataset_path, weights_only=False)
        prev = torch.load(manager.reg_sets[-1], weights_only=False)["train"]
        model = torch.load(manager.previous_task_model_path, map_location="cpu", weights_only=False)
        theta = _params_of(model)
        t0 = time.perf_counter()
        fisher = R.diag_fisher(theta, self.cfg, list(_batches(prev, args.batch_size, False)), len(prev))
        self.image_passes["train"] += len(prev)
        self.seconds["train"] += time.perf_counter() - t0
        older = getattr(model, "oracle_omega", None)                   # omega accumulated over the earlier tasks
        if older is None:
            older = _omega_of_reg_params(model)
        omega = [f if o is None else o + f for f, o in zip(fisher, older or [None] * len(fisher))]
        init = [t.clone() for t in theta]
        params = self._fresh_head(model, theta, len(dsets["train"].classes))
        omega_fit, init_fit = omega[:-2] + [None, None], init[:-2] + [None, None]       # the fresh head is free
        model.oracle_omega = [o.clone() for o in omega]
        acc = self._fit(model, params, omega_fit, init_fit, float(hyperparams["lambda"]), args.lr, dsets, args,
                        manager.heuristic_exp_dir)
        return None, acc

    def get_output(self, images, holder):
        raise NotImplementedError("the oracle evaluates through inference_eval")

    def inference_eval(self, args, manager):
        dsets = torch.load(args.dset_path, weights_only=False)
        split = dsets[args.test_set if args.test_set in dsets else "val"]
        model = torch.load(args.eval_model_path, map_location="cpu", weights_only=False)
        head_model = torch.load(args.head_paths, map_location="cpu", weights_only=False)
        params = _params_of(model)[:-2] + _params_of(head_model)[-2:]
        hits = 0
        t0 = time.perf_counter()
        with torch.no_grad():
            for x, y in _batches(split, args.batch_size, True):             # (framework/inference.py:27 shuffles its test loader)
                hits += int((vgg_ref.forward(params, self.cfg, x).argmax(1) == y).sum())
                self.image_passes["eval"] += x.shape[0]
        self.seconds["eval"] += time.perf_counter() - t0
        return 100.0 * hits / len(split)
