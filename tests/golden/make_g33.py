"""G33: teacher-forced evaluation outputs and SI consolidation from the reference's own code (dev container only).

The end-to-end fixtures (G10 / G17) compare accuracies after two independently evolving SGD trajectories, which pins
the test accuracy of the just-trained task to a few samples and older-task accuracies / forgetting not at all.  This
fixture removes the trajectory: the reference's UNCHANGED framework (`framework/main.py --test`, as in make_g10.py) runs
SI `first_task_basemodel_dump` and a 3-task EWC sweep, and what is stored is
  * the parameters of every model the evaluation reads (task-1 SI checkpoint, EWC task-2 / task-3 best_model), so the
    build's evaluation can be fed the reference-trained weights;
  * `seq_res` / `seq_forgetting` exactly as `framework/eval.py:146-247` wrote them to test_method_performances*.pth;
  * for every (task i, model j >= i) pair the reference's logits on the test split in dataset order, computed through
    the same calls `framework/inference.py:8-87` makes (utils.get_prev_heads :235-262 + method.get_output) — they let
    the test tell a wrong prediction from a near-tie.
SI part (`methods/SI/train_SI.py:301-364 update_reg_params`): the task-1 SI checkpoint carries the path integral `w`
of its whole training; the reference's consolidation on exactly that checkpoint gives Omega.  Consolidation is
element-wise, so per parameter tensor the inputs (theta, init_val, w, Omega before) and the output Omega are stored at a
fixed sample of positions (whole tensor up to 2^12 elements).
"""
import copy
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import harness  # noqa: E402

torch = harness.install()
import utilities.utils as utils  # noqa: E402
from g10_weights import det_weights  # noqa: E402

MODEL = "small_VGG9_cl_128_128"
LAMBDA0 = 40.0
COMMON = [MODEL, "--lr_grid", "1e-2,3e-3", "--num_epochs", "8", "--batch_size", "40", "--saving_freq", "100", "--drop_margin", "0.05"]
BIG, SAMPLE = 1 << 12, 4096


def sample_index(numel):
    return np.sort(np.random.RandomState(33).choice(numel, SAMPLE, replace=False))


def main():
    root = tempfile.mkdtemp(prefix="g33_")
    with open(os.path.join(root, "config.init"), "w") as f:
        f.write("[DEFAULT]\ntest_results_root_path='./results/test'\ntr_results_root_path='./results/train'\n"
                "models_root_path='./data/models'\nds_root_path='./data/datasets'\n")
    utils.get_root_src_path = lambda: root
    torch.cuda.is_available = lambda: False
    import torch.utils.data as tud
    _DL = tud.DataLoader

    class DL(_DL):      # no worker processes / pinning in the container; order semantics unchanged
        def __init__(self, *a, **k):
            k["num_workers"] = 0
            k["pin_memory"] = False
            super().__init__(*a, **k)
    tud.DataLoader = DL
    torch.utils.data.DataLoader = DL

    import collections
    import framework.main as ref_main
    import methods.method as ref_methods
    import methods.SI.train_SI as ref_si
    import models.VGGSlim as V
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    ds = SyntheticTaskSequence(os.path.join(root, "data", "datasets"), task_count=3, classes_per_task=4,
                               sizes=(160, 40, 40), hw=32, noise=0.4, name="tiny3")
    mdir = os.path.join(root, "data", "models", "customVGG_input=32x32")
    os.makedirs(mdir)
    m = V.VGGSlim(config="small_VGG9", num_classes=4, classifier_inputdim=128 * 2 * 2, classifier_dim1=128,
                  classifier_dim2=128)
    with torch.no_grad():
        for p, w in zip(m.parameters(), det_weights()):
            p.copy_(torch.from_numpy(w))
    torch.save(m, os.path.join(mdir, MODEL + ".pth.tar"))

    sys.argv = ["main.py"] + COMMON + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"]
    ref_main.main(method=ref_methods.parse("SI"), dataset=ds)
    sys.argv = ["main.py"] + COMMON + ["--method_name", "EWC", "--test"]
    ewc = ref_methods.EWC()
    ewc.hyperparams = collections.OrderedDict({"lambda": LAMBDA0})      # stable regime, see make_g10.py
    ref_main.main(method=ewc, dataset=ds)

    out = {}
    tr = os.path.join(root, "results", "train", "tiny3")
    si_root = os.path.join(tr, "SI", MODEL, "gridsearch", "first_task_basemodel")
    si_ckpt = os.path.join(si_root, os.listdir(si_root)[0], "task_1", "TASK_TRAINING", "best_model.pth.tar")
    exp = os.listdir(os.path.join(tr, "EWC", MODEL, "gridsearch", "demo"))[0]
    base = os.path.join(tr, "EWC", MODEL, "gridsearch", "demo", exp)
    model_paths = [si_ckpt] + [os.path.join(base, "task_%d" % t, "TASK_TRAINING", "best_model.pth.tar") for t in (2, 3)]
    ds_paths = [ds.get_task_dataset_path(str(t)) for t in (1, 2, 3)]
    out["exp_name"] = np.array(exp)
    for j, path in enumerate(model_paths):
        mj = torch.load(path)
        for i, p in enumerate(mj.parameters()):
            out["m%d_p%d" % (j, i)] = p.detach().numpy().copy()

    # seq_res / seq_forgetting as the reference's evaluation saved them
    te = os.path.join(root, "results", "test", "results", "tiny3", "EWC", MODEL, "demo", exp)
    for i in range(3):
        r = torch.load(os.path.join(te, "test_method_performancesEWC%d.pth" % i))["EWC"]
        out["seq_res%d" % i] = np.array(r["seq_res"][i], dtype=np.float64)
        out["seq_forgetting%d" % i] = np.array(r["seq_forgetting"][i], dtype=np.float64)

    # logits of every (task, model) pair through the calls of inference.test_model, test split in dataset order
    for i in range(3):
        dsets = torch.load(ds_paths[i])
        xs = torch.stack([dsets["test"][n][0] for n in range(len(dsets["test"]))])
        ys = torch.tensor([int(dsets["test"][n][1]) for n in range(len(dsets["test"]))])
        for j in range(i, 3):
            model = torch.load(model_paths[j])
            final = str(len(model.classifier._modules) - 1)
            heads = utils.get_prev_heads(model_paths[i], final)
            holder = type("Holder", (object,), {})()
            holder.model, holder.heads, holder.current_head_idx, holder.final_layer_idx = model, heads, 0, final
            holder.task_idx, holder.batch_size, holder.task_imgfolders = i, 40, dsets
            model.eval()
            with torch.no_grad():
                logits = ewc.get_output(xs, holder)
            out["logits_%d_%d" % (i, j)] = logits.numpy().copy()
            acc = 100.0 * float((logits.argmax(1) == ys).float().mean())
            assert abs(acc - float(out["seq_res%d" % i][j - i])) < 1e-4, (i, j, acc, out["seq_res%d" % i])

    # SI consolidation on the reference's task-1 checkpoint
    model = torch.load(si_ckpt)
    before = {p: {k: copy.deepcopy(v) for k, v in model.reg_params[p].items() if torch.is_tensor(v)}
              for p in model.parameters() if p in model.reg_params}
    theta = [p.detach().clone() for p in model.parameters()]
    reg = ref_si.update_reg_params(model)
    for i, p in enumerate(model.parameters()):
        n = p.numel()
        idx = np.arange(n) if n <= BIG else sample_index(n)
        pick = lambda t: t.detach().float().numpy().reshape(-1)[idx].copy()      # noqa: E731
        out["si_theta%d" % i] = pick(theta[i])
        out["si_init%d" % i] = pick(before[p]["init_val"])
        out["si_w%d" % i] = pick(before[p]["w"])
        out["si_omega_before%d" % i] = pick(before[p]["omega"])
        out["si_omega_after%d" % i] = pick(reg[p]["omega"])
        assert torch.equal(reg[p]["init_val"], theta[i]) and float(reg[p]["w"].abs().max()) == 0.0
    np.savez_compressed(os.path.join(HERE, "G33_teacher_forced_eval.npz"), **out)
    for k, v in out.items():
        if not k.startswith("m") or k.endswith("_p0"):
            print(k, v.shape if v.size > 8 else v)
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
