"""G18: end-to-end fixtures for LwF and EBLL from the reference's UNCHANGED framework/main.py
(dev container only) — the same procedure as make_g10.py for EWC: SI `first_task_basemodel_dump`, then a 3-task sweep with
--test per method on the tiny synthetic tasks, from deterministic start weights.  Recorded per method: per-LR validation
accuracies of every phase-1 grid, every phase-2 attempt (task, lambda, accuracy), the final (attempts, lambda, threshold)
state, seq_res / seq_forgetting, the number of stacked heads and (EBLL) the code sizes of the chosen autoencoders (grid: 4
epochs, alpha 0.1, dims 16 / 8).
"""
import collections
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
COMMON = [MODEL, "--lr_grid", "1e-2,3e-3", "--num_epochs", "8", "--batch_size", "40", "--saving_freq", "100", "--drop_margin", "0.05"]
# start values inside the stable regime of these tiny tasks (see make_g10.py for why the defaults of 400 are not)
LAMBDA0 = {"LWF": 10.0, "EBLL": 10.0}
ATTEMPTS = []


def main():
    root = tempfile.mkdtemp(prefix="g18_")
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

    import framework.main as ref_main
    import methods.method as ref_methods
    import models.VGGSlim as V
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    ds = SyntheticTaskSequence(os.path.join(root, "data", "datasets"), task_count=3, classes_per_task=4,
                               sizes=(160, 40, 40), hw=32, noise=0.4, name="tiny3")
    mdir = os.path.join(root, "data", "models", "customVGG_input=32x32")
    os.makedirs(mdir)
    m = V.VGGSlim(config="small_VGG9", num_classes=4, classifier_inputdim=128 * 2 * 2, classifier_dim1=128, classifier_dim2=128)
    with torch.no_grad():
        for p, w in zip(m.parameters(), det_weights()):
            p.copy_(torch.from_numpy(w))
    torch.save(m, os.path.join(mdir, MODEL + ".pth.tar"))

    out = {}
    sys.argv = ["main.py"] + COMMON + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"]
    ref_main.main(method=ref_methods.parse("SI"), dataset=ds)
    tr = os.path.join(root, "results", "train", "tiny3")
    for name in ("LWF", "EBLL"):
        global ATTEMPTS
        ATTEMPTS = []
        base_cls = getattr(ref_methods, name)

        def train(self, args, manager, hyperparams, _base=base_cls):       # instrumentation only: record EVERY attempt
            model, acc = _base.train(self, args, manager, hyperparams)
            ATTEMPTS.append((args.task_counter, float(hyperparams["lambda" if "lambda" in hyperparams else "reg_lambda"]), float(acc)))
            return model, acc
        cls = type("Logged" + name, (base_cls,), {"train": train})
        globals()["Logged" + name] = cls            # picklable by reference (the framework pickles vars(manager))
        cls.__qualname__ = "Logged" + name
        cls.__module__ = __name__
        meth = cls()
        if name == "LWF":
            meth.hyperparams = collections.OrderedDict({"lambda": LAMBDA0[name]})
        else:
            meth.hyperparams = collections.OrderedDict({"reg_lambda": LAMBDA0[name], "ebll_reg_alpha": 1})
            meth.static_hyperparams = collections.OrderedDict({"autoencoder_lr": [0.01], "autoencoder_epochs": 4,
                                                               "encoder_alphas": [1e-1], "encoder_dims": [16, 8]})
        sys.argv = ["main.py"] + COMMON + ["--method_name", name, "--test"]
        ref_main.main(method=meth, dataset=ds)
        pre = name.lower() + "_"
        out[pre + "attempts"] = np.array(ATTEMPTS, dtype=np.float64)
        out[pre + "lambda0"] = np.array(float(LAMBDA0[name]))
        exp = [d for d in os.listdir(os.path.join(tr, name, MODEL, "gridsearch", "demo"))][0]
        out[pre + "exp_name"] = np.array(exp)
        base = os.path.join(tr, name, MODEL, "gridsearch", "demo", exp)
        for t in (2, 3):
            g = torch.load(os.path.join(base, "task_%d" % t, "FT_LR_GRIDSEARCH", "grid_checkpoint.pth"))
            for lr, d in g["processed_lrs"].items():
                out[pre + "t%d_lr%g" % (t, lr)] = np.array(d["acc"], dtype=np.float64)
            h = torch.load(os.path.join(base, "task_%d" % t, "TASK_TRAINING", "hyperparams.pth.tar"))
            out[pre + "t%d_val_acc" % t] = np.array(float(h["val_acc"]))
            out[pre + "t%d_threshold" % t] = np.array(float(h["acc_threshold"]))
            out[pre + "t%d_attempts" % t] = np.array(int(h["state"]["attempts"]))
            hp = h["state"]["hyperparams"]
            out[pre + "t%d_lambda" % t] = np.array(float(hp["lambda"] if "lambda" in hp else hp["reg_lambda"]))
            mt = torch.load(os.path.join(base, "task_%d" % t, "TASK_TRAINING", "best_model.pth.tar"))
            out[pre + "t%d_n_heads" % t] = np.array(len(mt.model.classifier._modules if hasattr(mt, "model") else mt.classifier._modules))
            if name == "EBLL":
                out[pre + "t%d_n_encoders" % t] = np.array(len(mt.autoencoders._modules))
                out[pre + "t%d_code_dim" % t] = np.array([m[0].out_features for m in mt.autoencoders._modules.values()])
        te = os.path.join(root, "results", "test", "results", "tiny3", name, MODEL, "demo", exp)
        for i in range(3):
            r = torch.load(os.path.join(te, "test_method_performances%s%d.pth" % (name, i)))[name]
            out[pre + "seq_res%d" % i] = np.array(r["seq_res"][i], dtype=np.float64)
            out[pre + "seq_forgetting%d" % i] = np.array(r["seq_forgetting"][i], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "G18_framework_lwf_ebll.npz"), **out)
    for k, v in out.items():
        print(k, v if v.size < 8 else v.shape)
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
