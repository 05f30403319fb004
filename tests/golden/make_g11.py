"""G11: PackNet trainer fixture from the reference's UNCHANGED methods/packnet/main.py (dev container only).

Runs on CPU, on two tiny synthetic tasks (clsurvey_amd.framework.tasks.SyntheticTaskSequence), the call
sequence methods/method.py:PackNet issues: init_dump -> finetune(task 1, masks disabled) -> prune(+2 epochs)
-> finetune(task 2) -> prune(+2 epochs) -> eval(task 1), eval(task 2); torch.manual_seed(100 + stage) before
every stage.  Records the returned accuracies, the per-layer masks / weights of every checkpoint and the
module-index layout of the wrapper, i.e. what the build's own packnet_main.main must reproduce from the same
raw weights (tests/test_gpu_framework.py::test_packnet_trainer_g11).
"""
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
import torch.nn as nn  # noqa: E402
import models.VGGSlim as V  # noqa: E402
from g10_weights import det_weights  # noqa: E402

TINY = [32, "M", 32, "M", 32, 32, "M", 64, 64, "M"]
FC = (64, 64)
HW, NCLS = 32, 4
STAGES = ["ft1", "pr1", "ft2", "pr2"]
FT_EPOCHS, POST_EPOCHS, BATCH = 8, 2, 40
LR, PERC, WD = 0.01, 0.5, 5e-4


def main():
    V.cfg["tiny_VGG9"] = TINY
    root = tempfile.mkdtemp(prefix="g11_")
    import torch.utils.data as tud
    _DL = tud.DataLoader

    class DL(_DL):      # no worker processes / pinning in the container; order semantics unchanged
        def __init__(self, *a, **k):
            k["num_workers"] = 0
            k["pin_memory"] = False
            super().__init__(*a, **k)
    tud.DataLoader = DL
    torch.utils.data.DataLoader = DL
    import utilities.utils as uu
    uu.save_cuda_mem_req = lambda *a, **k: None        # needs a GPU in the reference
    import methods.packnet.main as PM
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=NCLS, sizes=(160, 40, 40),
                               hw=HW, noise=0.4, name="tiny2")
    paths = [ds.get_task_dataset_path(task_name=str(t)) for t in (1, 2)]

    m = V.VGGSlim(config="tiny_VGG9", num_classes=NCLS, classifier_inputdim=64 * 2 * 2, classifier_dim1=FC[0],
                  classifier_dim2=FC[1])
    with torch.no_grad():
        for p, q in zip(m.parameters(), det_weights(11, TINY, FC, NCLS, HW)):
            p.copy_(torch.from_numpy(q))
    raw = os.path.join(root, "raw.pth.tar")
    torch.save(m, raw)
    out = {"hyper": np.array([FT_EPOCHS, POST_EPOCHS, BATCH, LR, PERC, WD])}

    def snap(tag, path, weights=True, masks=True):
        ck = torch.load(path)
        model = ck["model"]
        out[tag + "_layout"] = np.array([[i, int(isinstance(mod, nn.Conv2d)), mod.weight.numel()]
                                         for i, mod in enumerate(model.shared.modules())
                                         if isinstance(mod, (nn.Conv2d, nn.Linear))])
        for i, mod in enumerate(model.shared.modules()):
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                if weights:
                    out["%s_w%d" % (tag, i)] = mod.weight.detach().numpy().copy()
                    out["%s_b%d" % (tag, i)] = mod.bias.detach().numpy().copy()
                if masks:
                    out["%s_m%d" % (tag, i)] = ck["previous_masks"][i].numpy().copy()
        for j, head in enumerate(model.classifiers):
            if weights:
                out["%s_hw%d" % (tag, j)] = head.weight.detach().numpy().copy()
                out["%s_hb%d" % (tag, j)] = head.bias.detach().numpy().copy()
        out[tag + "_datasets"] = np.array(list(model.datasets))
        if "accuracy" in ck:
            out[tag + "_ckpt_acc"] = np.array(float(ck["accuracy"]))

    init = os.path.join(root, "INIT_WRAPPED.pth")
    PM.main({"arch": "VGGslim_nopretrain", "init_dump": True, "cuda": False, "loadname": raw, "save_prefix": init,
             "last_layer_idx": 4, "current_dataset_idx": 1})
    snap("init", init, weights=False)
    prev = init
    accs = []
    for t in (1, 2):
        ft = os.path.join(root, "ft%d" % t, "best_model")
        os.makedirs(os.path.dirname(ft))
        torch.manual_seed(100 + 2 * t)
        a = PM.main({"weight_decay": WD, "disable_pruning_mask": t == 1, "train_path": paths[t - 1],
                     "test_path": paths[t - 1], "mode": "finetune", "dataset": "survey_TASK_%d" % t, "num_outputs": NCLS,
                     "loadname": prev, "lr": LR, "finetune_epochs": FT_EPOCHS, "cuda": False, "save_prefix": ft,
                     "batch_size": BATCH, "train_bn": False, "saving_freq": 100, "current_dataset_idx": t})
        accs.append(a)
        snap("ft%d" % t, ft + ".pth.tar", weights=(t == 1), masks=False)
        pr = os.path.join(root, "pr%d" % t, "best_model_PRUNED")
        os.makedirs(os.path.dirname(pr))
        torch.manual_seed(101 + 2 * t)
        a = PM.main({"weight_decay": WD, "train_path": paths[t - 1], "test_path": paths[t - 1], "mode": "prune",
                     "dataset": "survey_TASK_%d" % t, "loadname": ft + ".pth.tar", "post_prune_epochs": POST_EPOCHS,
                     "prune_perc_per_layer": PERC, "lr": LR * 0.1, "finetune_epochs": FT_EPOCHS, "cuda": False,
                     "save_prefix": pr, "train_bn": False, "saving_freq": 100, "current_dataset_idx": t,
                     "batch_size": BATCH})
        accs.append(a)
        snap("pp%d" % t, pr + "_postprune.pth.tar", weights=(t == 1))
        prev = pr + "_final.pth.tar" if os.path.exists(pr + "_final.pth.tar") else pr + "_postprune.pth.tar"
        out["pr%d_has_final" % t] = np.array(os.path.exists(pr + "_final.pth.tar"))
        snap("pr%d" % t, prev, weights=False)
    out["stage_acc"] = np.array(accs, dtype=np.float64)
    ev = []
    for t in (1, 2):
        torch.manual_seed(200 + t)
        ev.append(PM.main({"train_path": paths[t - 1], "test_path": paths[t - 1], "mode": "eval",
                           "dataset": "survey_TASK_%d" % t, "loadname": prev, "cuda": False, "batch_size": BATCH,
                           "current_dataset_idx": t}))
    out["eval_acc"] = np.array(ev, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "G11_packnet_trainer.npz"), **out)
    print("stage_acc", out["stage_acc"], "eval_acc", out["eval_acc"])
    print(os.path.getsize(os.path.join(HERE, "G11_packnet_trainer.npz")) // 1024, "KiB")
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
