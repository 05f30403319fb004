"""G12: HAT trainer fixture from the reference's UNCHANGED methods/HAT/run.py (dev container only).

Runs on CPU, on two tiny synthetic tasks, the call sequence methods/method.py:HAT issues through
_modular_accespoint: task 1 grid_train (hat_finetune.Appr, is_scratch) -> train (hat.Appr with warm-up);
task 2 grid_train -> train from the task-1 model.  torch.manual_seed(300 + stage) before every stage.
Records accuracies, the task-1 model's parameters (start of the isolated task-2 checks), cumulative masks and
back-mask statistics, and the final gates.
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
import models.VGGSlim as V  # noqa: E402
from g10_weights import det_weights  # noqa: E402

CFG = [32, "M", 32, "M", 32, 32, "M", 64, 64, "M"]
FC = (64, 64)
HW, NCLS = 32, 4
NEPOCHS, BATCH, LR, WD = 12, 40, 0.01, 5e-4
PARAM = [50.0, 0.75]        # smax, c


def main():
    V.cfg["tiny_VGG9"] = CFG
    root = tempfile.mkdtemp(prefix="g12_")
    import torch.utils.data as tud
    _DL = tud.DataLoader

    class DL(_DL):
        def __init__(self, *a, **k):
            k["num_workers"] = 0
            k["pin_memory"] = False
            super().__init__(*a, **k)
    tud.DataLoader = DL
    torch.utils.data.DataLoader = DL
    torch.cuda.is_available = lambda: True          # run.py:42 exits otherwise; .cuda() is identity (harness)
    import methods.HAT.run as RUN
    import methods.HAT.approaches.hat as HATA
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=NCLS, sizes=(160, 40, 40),
                               hw=HW, noise=0.4, name="tiny2")
    paths = [ds.get_task_dataset_path(task_name=str(t)) for t in (1, 2)]
    m = V.VGGSlim(config="tiny_VGG9", num_classes=NCLS, classifier_inputdim=64 * 2 * 2, classifier_dim1=FC[0],
                  classifier_dim2=FC[1])
    with torch.no_grad():
        for p, q in zip(m.parameters(), det_weights(12, CFG, FC, NCLS, HW)):
            p.copy_(torch.from_numpy(q))
    raw = os.path.join(root, "raw.pth.tar")
    torch.save(m, raw)
    out = {"hyper": np.array([NEPOCHS, BATCH, LR, WD] + PARAM)}

    def run(stage, t, prev, finetune, outdir):
        os.makedirs(outdir, exist_ok=True)
        torch.manual_seed(300 + stage)
        return RUN.main({"weight_decay": WD, "task_name": str(t), "task_count": t, "prev_model_path": prev,
                         "model_name": "tiny_VGG9_cl_64_64", "output": outdir, "nepochs": NEPOCHS, "parameter": PARAM,
                         "cuda": True, "dataset_path": paths[t - 1], "dataset": ds, "n_tasks": 2, "batch_size": BATCH,
                         "lr": LR, "is_scratch_model": t == 1, "approach": "hat", "nc_per_task": [NCLS, NCLS],
                         "finetune_mode": finetune, "save_freq": 1000})

    accs = []
    prev = raw
    for t in (1, 2):
        _, a = run(2 * t, t, prev, True, os.path.join(root, "grid%d" % t))
        accs.append(a)
        model, a = run(2 * t + 1, t, prev, False, os.path.join(root, "train%d" % t))
        accs.append(a)
        prev = os.path.join(root, "train%d" % t, "best_model.pth.tar")
        names = [n for n, _ in model.named_parameters()]
        out["param_names"] = np.array(names)
        if t == 1:
            for n, p in model.named_parameters():
                out["t1_p_" + n] = p.detach().numpy().copy()
        out["t%d_smax_lamb" % t] = np.array([float(model.smax), float(model.lamb)])
        for tt in range(t):
            gates = model.mask(torch.LongTensor([tt]), s=PARAM[0])
            for i, gmask in enumerate(gates):
                out["t%d_gate%d_task%d" % (t, i, tt)] = gmask.detach().numpy().reshape(-1).copy()
        mask_pre, mask_back = HATA.Appr.init_masks(t, model, PARAM[0])
        for i, mp in enumerate(mask_pre):
            out["t%d_maskpre%d" % (t, i)] = mp.detach().numpy().reshape(-1).copy()
        out["t%d_maskback_stats" % t] = np.array([[float(v.sum()), float(v.numel())] for _, v in sorted(mask_back.items())])
        out["t%d_maskback_names" % t] = np.array(sorted(mask_back))
    out["stage_acc"] = np.array(accs, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "G12_hat_trainer.npz"), **out)
    print("stage_acc", out["stage_acc"])
    print(os.path.getsize(os.path.join(HERE, "G12_hat_trainer.npz")) // 1024, "KiB")
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
