"""G29: the epoch loops of the reference's two HAT trainers (methods/HAT/approaches/hat.py:95-190 joint training with warm-up,
hat_finetune.py:39-106 the phase-1 search) as DATA: train_epoch / eval replaced by table look-ups, masks by nothing;
recorded for 36 scenarios (first / later task, warm-up on / off, plateau / mixed / rising accuracies, a run resumed from
epoch.pth.tar): learning rate and lambda of every epoch, epochs run, best accuracy, the checkpoint and the model kept.
run.py:107-109 constructs both with lr_factor = 2, lr_patience = 30.
Dev container only:   python tests/golden/make_g29.py   ->  tests/golden/G29_hat_trainer_loops.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

harness.install()
import g29_common as G  # noqa: E402

if __name__ == "__main__":
    import methods.HAT.approaches.hat as HJ
    import methods.HAT.approaches.hat_finetune as HF

    HJ.Appr.init_masks = staticmethod(lambda *a, **k: (None, {}))        # (hat.py:134 calls it on the class by name)

    def make_trainer(joint, model, exp_dir, nepochs, args):
        cls = HJ.Appr if joint else HF.Appr
        sub = type("Scripted" + cls.__name__, (cls,), {"init_masks": staticmethod(lambda *a, **k: (None, {})),
                                                       "get_ft_mask": lambda self, t: None})
        return sub(model, exp_dir, nepochs=nepochs, sbatch=args.batch_size, lr=args.lr, lr_factor=2, lr_patience=30, args=args)

    data = {"runs": G.generate(make_trainer)}
    path = os.path.join(HERE, "G29_hat_trainer_loops.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
