"""G27: the epoch loops of the reference's four train_model variants (Finetune/train_SGD.py:41-189, EWC/train_EWC.py:111-234,
MAS/train_MAS.py:208-335, SI/train_SI.py:152-283) of LwF's (LwF/main_LWF.py:100-250) of IMM's L2-transfer (IMM/train_L2transfer.py:124-240) and of EBLL's (EBLL/Finetune_SGD_EBLL.py:226-388) as DATA: a scripted network turns a table into the validation accuracy of
every epoch, an optimizer logs the learning rate it is stepped with; recorded per scenario (plateau, short rising run,
mixed, NaN loss, a run resumed from epoch.pth.tar, save_models_mode off) and variant: best accuracy returned, forward
calls made (= epochs run), learning rate per epoch, and the checkpoint / best-model files with what they hold.
Dev container only:   python tests/golden/make_g27.py   ->  tests/golden/G27_epoch_loops.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

torch = harness.install()
import g27_common as G  # noqa: E402


def reference_train(variant, model, optimizer, lr, loaders, sizes, num_epochs, exp_dir, resume, saving_freq, save_models_mode):
    import torch.nn as nn
    crit = nn.CrossEntropyLoss()
    if variant == "sgd":
        import methods.Finetune.train_SGD as T
        return T.train_model(model, crit, optimizer, lr, loaders, sizes, False, num_epochs, exp_dir, resume,
                             save_models_mode=save_models_mode, saving_freq=saving_freq)
    if variant == "lwf":
        import methods.LwF.main_LWF as T
        return T.train_model_lwf(model, G.ScriptedTeacher(), crit, optimizer, lr, loaders, sizes, False, num_epochs, exp_dir, resume,
                                 temperature=2, saving_freq=saving_freq, reg_lambda=1)
    if variant == "ebll":
        import methods.EBLL.Finetune_SGD_EBLL as T
        return T.train_model_ebll(model, G.ScriptedEbllTeacher(), crit, nn.MSELoss(), optimizer, lr, loaders, sizes, False, num_epochs,
                                  exp_dir, resume, temperature=2, reg_alpha=1e-6, saving_freq=saving_freq, reg_lambda=1)
    mod = {"ewc": "methods.EWC.train_EWC", "mas": "methods.MAS.train_MAS", "si": "methods.SI.train_SI",
           "imm": "methods.IMM.train_L2transfer"}[variant]
    T = __import__(mod, fromlist=["train_model"])
    return T.train_model(model, crit, optimizer, lr, loaders, sizes, False, num_epochs, exp_dir, resume, saving_freq)


if __name__ == "__main__":
    data = {"runs": G.generate(reference_train)}
    path = os.path.join(HERE, "G27_epoch_loops.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
