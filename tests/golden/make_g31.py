"""G31: the epoch loop of the reference's rehearsal trainer (methods/rehearsal/train_rehearsal.py:57-199, GEM's observe and the
finetune-mode observe_FT) as DATA, over a scripted wrapper: learning rate per epoch, early stop, NaN-loss exit, the
best model written at the end, checkpoints and resume, save_models_mode off.
Dev container only:   python tests/golden/make_g31.py   ->  tests/golden/G31_rehearsal_loop.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

torch = harness.install()
import g31_common as G  # noqa: E402

if __name__ == "__main__":
    import methods.rehearsal.main_rehearsal as MR
    import methods.rehearsal.train_rehearsal as TR
    MR.eval_batch = lambda model, x, y, args: (torch.tensor(0.3), torch.tensor(model.hits(x)))     # main_rehearsal.py:19-36
    data = {"runs": G.generate(TR.train_model, with_paths=True)}
    path = os.path.join(HERE, "G31_rehearsal_loop.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
