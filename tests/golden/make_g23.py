"""G23: decisions of the reference's phase-1 LR grid (framework/lr_grid_train.py:9-160) as DATA.  The reference's
unchanged lr_grid_single_task runs over a stand-in method whose grid_train returns accuracies from a seeded table (ties and
all-zero tables included) and drops a marker file into the node directory; recorded per table and per save_models_mode:
best_lr, best_acc, the winning node directory, the node directories that survive the clean-up, the checkpointed
accuracy lists, and — for an interrupted run resumed from grid_checkpoint.pth — which nodes are trained again.
Dev container only:   python tests/golden/make_g23.py   ->  tests/golden/G23_lr_grid_decisions.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

harness.install()
import g23_common as G  # noqa: E402

if __name__ == "__main__":
    import framework.lr_grid_train as RG
    data = {"lrs": G.LRS, "modes": G.MODES, "tables": G.generate(RG.lr_grid_single_task)}
    path = os.path.join(HERE, "G23_lr_grid_decisions.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
