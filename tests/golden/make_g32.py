"""G32: EBLL's prestep — the autoencoder grid on the previous task (methods/method.py:835-908) — as DATA: the reference's
unchanged EBLL.prestep over a stand-in autoencoder trainer and six accuracy tables (best node first / second / last, a tie,
all nodes weak, rising): the nodes trained and with what, the directories kept, the path handed to phase 2, the grid
checkpoint; the same on the finished tree again, and for a run interrupted after two nodes and continued.
Dev container only:   python tests/golden/make_g32.py   ->  tests/golden/G32_ebll_autoencoder_grid.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

harness.install()
import g32_common as G  # noqa: E402

if __name__ == "__main__":
    import methods.method as RM

    def install(trainer):
        saved = RM.trainEBLL.fine_tune_Adam_Autoencoder
        RM.trainEBLL.fine_tune_Adam_Autoencoder = trainer
        return lambda: setattr(RM.trainEBLL, "fine_tune_Adam_Autoencoder", saved)

    data = {"tables": G.generate(lambda: RM.parse("EBLL"), install)}
    path = os.path.join(HERE, "G32_ebll_autoencoder_grid.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
