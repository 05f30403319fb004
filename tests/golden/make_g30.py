"""G30: the session loops of the reference's PackNet Manager (methods/packnet/main.py:234-339, train and prune) as DATA: a
Manager created without its constructor, do_epoch / eval as table look-ups, the pruner a logger; recorded for five training
cases (plateau, mixed, accuracies below a given best, resumed counters, save off) and two prune cases (with / without the
post-prune retraining): learning rate per epoch, early stop, the order of evaluations / pruning / checks, every file
written with the fields it holds.
Dev container only:   python tests/golden/make_g30.py   ->  tests/golden/G30_packnet_loops.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

torch = harness.install()
import g30_common as G  # noqa: E402

if __name__ == "__main__":
    import methods.packnet.main as PM
    data = {"runs": G.generate(PM.Manager, lambda m: {}, lambda m: torch.optim.SGD(m.model.parameters(), lr=m.args.lr, momentum=0.9))}
    path = os.path.join(HERE, "G30_packnet_loops.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
