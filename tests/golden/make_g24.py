"""G24: phase 2 of the framework (framework/framework_train.py:76-166, HyperparameterFramework.stabilityDecay with the
reference's own Manager, framework/main.py:181-220) as DATA: 24 scenarios (1 / 2 / 3 hyper-parameters, a method with its
own decay operator; thresholds met at once, after some decays, never) over a stand-in method.  Recorded per scenario: the
hyper-parameters of every training call, the framework state afterwards, hyperparams.pth.tar (threshold, accuracy, state),
SUCCESS.FLAG, the files left in TASK_TRAINING and the model that survived; a second run on the finished tree; a run that
dies after k attempts and its continuation by fresh objects.
Dev container only:   python tests/golden/make_g24.py   ->  tests/golden/G24_stability_decay.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

harness.install()
import g24_common as G  # noqa: E402

if __name__ == "__main__":
    import framework.framework_train as FT
    import framework.main as FM
    data = {"tables": G.generate(FT.HyperparameterFramework, FM.Manager)}
    path = os.path.join(HERE, "G24_stability_decay.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
