"""G26: the reference's per-task orchestration (framework/framework_train.py:219-292, framework_single_task with the
reference's own Manager and phases) as DATA: 24 scenarios (hook sets x first task with / without training / wrapping, a
later task, PackNet's storage policy, --save_models_FT_heuristic) over a stand-in method that logs every hook call and the
args / manager fields visible at that moment.
Dev container only:   python tests/golden/make_g26.py   ->  tests/golden/G26_single_task_trace.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

harness.install()
import g26_common as G  # noqa: E402

if __name__ == "__main__":
    import framework.framework_train as FT
    import framework.main as FM
    data = {"tables": G.generate(FT.framework_single_task, FM.Manager)}
    path = os.path.join(HERE, "G26_single_task_trace.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
