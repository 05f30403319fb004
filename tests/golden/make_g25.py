"""G25: the reference's evaluation driver and naming helpers as DATA.
  * framework/eval.py:146-247 (eval_all_models_all_tasks / eval_task_steps_accuracy) over a stand-in method: which
    (task, model) pairs are evaluated with which paths, and the result files written — for the whole sequence, a task
    window, an evaluation that fails on a later / on the first model of a task, a result file that already exists (with and
    without overwrite mode) and debug mode;
  * utilities/utils.py:get_exp_name and models/net.py:get_init_modelname for six argument sets (weight decay, DROP / BN model
    names, list-valued static hyper-parameters, no hyper-parameters).
Dev container only:   python tests/golden/make_g25.py   ->  tests/golden/G25_eval_and_names.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

harness.install()
import g25_common as G  # noqa: E402

if __name__ == "__main__":
    import framework.eval as FE
    import models.net as MN
    import utilities.utils as U
    import methods.method as RM
    data = {"names": G.names(U.get_exp_name, MN.get_init_modelname),
            "evals": G.evals(FE.eval_all_models_all_tasks, U.get_perf_output_filename),
            "adopt": G.adopt(RM.Finetune.grid_poststep)}
    path = os.path.join(HERE, "G25_eval_and_names.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
