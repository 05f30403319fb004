"""G28: what the reference's method objects (methods/method.py) hand to their trainers, as DATA.  Every trainer entry point
(fine_tune_EWC_acuumelation, fine_tune_objective_based_acuumelation, fine_tune_elastic, fine_tune_SGD_LwF, fine_tune_SGD_EBLL,
fine_tune_l2transfer, fine_tune_SGD, packnet / HAT / rehearsal main) is replaced by a recorder, the hooks the framework
calls (grid_prestep, grid_train, grid_poststep, train_init, train, poststep) run for tasks 1 and 2 with one fixed
(args, manager) pair, and the bound arguments of every trainer call + the args / manager fields afterwards are stored.
Dev container only:   python tests/golden/make_g28.py   ->  tests/golden/G28_trainer_calls.json"""
import contextlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, HERE)
import harness  # noqa: E402

harness.install()
import g28_common as G  # noqa: E402

if __name__ == "__main__":
    import framework.main as FM
    import methods.method as RM

    TARGETS = [(RM.trainEWC, "fine_tune_EWC_acuumelation", (None, 0.5)), (RM.trainMAS, "fine_tune_objective_based_acuumelation", (None, 0.5)),
               (RM.trainSI, "fine_tune_elastic", (None, 0.5)), (RM.trainLWF, "fine_tune_SGD_LwF", (None, 0.5)),
               (RM.trainLWF, "fine_tune_freeze", None), (RM.trainEBLL, "fine_tune_SGD_EBLL", (None, 0.5)),
               (RM.trainEBLL, "fine_tune_Adam_Autoencoder", (None, 0.5)), (RM.trainIMM, "fine_tune_l2transfer", (None, 0.5)),
               (RM.trainFT, "fine_tune_SGD", (None, 0.5))]
    MAINS = [(RM.trainPacknet, "packnet.main", 0.5), (RM.trainHAT, "hat.main", (None, 0.5)), (RM.trainRehearsal, "rehearsal.main", (None, 0.5))]

    @contextlib.contextmanager
    def patches(log):
        saved = []
        for mod, fn, res in TARGETS:
            saved.append((mod, fn, getattr(mod, fn)))
            setattr(mod, fn, G.Recorder(log, fn, getattr(mod, fn), res))
        for mod, label, res in MAINS:
            saved.append((mod, "main", mod.main))
            mod.main = G.Recorder(log, label, None, res)
        compose = RM.Finetune.compose_dataset
        RM.Finetune.compose_dataset = staticmethod(G.Recorder(log, "compose_dataset", None, ("<loaders>", "<sizes>", "<classes>")))
        try:
            yield
        finally:
            for mod, fn, orig in saved:
                setattr(mod, fn, orig)
            RM.Finetune.compose_dataset = staticmethod(compose)

    exists = {G.ROOT + "/parent/task_2/TASK_TRAINING/best_model_PRUNED_final.pth.tar"}
    data = {"hooks": G.run(RM.parse, FM.Manager, patches, exists)}
    path = os.path.join(HERE, "G28_trainer_calls.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")
