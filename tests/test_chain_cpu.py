"""CPU checks of bench.py's CPU-vs-HIP chain (row s of the scope table): the oracle leg must be able to start a task's EWC training
from a model file as the HIP path (or the reference itself) writes it — a pickled nn.Module whose `reg_params[param]["omega"]` holds
the importance accumulated so far (EWC/main_EWC.py:160-232) — and report what bench.chain_collect compares.

No HIP code runs here: the model file is fabricated on the host with the product's model factory and a known omega."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _model_file(path, with_reg_params):
    from clsurvey_amd import models
    torch.manual_seed(3)
    model = models.parse_model_name("small_VGG9_cl_128_128", (64, 64), 20)
    omegas = None
    if with_reg_params:
        params = list(model.parameters())
        omegas = [torch.rand_like(p) * 1e-3 for p in params[:-2]]
        # the head of the last training has no entry; "lambda" sits in the same dict (main_EWC.py:43)
        model.reg_params = {p: {"omega": o, "init_val": p.data.clone()} for p, o in zip(params[:-2], omegas)}
        model.reg_params["lambda"] = 400.0
    torch.save(model, path)
    return omegas


def test_oracle_reads_omega_from_reg_params(tmp_path):
    from oracle import sweep_ref
    path = str(tmp_path / "m.pth.tar")
    want = _model_file(path, True)
    model = torch.load(path, map_location="cpu", weights_only=False)
    got = sweep_ref._omega_of_reg_params(model)
    assert len(got) == len(list(model.parameters()))
    assert got[-1] is None and got[-2] is None
    for g, w in zip(got[:-2], want):
        assert torch.equal(g, w)
    _model_file(path, False)
    assert sweep_ref._omega_of_reg_params(torch.load(path, map_location="cpu", weights_only=False)) is None


@pytest.mark.timeout(600)
def test_chain_cpu_leg_accumulates_onto_the_model_files_omega(tmp_path):
    """One job through bench.chain_cpu_leg on a small task: the omega the training was penalised with = the file's omega + the
    Fisher diagonal of the previous task under the file's parameters (main_EWC.py:138-157, :205-232), on the trunk."""
    import bench
    from clsurvey_amd.data import synthetic_task
    from oracle import regularizers_ref as R
    from oracle import sweep_ref, vgg_ref
    for t in (1, 2):
        torch.save(synthetic_task(60, 80, 20, 20, hw=64, seed=7000 + t, noise=0.5, kind="blobs", blobs=bench.SWEEP_DATA["blobs"]),
                   str(tmp_path / ("task_%d.pth.tar" % t)))
    mpath = str(tmp_path / "prev.pth.tar")
    older = _model_file(mpath, True)
    plan = {"jobs": [{"task": 2, "dataset": str(tmp_path / "task_2.pth.tar"), "previous_dataset": str(tmp_path / "task_1.pth.tar"),
                      "previous_model": mpath, "lr": 1e-3, "lambda": 100.0}], "epochs": 2, "batch": 20}
    ppath = str(tmp_path / "plan.json")
    with open(ppath, "w") as f:
        json.dump(plan, f)
    threads = torch.get_num_threads()
    try:
        row = bench.chain_cpu_leg(ppath, 0, 2)           # (a leg is a process of its own in the bench: it sets its thread count)
    finally:
        torch.set_num_threads(threads)
    assert not row.get("diverged"), row
    for k in ("val_acc", "test_acc", "previous_task_test_acc", "omega_sum_trunk", "omega_max", "seconds"):
        assert k in row, row
    assert 0.0 <= row["test_acc"] <= 100.0 and 0.0 <= row["previous_task_test_acc"] <= 100.0
    # the same sum by hand
    model = torch.load(mpath, map_location="cpu", weights_only=False)
    theta = [p.detach().clone() for p in model.parameters()]
    prev = torch.load(plan["jobs"][0]["previous_dataset"], weights_only=False)["train"]
    torch.manual_seed(0)
    fisher = R.diag_fisher(theta, vgg_ref.CFGS["small_VGG9"], list(sweep_ref._batches(prev, 20, False)), len(prev))
    want = sum(float((o + f).double().sum()) for o, f in zip(older, fisher[:-2]))
    assert abs(row["omega_sum_trunk"] - want) <= 1e-6 * abs(want), (row["omega_sum_trunk"], want)
