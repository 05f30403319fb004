"""Row s of the scope table (the sweep's observables): a 3-task EWC sweep through the driver with the reference's defaults, then
every task's accepted training repeated on the three fp32-grade kernel paths (CLHIP_BS = 0 / 1 / 2) from the free run's own
previous model, learning rate and lambda (bench.forced_paths; framework_train.py:76-144, EWC/train_EWC.py:204-205, method.py:668).

Asserted: per task whose training sits clear of the stability limit (x <= 3.0) the paths' accuracies (new task, previous task under
the new trunk, validation) within 1 point; Sum(Omega) within 1e-3 relative on every task; and the stability-decay decisions of the free run on the side of the heavy-ball limit 2 (1 + 0.9) that their outcome
says (no attempt below the limit rejected, every rejected attempt above it)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_three_kernel_paths_agree_task_by_task_on_a_sweep():
    """Three tasks at the sweep's own size (8000 / 2000 / 1000 images: at reduced sizes Omega = sum_b (sum_i g_i)^2 / data_len is several
    times larger and every accepted training sits above the stability limit — measured, profiles/r06_k_reduced_sweeps.txt)."""
    assert torch.cuda.is_available()
    import bench
    res = bench.full_sweep(0, 0, tasks=3, sizes=(8000, 2000, 1000), epochs=70)
    assert "gpu_error" not in res, res.get("gpu_error")
    cond = res["conditioning"]
    assert cond["below_limit_rejected"] == 0, cond            # x < 3.8 => the penalised training is stable and meets the threshold
    assert cond["rejected_all_above_limit"], cond             # a rejected attempt diverged: only possible above the limit
    fp = res["forced_paths"]
    assert "error" not in fp, fp
    assert len(fp["per_task"]) == 2
    well = [e for e in fp["per_task"] if e["x"] <= bench.NEAR_LIMIT]
    assert well, "no training of this sweep sits clear of the stability limit: %s" % fp["per_task"]
    for e in fp["per_task"]:
        # the Fisher pass starts from the same model on every path: always a well-conditioned comparison
        assert e["omega_sum_rel_spread"] <= 1e-3, e
    for e in well:
        assert not e["diverged"], e
        assert e["gap_points"]["test_acc"] <= 1.0, e
        assert e["gap_points"]["previous_task_test_acc"] <= 1.0, e
        assert e["gap_points"]["val_acc"] <= 1.0, e
