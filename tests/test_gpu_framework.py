"""End-to-end on the GPU: SI first-task bootstrap, then EWC / MAS / SI through the two-phase
framework on three tiny synthetic tasks, with --test.  Checks the drop-in wire format
(best_model.pth.tar with reg_params, hyperparams.pth.tar, SUCCESS.FLAG, seq_res/seq_forgetting)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("fw"))


def _dataset(root):
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    return SyntheticTaskSequence(os.path.join(root, "data"), task_count=3, classes_per_task=4, sizes=(160, 40, 40),
                                 hw=32, noise=0.4, name="tiny3")


def _friendly_base_model(root):
    """VGG's N(0, .01) classifier init needs tens of epochs before the loss moves; the test has 8.
    Pre-create the framework's base-model file with a kaiming classifier init (the driver reuses an
    existing file, like models/net.py:158-169)."""
    from clsurvey_amd import models
    torch.manual_seed(0)
    m = models.parse_model_name("small_VGG9_cl_128_128", (32, 32), 4)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    torch.save(m, os.path.join(root, "models", "small_VGG9_cl_128_128.pth.tar"))


COMMON = ["small_VGG9_cl_128_128", "--lr_grid", "1e-2,3e-3", "--num_epochs", "8", "--batch_size", "40",
          "--saving_freq", "100"]


def test_si_first_task_dump_then_methods(workdir):
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    ds = _dataset(workdir)
    _friendly_base_model(workdir)
    out = driver.main(COMMON + ["--method_name", "SI", "--results_root", workdir, "--runmode",
                                "first_task_basemodel_dump"], method=M.parse("SI"), dataset=ds)
    first = out["manager"].best_model_path
    assert os.path.exists(first)
    m = torch.load(first, weights_only=False)
    assert hasattr(m, "reg_params") and "lambda" in m.reg_params
    live = [p for p in m.parameters() if p in m.reg_params]
    assert len(live) == 18 and all(set(m.reg_params[p]) >= {"omega", "w", "init_val"} for p in live)

    for name in ("EWC", "MAS", "SI"):
        out = driver.main(COMMON + ["--method_name", name, "--results_root", workdir, "--test"],
                          method=M.parse(name), dataset=ds)
        res = out["results"]
        assert sorted(res) == [0, 1, 2]
        assert len(res[0]["seq_res"][0]) == 3 and len(res[0]["seq_forgetting"][0]) == 2
        assert len(res[2]["seq_res"][2]) == 1
        accs = [a for i in res for a in res[i]["seq_res"][i]]
        assert all(0.0 <= a <= 100.0 for a in accs)
        assert res[2]["seq_res"][2][0] > 30.0, "%s: task-3 accuracy should beat 25%% chance" % name
        for t in (2, 3):
            tdir = os.path.join(out["manager"].parent_exp_dir, "task_%d" % t, "TASK_TRAINING")
            assert os.path.exists(os.path.join(tdir, "SUCCESS.FLAG"))
            assert os.path.exists(os.path.join(tdir, "hyperparams.pth.tar"))
            mt = torch.load(os.path.join(tdir, "best_model.pth.tar"), weights_only=False)
            rp = mt.reg_params
            assert "__arena__" not in rp and "lambda" in rp
            om = [rp[p]["omega"] for p in mt.parameters() if p in rp]
            assert om and all(bool((o >= 0).all()) for o in om)
            if name != "SI" or t == 3:
                assert any(float(o.abs().max()) > 0 for o in om), "%s task %d: omega all zero" % (name, t)
        hf = out["frameworks"][1]
        assert len(hf.trace) >= 1 and hf.trace[-1][1] >= 0.0


def test_end_to_end_matches_reference_driver_g10(tmp_path, golden):
    """Same tiny 3-task sequence, same deterministic start weights, same CLI flags and seeds as the run of the
    reference's UNCHANGED framework/main.py recorded in tests/golden/G10 (make_g10.py): SI first-task dump, then
    EWC with --test.  The build's driver + HIP path must reproduce the per-LR grid accuracies, the phase-2
    state (attempts, lambda, threshold), seq_res / seq_forgetting and the Omega statistics."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from g10_weights import det_weights
    from clsurvey_amd import models
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import method as M
    g = golden("G10_framework_ewc")
    root = str(tmp_path)
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=3, classes_per_task=4, sizes=(160, 40, 40),
                               hw=32, noise=0.4, name="tiny3")
    m = models.parse_model_name("small_VGG9_cl_128_128", (32, 32), 4)
    with torch.no_grad():
        for p, w in zip(m.parameters(), det_weights()):
            p.copy_(torch.from_numpy(w))
    os.makedirs(os.path.join(root, "models"))
    torch.save(m, os.path.join(root, "models", "small_VGG9_cl_128_128.pth.tar"))
    G10 = COMMON + ["--drop_margin", "0.05"]
    out = driver.main(G10 + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                      method=M.parse("SI"), dataset=ds)
    one = 1.0 / 40 + 1e-9          # one validation sample
    trace = {lr: acc for lr, it, acc in out["manager"].grid_trace}
    for lr in (1e-2, 3e-3):
        assert abs(trace[lr] - float(g["si_t1_lr%g" % lr][0])) <= one, ("SI grid", lr, trace[lr], g["si_t1_lr%g" % lr])
    assert abs(out["frameworks"][0].trace[-1][1] - float(g["si_t1_val_acc"])) <= one
    assert os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(out["manager"].best_model_path)))) == \
        str(g["si_first_task_modelname"])

    ewc = M.parse("EWC")
    ewc.hyperparams["lambda"] = float(g["ewc_lambda0"])      # stable regime, see make_g10.py
    out = driver.main(G10 + ["--method_name", "EWC", "--results_root", root, "--test"], method=ewc, dataset=ds)
    assert out["args"].exp_name == str(g["ewc_exp_name"])
    ref_attempts = g["ewc_attempts"]
    mine = [(t, tr[0]["lambda"], tr[1]) for t, hf in zip((2, 3), out["frameworks"][1:]) for tr in hf.trace]
    assert len(mine) == len(ref_attempts), (mine, ref_attempts)
    for a, b in zip(mine, ref_attempts):
        assert a[0] == int(b[0]) and a[1] == float(b[1]) and abs(a[2] - float(b[2])) <= one, (mine, ref_attempts)
    for t, hf in zip((2, 3), out["frameworks"][1:]):
        assert abs(hf.trace[-1][1] - float(g["ewc_t%d_val_acc" % t])) <= one, (t, hf.trace)
        assert abs(hf.trace[-1][2] - float(g["ewc_t%d_threshold" % t])) <= one
        assert hf.attempts == int(g["ewc_t%d_attempts" % t]), (t, hf.trace)
        assert hf.hyperparams["lambda"] == float(g["ewc_t%d_lambda" % t])
        tdir = os.path.join(out["manager"].parent_exp_dir, "task_%d" % t)
        grid = torch.load(os.path.join(tdir, "FT_LR_GRIDSEARCH", "grid_checkpoint.pth"), weights_only=False)["processed_lrs"]
        for lr in (1e-2, 3e-3):
            assert abs(grid[lr]["acc"][0] - float(g["ewc_t%d_lr%g" % (t, lr)][0])) <= one, (t, lr, grid[lr])
        mt = torch.load(os.path.join(tdir, "TASK_TRAINING", "best_model.pth.tar"), weights_only=False)
        om = [mt.reg_params[p]["omega"].double() for p in mt.parameters() if p in mt.reg_params]
        ref = g["ewc_t%d_omega_stats" % t]
        assert len(om) == ref.shape[0]
        for o, r in zip(om, ref):
            st = np.array([float(o.sum()), float(o.max()), float(o.pow(2).sum().sqrt())])
            # Omega is the Fisher of a model that went through >30 SGD steps on each side: the trajectories
            # separate at fp32 round-off and the near-converged gradients are very sensitive to that (observed
            # 2-20 % on these statistics).  The Fisher arithmetic itself is pinned exactly by G2; here only
            # the order of magnitude is.
            assert np.all((st <= 2.0 * r + 1e-12) & (st >= 0.5 * r - 1e-12)), (t, st, r)
    res = out["results"]
    # Test accuracies: the just-trained task within three test samples.  Accuracies on OLDER tasks after further
    # training depend on which task-2 / task-3 model the >60 chaotic SGD steps per side end in (the build and the
    # CPU oracle started from the SAME model agree to one sample: test_ewc_task_training_matches_oracle below;
    # the reference's CPU run and this run do not start task 3 from the same weights), so they are only
    # range-checked here.
    three = 100.0 * 3 / 40 + 1e-9
    for i in range(3):
        got, ref = np.array(res[i]["seq_res"][i]), g["seq_res%d" % i]
        assert got.shape == ref.shape, (i, got, ref)
        assert abs(got[0] - ref[0]) <= three, (i, got, ref)
        assert np.all((got >= 0) & (got <= 100))
        assert np.array(res[i]["seq_forgetting"][i]).shape == g["seq_forgetting%d" % i].shape
    print("G10 seq_res build:", {i: res[i]["seq_res"][i] for i in range(3)}, " reference:",
          {i: list(g["seq_res%d" % i]) for i in range(3)})


def test_ewc_task_training_matches_oracle(tmp_path):
    """fine_tune_EWC_acuumelation (Fisher + head swap + 8 epochs of penalised SGD with the early-stop / LR
    schedule) on the GPU vs the same procedure spelled out with the CPU oracle, from the SAME start model, same
    seeds and loader order: Omega identical, |theta - theta*| per tensor within 10 %, accuracies within a sample."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from g10_weights import det_weights, SMALL
    from clsurvey_amd import models
    from clsurvey_amd.data import DeviceLoader
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import ewc as EW
    from oracle import regularizers_ref as R, vgg_ref
    root = str(tmp_path)
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=4, sizes=(160, 40, 40), hw=32,
                               noise=0.4, name="tiny2")
    d1 = torch.load(ds.get_task_dataset_path("1"), weights_only=False)
    d2 = torch.load(ds.get_task_dataset_path("2"), weights_only=False)
    m = models.parse_model_name("small_VGG9_cl_128_128", (32, 32), 4)
    with torch.no_grad():
        for p, w in zip(m.parameters(), det_weights()):
            p.copy_(torch.from_numpy(w))
    start = os.path.join(root, "start.pth.tar")
    torch.save(m, start)
    lam, lr = 40.0, 1e-2

    def test_acc(params, dset):
        with torch.no_grad():
            return float((vgg_ref.forward(params, SMALL, dset.x).argmax(1) == dset.y).float().mean())

    driver.set_random(0)
    _, acc_b = EW.fine_tune_EWC_acuumelation(d2, start, os.path.join(root, "build"), None, [d1], reg_lambda=lam,
                                             num_epochs=8, lr=lr, batch_size=40)
    best = torch.load(os.path.join(root, "build", "best_model.pth.tar"), weights_only=False)
    pb = [p.detach().cpu() for p in best.parameters()]
    rp = best.reg_params
    plist = list(best.parameters())

    driver.set_random(0)
    theta = [torch.from_numpy(w.copy()) for w in det_weights()]
    omega = R.diag_fisher(theta, SMALL, list(DeviceLoader(d1["train"], 40, False, "cpu")), len(d1["train"]))
    init = [t.clone() for t in theta]
    head = torch.nn.Linear(128, 4)
    theta[-2], theta[-1] = head.weight.detach().clone(), head.bias.detach().clone()
    omega[-2] = omega[-1] = init[-2] = init[-1] = None
    loaders = {x: DeviceLoader(d2[x], 40, True, "cpu") for x in ("train", "val")}
    bufs = [None] * len(theta)
    best_acc, best_theta, count, first = 0.0, None, 0, True
    for ep in range(8):
        if count > 10:
            break
        if count == 5:
            lr *= 0.1
        for x, y in loaders["train"]:
            _, _, gr, _ = vgg_ref.loss_and_grads(theta, SMALL, x, y, "ce_mean")
            nn_ = [R.reg_sgd_step(t, gi, o, iv, b, lam, lr, 0.9, 0.0, first) for t, gi, o, iv, b in zip(theta, gr, omega, init, bufs)]
            theta, bufs = [a[0] for a in nn_], [a[1] for a in nn_]
            first = False
        corr = 0
        for x, y in loaders["val"]:
            with torch.no_grad():
                corr += int((vgg_ref.forward(theta, SMALL, x).argmax(1) == y).sum())
        acc = corr / len(d2["val"])
        if acc > best_acc:
            best_acc, best_theta, count = acc, [t.clone() for t in theta], 0
        else:
            count += 1
    assert abs(acc_b - best_acc) <= 1.0 / 40 + 1e-9, (acc_b, best_acc)
    assert abs(test_acc(pb, d2["test"]) - test_acc(best_theta, d2["test"])) <= 2.0 / 40 + 1e-9
    for i, (p, o, iv, t) in enumerate(zip(plist, omega, init, best_theta)):
        if o is None:
            assert p not in rp
            continue
        ob = rp[p]["omega"].detach().cpu()
        assert float((ob - o).abs().max()) <= 1e-2 * float(o.abs().max()) + 1e-12, "omega %d" % i   # ReLU/pool decision flips, see test_engine_full_size_vs_oracle
        assert torch.equal(rp[p]["init_val"].detach().cpu(), iv), "init_val %d" % i
        db, do = float((pb[i] - iv).abs().max()), float((t - iv).abs().max())
        assert abs(db - do) <= 0.1 * do + 1e-6, ("drift", i, db, do)
