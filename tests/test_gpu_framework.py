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
    one = 2.0 / 40 + 1e-9          # two validation samples: best-of-epochs accuracies of fp32 trajectories that
    #                                separate at round-off (summation order differs from torch-CPU in every kernel)
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
            assert np.all((st <= 3.0 * r + 1e-12) & (st >= r / 3.0 - 1e-12)), (t, st, r)
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
    # best-of-8-epochs validation accuracy of two fp32 trajectories that separate at round-off (different summation
    # order in the first-layer kernel): observed 0-2 of 40 validation samples
    assert abs(acc_b - best_acc) <= 2.0 / 40 + 1e-9, (acc_b, best_acc)
    assert abs(test_acc(pb, d2["test"]) - test_acc(best_theta, d2["test"])) <= 3.0 / 40 + 1e-6
    for i, (p, o, iv, t) in enumerate(zip(plist, omega, init, best_theta)):
        if o is None:
            assert p not in rp
            continue
        ob = rp[p]["omega"].detach().cpu()
        assert float((ob - o).abs().max()) <= 1e-2 * float(o.abs().max()) + 1e-12, "omega %d" % i   # ReLU/pool decision flips, see test_engine_full_size_vs_oracle
        assert torch.equal(rp[p]["init_val"].detach().cpu(), iv), "init_val %d" % i
        db, do = float((pb[i] - iv).abs().max()), float((t - iv).abs().max())
        # the two runs may pick a different "best" epoch once their validation accuracies differ by a sample, so the
        # distance moved from theta* is only comparable in order of magnitude (the per-step arithmetic is pinned by G5)
        assert 0.4 * do - 1e-6 <= db <= 2.5 * do + 1e-6, ("drift", i, db, do)


# --------------------------------------------------------------------------- PackNet trainer (a17/a18)
G11_CFG = [32, "M", 32, "M", 32, 32, "M", 64, 64, "M"]


def _g11_setup(root):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from g10_weights import det_weights
    from clsurvey_amd import models
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=4, sizes=(160, 40, 40),
                               hw=32, noise=0.4, name="tiny2")
    paths = [ds.get_task_dataset_path(task_name=str(t)) for t in (1, 2)]
    m = models.VGGSlim(cfg=G11_CFG, num_classes=4, classifier_inputdim=64 * 2 * 2, classifier_dim1=64,
                       classifier_dim2=64)
    with torch.no_grad():
        for p, q in zip(m.parameters(), det_weights(11, G11_CFG, (64, 64), 4, 32)):
            p.copy_(torch.from_numpy(q))
    raw = os.path.join(root, "raw.pth.tar")
    torch.save(m, raw)
    return paths, raw


def test_packnet_trainer_g11(tmp_path, golden):
    """packnet_main.main driven with the overwrite_args sequence of methods/method.py:PackNet vs the run of the
    reference's unchanged packnet/main.py recorded in G11 (make_g11.py): wrapper layout, init-dump masks, task-1
    finetune (weights after 8 epochs), prune (mask counts exact, masks/weights vs the reference's from the SAME
    inputs in test_packnet_prune_stage), stage accuracies, and PackNet's invariants (old-task weights frozen
    bit-exactly, zero forgetting)."""
    import numpy as np
    import torch.nn as nn
    from clsurvey_amd.methods import packnet_main as PM
    g = golden("G11_packnet_trainer")
    ft_epochs, post_epochs, batch, lr, perc, wd = [float(v) for v in g["hyper"]]
    ft_epochs, post_epochs, batch = int(ft_epochs), int(post_epochs), int(batch)
    root = str(tmp_path)
    paths, raw = _g11_setup(root)
    init = os.path.join(root, "INIT_WRAPPED.pth")
    PM.main({"arch": "VGGslim_nopretrain", "init_dump": True, "cuda": True, "loadname": raw, "save_prefix": init,
             "last_layer_idx": 4, "current_dataset_idx": 1})
    ck = torch.load(init, weights_only=False)
    layout = [[i, int(isinstance(mod, nn.Conv2d)), mod.weight.numel()]
              for i, mod in enumerate(ck["model"].shared.modules()) if isinstance(mod, (nn.Conv2d, nn.Linear))]
    assert layout == g["init_layout"].tolist()
    assert sorted(ck["previous_masks"]) == [r[0] for r in layout]
    assert all(int(v.sum()) == 0 and v.dtype == torch.uint8 for v in ck["previous_masks"].values())
    assert ck["dataset2idx"] == {"nopretrain": 1}

    one = 100.0 / 40 + 1e-9
    prev, accs, finals = init, [], []
    for t in (1, 2):
        ft = os.path.join(root, "ft%d" % t, "best_model")
        os.makedirs(os.path.dirname(ft))
        torch.manual_seed(100 + 2 * t)
        accs.append(PM.main({
            "weight_decay": wd, "disable_pruning_mask": t == 1, "train_path": paths[t - 1], "test_path": paths[t - 1],
            "mode": "finetune", "dataset": "survey_TASK_%d" % t, "num_outputs": 4, "loadname": prev, "lr": lr,
            "finetune_epochs": ft_epochs, "cuda": True, "save_prefix": ft, "batch_size": batch, "train_bn": False,
            "saving_freq": 100, "current_dataset_idx": t}))
        ckf = torch.load(ft + ".pth.tar", weights_only=False)
        assert list(ckf["model"].datasets) == [str(s) for s in g["ft%d_datasets" % t]]
        if t == 1:
            # same start weights, same batch order, 8 epochs x 4 batches of momentum SGD: weights agree to
            # fp32 trajectory noise (ReLU / max-pool flips), head included
            for i, mod in enumerate(ckf["model"].shared.modules()):
                if isinstance(mod, (nn.Conv2d, nn.Linear)):
                    ref = torch.from_numpy(g["ft1_w%d" % i])
                    rel = float((mod.weight.detach().cpu() - ref).norm() / ref.norm())
                    assert rel < 2e-2, ("ft1 weight", i, rel)
            assert abs(float(ckf["accuracy"]) - float(g["ft1_ckpt_acc"])) <= 2 * one
        pr = os.path.join(root, "pr%d" % t, "best_model_PRUNED")
        os.makedirs(os.path.dirname(pr))
        torch.manual_seed(101 + 2 * t)
        accs.append(PM.main({
            "weight_decay": wd, "train_path": paths[t - 1], "test_path": paths[t - 1], "mode": "prune",
            "dataset": "survey_TASK_%d" % t, "loadname": ft + ".pth.tar", "post_prune_epochs": post_epochs,
            "prune_perc_per_layer": perc, "lr": lr * 0.1, "finetune_epochs": ft_epochs, "cuda": True,
            "save_prefix": pr, "train_bn": False, "saving_freq": 100, "current_dataset_idx": t, "batch_size": batch}))
        pp = torch.load(pr + "_postprune.pth.tar", weights_only=False)
        for i, _, numel in layout:
            mine = pp["previous_masks"][i].cpu().numpy()
            ref = g["pp%d_m%d" % (t, i)]
            for owner in range(0, t + 1):
                a, b = int((mine == owner).sum()), int((ref == owner).sum())
                # k = round(perc * n) is data independent; from task 2 on, free weights of dead units are still
                # exactly 0 and tie below the cutoff (prune.py:43 uses <=), and WHICH units died is trajectory noise
                assert a == b if (t == 1 or owner == 1) else abs(a - b) <= 0.05 * numel, (t, i, owner, a, b)
        prev = pr + "_final.pth.tar" if os.path.exists(pr + "_final.pth.tar") else pr + "_postprune.pth.tar"
        finals.append(prev)
    ref_acc = g["stage_acc"]
    for a, b in zip(accs, ref_acc):
        assert abs(a - float(b)) <= 3 * one, (accs, ref_acc)

    # PackNet invariants across the two tasks
    c1 = torch.load(finals[0], weights_only=False)
    c2 = torch.load(finals[1], weights_only=False)
    mods1 = dict(enumerate(c1["model"].shared.modules()))
    mods2 = dict(enumerate(c2["model"].shared.modules()))
    for i, _, _ in layout:
        own1 = c2["previous_masks"][i].cpu() == 1
        assert torch.equal(c1["previous_masks"][i].cpu() == 1, own1), "task-1 ownership changed"
        assert torch.equal(mods1[i].weight.detach().cpu()[own1], mods2[i].weight.detach().cpu()[own1]), \
            "task-1 weights moved while training task 2"
        assert torch.equal(mods1[i].bias.detach().cpu(), mods2[i].bias.detach().cpu()), "shared biases moved"
        free = c2["previous_masks"][i].cpu() == 0
        assert float(mods2[i].weight.detach().cpu()[free].abs().max()) == 0.0
    ev = {}
    for name, path in (("after1", finals[0]), ("after2", finals[1])):
        torch.manual_seed(201)
        ev[name] = PM.main({"train_path": paths[0], "test_path": paths[0], "mode": "eval", "dataset": "survey_TASK_1",
                            "loadname": path, "cuda": True, "batch_size": batch, "current_dataset_idx": 1})
    assert ev["after1"] == ev["after2"], "PackNet must not forget task 1: %r" % (ev,)
    torch.manual_seed(202)
    e2 = PM.main({"train_path": paths[1], "test_path": paths[1], "mode": "eval", "dataset": "survey_TASK_2",
                  "loadname": finals[1], "cuda": True, "batch_size": batch, "current_dataset_idx": 2})
    assert abs(ev["after2"] - float(g["eval_acc"][0])) <= 3 * one and abs(e2 - float(g["eval_acc"][1])) <= 3 * one


def test_packnet_prune_stage_from_reference_checkpoint(tmp_path, golden):
    """Prune stage in isolation: start from the reference's task-1 finetuned weights (G11 ft1_*), run mode='prune'
    with no post-prune epochs: masks and surviving weights must equal the reference's _postprune checkpoint
    bit for bit, and the post-prune accuracy within one validation sample."""
    import torch.nn as nn
    from clsurvey_amd.methods import packnet_main as PM
    g = golden("G11_packnet_trainer")
    _, _, batch, lr, perc, wd = [float(v) for v in g["hyper"]]
    root = str(tmp_path)
    paths, raw = _g11_setup(root)
    init = os.path.join(root, "INIT_WRAPPED.pth")
    PM.main({"arch": "VGGslim_nopretrain", "init_dump": True, "cuda": True, "loadname": raw, "save_prefix": init,
             "last_layer_idx": 4, "current_dataset_idx": 1})
    ck = torch.load(init, weights_only=False)
    model = ck["model"]
    model.add_dataset("survey_TASK_1", 4)
    with torch.no_grad():
        for i, mod in enumerate(model.shared.modules()):
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                mod.weight.copy_(torch.from_numpy(g["ft1_w%d" % i]))
                mod.bias.copy_(torch.from_numpy(g["ft1_b%d" % i]))
                ck["previous_masks"][i].fill_(1)           # what make_finetuning_mask left in the ft1 checkpoint
        model.classifiers[0].weight.copy_(torch.from_numpy(g["ft1_hw0"]))
        model.classifiers[0].bias.copy_(torch.from_numpy(g["ft1_hb0"]))
    ft = os.path.join(root, "ft1.pth.tar")
    torch.save({"model": model, "previous_masks": ck["previous_masks"], "dataset2idx": {"nopretrain": 1}}, ft)
    pr = os.path.join(root, "PRUNED")
    torch.manual_seed(103)
    acc = PM.main({"weight_decay": wd, "train_path": paths[0], "test_path": paths[0], "mode": "prune",
                   "dataset": "survey_TASK_1", "loadname": ft, "post_prune_epochs": 0, "prune_perc_per_layer": perc,
                   "lr": lr * 0.1, "cuda": True, "save_prefix": pr, "train_bn": False, "saving_freq": 100,
                   "current_dataset_idx": 1, "batch_size": int(batch)})
    pp = torch.load(pr + "_postprune.pth.tar", weights_only=False)
    for i, mod in enumerate(pp["model"].shared.modules()):
        if isinstance(mod, (nn.Conv2d, nn.Linear)):
            assert torch.equal(pp["previous_masks"][i].cpu(), torch.from_numpy(g["pp1_m%d" % i])), ("mask", i)
            assert torch.equal(mod.weight.detach().cpu(), torch.from_numpy(g["pp1_w%d" % i])), ("weight", i)
    assert abs(acc - float(g["pp1_ckpt_acc"])) <= 100.0 / 40 + 1e-9


def test_packnet_through_driver(tmp_path):
    """PackNet Method class through the build's two-phase driver on three tiny tasks with --test: wire files,
    and zero forgetting in seq_forgetting (frozen old-task weights)."""
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    root = str(tmp_path)
    ds = _dataset(root)
    _friendly_base_model(root)
    pk = M.parse("packnet")
    pk.grid_batch_size = 40                 # 160-image tasks: 4 steps per epoch instead of 1
    pk.hyperparams["prune_perc_per_layer"] = 0.5
    out = driver.main(COMMON + ["--method_name", "packnet", "--results_root", root, "--test"], method=pk, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1, 2]
    for i in res:
        assert all(abs(f) < 1e-9 for f in res[i]["seq_forgetting"][i]), res[i]["seq_forgetting"]
    assert res[0]["seq_res"][0][0] > 40.0 and res[2]["seq_res"][2][0] > 30.0, res
    for t in (1, 2, 3):
        tdir = os.path.join(out["manager"].parent_exp_dir, "task_%d" % t, "TASK_TRAINING")
        assert os.path.exists(os.path.join(tdir, "best_model_PRUNED_postprune.pth.tar"))
        assert os.path.exists(os.path.join(tdir, "SUCCESS.FLAG"))
    ck = torch.load(out["model_paths"][-1], weights_only=False)
    assert len(ck["model"].classifiers) == 3
    owners = torch.cat([m.view(-1) for m in ck["previous_masks"].values()])
    assert set(owners.unique().tolist()) <= {0, 1, 2, 3} and int((owners == 3).sum()) > 0


# --------------------------------------------------------------------------- HAT trainer (a19-a21)
def _g12_setup(root):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from g10_weights import det_weights
    from clsurvey_amd import models
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=4, sizes=(160, 40, 40),
                               hw=32, noise=0.4, name="tiny2")
    paths = [ds.get_task_dataset_path(task_name=str(t)) for t in (1, 2)]
    m = models.VGGSlim(cfg=G11_CFG, num_classes=4, classifier_inputdim=64 * 2 * 2, classifier_dim1=64,
                       classifier_dim2=64)
    with torch.no_grad():
        for p, q in zip(m.parameters(), det_weights(12, G11_CFG, (64, 64), 4, 32)):
            p.copy_(torch.from_numpy(q))
    raw = os.path.join(root, "raw.pth.tar")
    torch.save(m, raw)
    return ds, paths, raw


def test_hat_trainer_g12(tmp_path, golden):
    """hat_main.main driven with the overwrite_args of methods/method.py:_modular_accespoint vs the run of the
    reference's unchanged HAT/run.py recorded in G12 (make_g12.py): phase-1 (all gates open) and joint-training
    accuracies on task 1 from the same start weights and seeds, then task 2 from the build's own task-1 model;
    HAT's invariant: parameters whose back-mask is exactly 0 do not move at all while task 2 trains."""
    from clsurvey_amd.methods import hat_main as HM
    from clsurvey_amd.methods import hat as H
    g = golden("G12_hat_trainer")
    nepochs, batch, lr, wd, smax, c = [float(v) for v in g["hyper"]]
    root = str(tmp_path)
    ds, paths, raw = _g12_setup(root)

    def run(stage, t, prev, finetune, outdir):
        os.makedirs(outdir, exist_ok=True)
        torch.manual_seed(300 + stage)
        return HM.main({"weight_decay": wd, "task_name": str(t), "task_count": t, "prev_model_path": prev,
                        "model_name": "tiny_VGG9_cl_64_64", "output": outdir, "nepochs": int(nepochs),
                        "parameter": [smax, c], "cuda": True, "dataset_path": paths[t - 1], "dataset": ds, "n_tasks": 2,
                        "batch_size": int(batch), "lr": lr, "is_scratch_model": t == 1, "approach": "hat",
                        "nc_per_task": [4, 4], "finetune_mode": finetune, "save_freq": 1000})
    one = 1.0 / 40 + 1e-9
    ref = g["stage_acc"]
    _, a = run(2, 1, raw, True, os.path.join(root, "grid1"))
    assert abs(a - float(ref[0])) <= 2 * one, ("task-1 phase-1 acc", a, ref[0])
    m1, a = run(3, 1, raw, False, os.path.join(root, "train1"))
    assert abs(a - float(ref[1])) <= 4 * one, ("task-1 joint acc", a, ref[1])
    assert [n for n, _ in m1.named_parameters()] == [str(n) for n in g["param_names"]]
    assert m1.smax == smax and m1.lamb == c
    prev = os.path.join(root, "train1", "best_model.pth.tar")
    _, a = run(4, 2, prev, True, os.path.join(root, "grid2"))
    assert a >= float(ref[2]) - 0.25, ("task-2 phase-1 acc", a, ref[2])
    m1 = torch.load(prev, weights_only=False)
    before = {n: p.detach().clone() for n, p in m1.named_parameters()}
    hat = H.HatEngine(m1, int(batch), (3, 32, 32), "cuda")
    _, mask_back = H.init_masks(hat, 1, smax)
    m2, a = run(5, 2, prev, False, os.path.join(root, "train2"))
    assert a >= 0.3, ("task-2 joint acc", a, ref[3])
    moved_somewhere = False
    for n, p in m2.named_parameters():
        if n in mask_back:
            frozen = (mask_back[n] == 0).cpu()
            assert torch.equal(p.detach().cpu()[frozen], before[n].cpu()[frozen]), "hard-masked entries of %s moved" % n
            moved_somewhere |= bool((p.detach().cpu()[~frozen] != before[n].cpu()[~frozen]).any())
    assert moved_somewhere
    e0 = dict(m2.named_parameters())["conv_embs.0.weight"].detach().cpu()
    assert torch.equal(e0[0], before["conv_embs.0.weight"].cpu()[0]), "task-1 embedding row changed during task 2"
    assert float(e0.abs().max()) <= 6.0


def test_hat_masks_from_reference_model_g12(golden):
    """Gates, cumulative mask and back-mask of the REFERENCE's trained task-1 model (G12 t1_p_*), computed by the
    build's kernels (clhip_hat_gate / clhip_hat_backmask) vs the reference's Net.mask / Appr.init_masks."""
    import numpy as np
    from clsurvey_amd import models
    from clsurvey_amd.methods import hat as H
    g = golden("G12_hat_trainer")
    smax = float(g["hyper"][4])
    raw = models.VGGSlim(cfg=G11_CFG, num_classes=4, classifier_inputdim=64 * 2 * 2, classifier_dim1=64,
                         classifier_dim2=64)
    net = H.HatNet(raw, (3, 32, 32), [(0, 4), (1, 4)])
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(torch.from_numpy(g["t1_p_" + n]))
    hat = H.HatEngine(net, 8, (3, 32, 32), "cuda")
    for i, gt in enumerate(hat.masks_at(0, smax)):
        np.testing.assert_allclose(gt.cpu().numpy(), g["t1_gate%d_task0" % i], rtol=1e-5, atol=1e-6)
    mask_pre, mask_back = H.init_masks(hat, 1, smax)
    for i, mp in enumerate(mask_pre):
        np.testing.assert_allclose(mp.cpu().numpy(), g["t1_maskpre%d" % i], rtol=1e-5, atol=1e-6)
    names = [str(n) for n in g["t1_maskback_names"]]
    assert sorted(mask_back) == names
    for n, (s, numel) in zip(names, g["t1_maskback_stats"]):
        assert mask_back[n].numel() == int(numel)
        assert abs(float(mask_back[n].double().sum()) - float(s)) <= 1e-4 * max(1.0, float(numel))


def test_hat_through_driver(tmp_path):
    """HAT Method class through the two-phase driver on three tiny tasks with --test."""
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    root = str(tmp_path)
    ds = _dataset(root)
    _friendly_base_model(root)
    hat = M.parse("HAT")
    hat.hyperparams["smax"], hat.hyperparams["c"] = 50.0, 0.75
    out = driver.main(COMMON + ["--method_name", "HAT", "--results_root", root, "--test"], method=hat, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1, 2]
    assert res[0]["seq_res"][0][0] > 40.0, res
    accs = [a for i in res for a in res[i]["seq_res"][i]]
    assert all(0.0 <= a <= 100.0 for a in accs)
    mt = torch.load(out["model_paths"][-1], weights_only=False)
    assert mt.smax is not None and len(mt.conv_embs) == 6 and mt.conv_embs[0].weight.shape[0] == 3
    for t in (1, 2, 3):
        tdir = os.path.join(out["manager"].parent_exp_dir, "task_%d" % t, "TASK_TRAINING")
        assert os.path.exists(os.path.join(tdir, "best_model.pth.tar")) and os.path.exists(os.path.join(tdir, "SUCCESS.FLAG"))


# --------------------------------------------------------------------------- GEM trainer (a13-a16)
def test_gem_through_driver(tmp_path):
    """GEM Method class through the driver on three tiny tasks with --test: task 1 = the shared SI model wrapped
    with its exemplars (postprocess), tasks 2-3 = phase-1 observe_FT grid + phase-2 observe with memory passes,
    Gram, QP and projection.  Checks the wire files, the exemplar buffers and that the pickled wrapper reloads."""
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    root = str(tmp_path)
    ds = _dataset(root)
    _friendly_base_model(root)
    driver.main(COMMON + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    gem = M.parse("GEM")
    gem.static_hyperparams = {"mem_per_task": 64}
    out = driver.main(COMMON + ["--method_name", "GEM", "--results_root", root, "--test"], method=gem, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1, 2]
    accs = [a for i in res for a in res[i]["seq_res"][i]]
    assert all(0.0 <= a <= 100.0 for a in accs)
    assert res[0]["seq_res"][0][0] > 40.0 and res[2]["seq_res"][2][0] > 30.0, res
    wrapped = torch.load(out["model_paths"][0], weights_only=False)
    assert wrapped.observed_tasks == [0] and wrapped.n_memories == 64 and wrapped.n_outputs == 12
    assert float(wrapped.memory_x[0].abs().sum()) > 0 and float(wrapped.memory_x[1].abs().sum()) == 0
    last = torch.load(out["model_paths"][-1], weights_only=False)
    assert last.observed_tasks == [0, 1, 2] and last.cum_nc_per_task == [4, 8, 12]
    assert all(float(last.memory_x[t].abs().sum()) > 0 for t in range(3))
    x = torch.randn(5, 3, 32, 32, device="cuda")
    lo = last(x, 1)
    assert lo.shape == (5, 12) and bool((lo[:, :4] < -1e10).all()) and bool((lo[:, 8:] < -1e10).all())
    hf = out["frameworks"][-1]
    assert len(hf.trace) >= 1


# --------------------------------------------------------------------------- IMM
@pytest.mark.parametrize("mode", ["mean", "mode"])
def test_imm_through_driver(tmp_path, mode):
    """IMM (no_framework: L2-transfer LR grid per task, merge before evaluation) through the driver with --test."""
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    root = str(tmp_path)
    ds = _dataset(root)
    _friendly_base_model(root)
    driver.main(COMMON + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    imm = M.parse("IMM_" + mode)
    assert imm.eval_name == "IMM_" + mode
    out = driver.main(COMMON + ["--method_name", "IMM", "--results_root", root, "--test"], method=imm, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1, 2]
    accs = [a for i in res for a in res[i]["seq_res"][i]]
    assert all(0.0 <= a <= 100.0 for a in accs) and res[0]["seq_res"][0][0] > 40.0
    tdir = os.path.dirname(out["model_paths"][-1])
    merged = os.path.join(tdir, "best_model_%s_merge.pth.tar" % mode)
    assert os.path.exists(merged)
    assert out["model_paths"][-1] == merged and out["model_paths"][0].endswith("best_model.pth.tar")
    mm = torch.load(merged, weights_only=False)
    last = torch.load(os.path.join(tdir, "best_model.pth.tar"), weights_only=False)
    same = all(torch.equal(a.detach().cpu(), b.detach().cpu()) for a, b in zip(mm.parameters(), last.parameters()))
    if mode == "mean":
        assert same, "the reference's mean-IMM leaves the task model unmerged (merge.py:223-239)"
    else:
        assert not same
        assert os.path.exists(os.path.join(tdir, "precision_mode.pth.tar"))
        assert os.path.exists(os.path.join(tdir, "sum_precision_mode.pth.tar"))
        rp = last.reg_params
        assert all(bool((rp[p]["omega"] == 1).all()) for p in last.parameters() if p in rp)


# --------------------------------------------------------------------------- LwF
def test_lwf_through_driver(tmp_path):
    """LWF through the two-phase driver with --test: task >= 2 models are AlexNet_LwF wrappers with one head per task,
    phase 1 unwraps them (main_SGD.py:50-53), evaluation picks the task's head."""
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    from clsurvey_amd.methods.lwf import AlexNet_LwF
    root = str(tmp_path)
    ds = _dataset(root)
    _friendly_base_model(root)
    driver.main(COMMON + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    lwf = M.parse("LWF")
    lwf.hyperparams["lambda"] = 1.0
    out = driver.main(COMMON + ["--method_name", "LWF", "--results_root", root, "--test"], method=lwf, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1, 2]
    accs = [a for i in res for a in res[i]["seq_res"][i]]
    assert all(0.0 <= a <= 100.0 for a in accs)
    assert res[0]["seq_res"][0][0] > 40.0 and res[2]["seq_res"][2][0] > 30.0, res
    last = torch.load(out["model_paths"][-1], weights_only=False)
    assert isinstance(last, AlexNet_LwF) and last.last_layer_name == 4
    heads = list(last.model.classifier.children())[4:]
    assert len(heads) == 3 and all(h.out_features == 4 for h in heads)
    assert last.reg_params["reg_lambda"] == out["frameworks"][-1].hyperparams["lambda"]


# --------------------------------------------------------------------------- AlexNet-structured nets (BASELINE configs[3])
def test_gem_alexnet_through_driver(tmp_path):
    """'GEM AlexNet': an AlexNet-structured base model (11x11/4, 5x5, 3x3 convs, 3x3/2 max-pools, Dropout classifier;
    narrow so the test stays small) on 224x224 tasks through the driver: SI first-task dump (nn.Dropout semantics in
    the executor), then GEM (its own per-observe masks) with --test."""
    from clsurvey_amd import models
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import method as M
    root = str(tmp_path)
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=4, sizes=(96, 24, 24), hw=224,
                               noise=0.4, name="tiny224")
    torch.manual_seed(0)
    m = models.AlexNet(num_classes=4, widths=(16, 24, 32, 32, 16), fc=128, feat_hw=6)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d)):
            torch.nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.1     # at p = 0.5 this narrow net does not leave chance level on 96 samples (torch-CPU does not either)
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    torch.save(m, os.path.join(root, "models", "alexnet_scratch.pth.tar"))
    common = ["alexnet_scratch", "--lr_grid", "3e-3", "--num_epochs", "12", "--batch_size", "24", "--saving_freq", "100"]
    driver.main(common + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    gem = M.parse("GEM")
    gem.static_hyperparams = {"mem_per_task": 16}
    out = driver.main(common + ["--method_name", "GEM", "--results_root", root, "--test"], method=gem, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1]
    accs = [a for i in res for a in res[i]["seq_res"][i]]
    assert all(0.0 <= a <= 100.0 for a in accs)
    assert res[0]["seq_res"][0][0] > 40.0, res
    last = torch.load(out["model_paths"][-1], weights_only=False)
    assert last.observed_tasks == [0, 1] and last.cum_nc_per_task == [4, 8]
    assert isinstance(last.net, models.AlexNet) and sorted(last.engine.drops) == [5, 6]
    lo = last(torch.randn(3, 3, 224, 224, device="cuda"), 1)
    assert lo.shape == (3, 8) and bool((lo[:, :4] < -1e10).all())


# --------------------------------------------------------------------------- '_DROP_BN' model variants
def test_drop_bn_variant_through_driver(tmp_path):
    """A '_DROP_BN' VGGSlim (conv-BatchNorm-ReLU features, Dropout classifier, last_layer_idx 6) through the driver: SI
    first-task dump, then EWC and MAS with --test; BatchNorm weights / biases carry importance weights like any other
    parameter and the running statistics travel in the pickles."""
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import method as M
    root = str(tmp_path)
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=4, sizes=(160, 40, 40), hw=32,
                               noise=0.4, name="tiny2")
    from clsurvey_amd import models
    torch.manual_seed(0)
    m0 = models.parse_model_name("small_VGG9_cl_128_128_DROP_BN", (32, 32), 4)      # kaiming classifier, as _friendly_base_model
    for mod in m0.modules():
        if isinstance(mod, torch.nn.Linear):
            torch.nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    torch.save(m0, os.path.join(root, "models", "small_VGG9_cl_128_128_DROP_BN.pth.tar"))
    common = ["small_VGG9_cl_128_128_DROP_BN", "--lr_grid", "1e-2", "--num_epochs", "8", "--batch_size", "40", "--saving_freq", "100"]
    out = driver.main(common + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                      method=M.parse("SI"), dataset=ds)
    m = torch.load(out["manager"].best_model_path, weights_only=False)
    bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
    assert len(bns) == 6 and len(m.classifier._modules) == 7
    assert all(int(b.num_batches_tracked) > 0 and float(b.running_mean.abs().sum()) > 0 for b in bns)
    assert all(b.weight in m.reg_params for b in bns)
    for name in ("EWC", "MAS"):
        out = driver.main(common + ["--method_name", name, "--results_root", root, "--test"], method=M.parse(name), dataset=ds)
        res = out["results"]
        assert sorted(res) == [0, 1]
        accs = [a for i in res for a in res[i]["seq_res"][i]]
        assert all(0.0 <= a <= 100.0 for a in accs)
        assert res[0]["seq_res"][0][0] > 40.0, (name, res)
        mt = torch.load(out["model_paths"][-1], weights_only=False)
        b0 = [mod for mod in mt.modules() if isinstance(mod, torch.nn.BatchNorm2d)][0]
        assert float(mt.reg_params[b0.weight]["omega"].abs().max()) > 0


# --------------------------------------------------------------------------- EBLL
def test_ebll_through_driver(tmp_path):
    """EBLL through the driver with --test: per task an autoencoder grid on the previous task (prestep), phase-1 finetune
    (unwrapping the EBLL wrapper), phase-2 training with distillation + code loss; evaluation picks the task's head."""
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    from clsurvey_amd.methods.ebll import AlexNet_EBLL
    root = str(tmp_path)
    ds = _dataset(root)
    _friendly_base_model(root)
    driver.main(COMMON + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    ebll = M.parse("EBLL")
    ebll.static_hyperparams = {"autoencoder_lr": [0.01], "autoencoder_epochs": 4, "encoder_alphas": [1e-1], "encoder_dims": [16, 8]}
    out = driver.main(COMMON + ["--method_name", "EBLL", "--results_root", root, "--test"], method=ebll, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1, 2]
    accs = [a for i in res for a in res[i]["seq_res"][i]]
    assert all(0.0 <= a <= 100.0 for a in accs) and res[0]["seq_res"][0][0] > 40.0, res
    last = torch.load(out["model_paths"][-1], weights_only=False)
    assert isinstance(last, AlexNet_EBLL) and len(last.autoencoders._modules) == 2 and len(last.classifier._modules) == 7
    assert set(last.reg_params) >= {"lambda", "reg_alpha"}
    enc_dir = os.path.join(out["manager"].parent_exp_dir, "task_1", "ENCODER_TRAINING")
    assert os.path.exists(os.path.join(enc_dir, "grid_checkpoint.pth"))
    assert len([d for d in os.listdir(enc_dir) if d.startswith("dim=")]) == 1       # the losing grid node is removed


def test_end_to_end_mas_si_match_reference_driver_g17(tmp_path, golden):
    """BASELINE configs[2] methods at the framework level: same tiny 3-task sequence, start weights, flags and seeds as the
    runs of the reference's UNCHANGED framework/main.py recorded in G17 (make_g17.py) — SI first-task dump, then MAS and
    SI with --test.  Same criteria as the EWC run of G10: grid accuracies, every phase-2 attempt, the final state,
    the just-trained task's test accuracy; importance-weight statistics to their order of magnitude."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from g10_weights import det_weights
    from clsurvey_amd import models
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import method as M
    g = golden("G17_framework_mas_si")
    root = str(tmp_path)
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=3, classes_per_task=4, sizes=(160, 40, 40),
                               hw=32, noise=0.4, name="tiny3")
    m = models.parse_model_name("small_VGG9_cl_128_128", (32, 32), 4)
    with torch.no_grad():
        for p, w in zip(m.parameters(), det_weights()):
            p.copy_(torch.from_numpy(w))
    os.makedirs(os.path.join(root, "models"))
    torch.save(m, os.path.join(root, "models", "small_VGG9_cl_128_128.pth.tar"))
    flags = COMMON + ["--drop_margin", "0.05"]
    driver.main(flags + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    one = 2.0 / 40 + 1e-9
    three = 100.0 * 3 / 40 + 1e-9
    for name in ("MAS", "SI"):
        pre = name.lower() + "_"
        meth = M.parse(name)
        meth.hyperparams["lambda"] = float(g[pre + "lambda0"])
        out = driver.main(flags + ["--method_name", name, "--results_root", root, "--test"], method=meth, dataset=ds)
        assert out["args"].exp_name == str(g[pre + "exp_name"])
        ref_attempts = g[pre + "attempts"]
        mine = [(t, tr[0]["lambda"], tr[1]) for t, hf in zip((2, 3), out["frameworks"][1:]) for tr in hf.trace]
        assert len(mine) == len(ref_attempts), (name, mine, ref_attempts)
        for a, b in zip(mine, ref_attempts):
            assert a[0] == int(b[0]) and a[1] == float(b[1]) and abs(a[2] - float(b[2])) <= one, (name, mine, ref_attempts)
        for t, hf in zip((2, 3), out["frameworks"][1:]):
            assert abs(hf.trace[-1][1] - float(g[pre + "t%d_val_acc" % t])) <= one, (name, t, hf.trace)
            assert abs(hf.trace[-1][2] - float(g[pre + "t%d_threshold" % t])) <= one
            assert hf.attempts == int(g[pre + "t%d_attempts" % t]) and hf.hyperparams["lambda"] == float(g[pre + "t%d_lambda" % t])
            tdir = os.path.join(out["manager"].parent_exp_dir, "task_%d" % t)
            grid = torch.load(os.path.join(tdir, "FT_LR_GRIDSEARCH", "grid_checkpoint.pth"), weights_only=False)["processed_lrs"]
            for lr in (1e-2, 3e-3):
                assert abs(grid[lr]["acc"][0] - float(g[pre + "t%d_lr%g" % (t, lr)][0])) <= one, (name, t, lr, grid[lr])
            mt = torch.load(os.path.join(tdir, "TASK_TRAINING", "best_model.pth.tar"), weights_only=False)
            om = [mt.reg_params[p]["omega"].double() for p in mt.parameters() if p in mt.reg_params]
            ref = g[pre + "t%d_omega_stats" % t]
            assert len(om) == ref.shape[0], (name, t, len(om), ref.shape)
            for o, r in zip(om, ref):
                st = np.array([float(o.sum()), float(o.max()), float(o.pow(2).sum().sqrt())])
                # chaotic trajectories (see the G10 test): order of magnitude only; the arithmetic is pinned by G3 / G4
                assert np.all((st <= 3.0 * r + 1e-9) & (st >= r / 3.0 - 1e-9)), (name, t, st, r)
        res = out["results"]
        for i in range(3):
            got, ref = np.array(res[i]["seq_res"][i]), g[pre + "seq_res%d" % i]
            assert got.shape == ref.shape and abs(got[0] - ref[0]) <= three, (name, i, got, ref)
            assert np.all((got >= 0) & (got <= 100))
            assert np.array(res[i]["seq_forgetting"][i]).shape == g[pre + "seq_forgetting%d" % i].shape
        print("G17", name, "seq_res build:", {i: res[i]["seq_res"][i] for i in range(3)}, " reference:",
              {i: list(g[pre + "seq_res%d" % i]) for i in range(3)})


def test_end_to_end_lwf_ebll_match_reference_driver_g18(tmp_path, golden):
    """LwF and EBLL at the framework level against runs of the reference's UNCHANGED framework/main.py (G18,
    make_g18.py): same tasks, start weights, flags, seeds (and, for EBLL, the same small autoencoder grid).  Checked: the
    experiment name, phase-1 grid accuracies, every phase-2 attempt with its decayed hyper-parameter, the final state, the
    stacked heads / chosen code sizes of the saved wrappers, the just-trained task's test accuracy."""
    import collections
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from g10_weights import det_weights
    from clsurvey_amd import models
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import method as M
    g = golden("G18_framework_lwf_ebll")
    root = str(tmp_path)
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=3, classes_per_task=4, sizes=(160, 40, 40),
                               hw=32, noise=0.4, name="tiny3")
    m = models.parse_model_name("small_VGG9_cl_128_128", (32, 32), 4)
    with torch.no_grad():
        for p, w in zip(m.parameters(), det_weights()):
            p.copy_(torch.from_numpy(w))
    os.makedirs(os.path.join(root, "models"))
    torch.save(m, os.path.join(root, "models", "small_VGG9_cl_128_128.pth.tar"))
    flags = COMMON + ["--drop_margin", "0.05"]
    driver.main(flags + ["--method_name", "SI", "--results_root", root, "--runmode", "first_task_basemodel_dump"],
                method=M.parse("SI"), dataset=ds)
    one = 2.0 / 40 + 1e-9
    three = 100.0 * 3 / 40 + 1e-9
    for name, key in (("LWF", "lambda"), ("EBLL", "reg_lambda")):
        pre = name.lower() + "_"
        meth = M.parse(name)
        if name == "LWF":
            meth.hyperparams = collections.OrderedDict({"lambda": float(g[pre + "lambda0"])})
        else:
            meth.hyperparams = collections.OrderedDict({"reg_lambda": float(g[pre + "lambda0"]), "ebll_reg_alpha": 1})
            meth.static_hyperparams = collections.OrderedDict({"autoencoder_lr": [0.01], "autoencoder_epochs": 4,
                                                               "encoder_alphas": [1e-1], "encoder_dims": [16, 8]})
        out = driver.main(flags + ["--method_name", name, "--results_root", root, "--test"], method=meth, dataset=ds)
        assert out["args"].exp_name == str(g[pre + "exp_name"]), (out["args"].exp_name, str(g[pre + "exp_name"]))
        ref_attempts = g[pre + "attempts"]
        mine = [(t, tr[0][key], tr[1]) for t, hf in zip((2, 3), out["frameworks"][1:]) for tr in hf.trace]
        print("G18", name, "attempts build:", mine, " reference:", ref_attempts.tolist())
        # Task 2 starts from the SAME model on both sides: its attempt trace (incl. the failed first attempt of LwF at
        # lambda 10: 40 % after 8 epochs, then 100 % at lambda 5) must agree.  Task 3 starts from two task-2 models that
        # differ at fp32 round-off and its 8-epoch accuracy sits on a cliff (reference LwF: 62.5 % at lambda 5, 100 % at
        # 2.5), so there only the outcome is checked: the accepted attempt clears the same threshold.
        # EBLL's autoencoder stage sits between the phase-1 grid and the first attempt and draws from the global generator
        # (autoencoder init, one shuffle per epoch until ITS early stop), so the new head's init is not the reference's
        # any more: lambda 10 ends at 100 % there and at 42.5 % here — as LwF does at lambda 10 on both sides.
        mine2, ref2 = [a for a in mine if a[0] == 2], [b for b in ref_attempts if int(b[0]) == 2]
        if name == "LWF":
            assert len(mine2) == len(ref2), (name, mine, ref_attempts)
            for a, b in zip(mine2, ref2):
                assert a[1] == float(b[1]) and abs(a[2] - float(b[2])) <= one, (name, mine, ref_attempts)
        assert mine[0][1] == float(ref_attempts[0][1]) and all(0.0 <= a[2] <= 1.0 for a in mine)
        for t, hf in zip((2, 3), out["frameworks"][1:]):
            assert abs(hf.trace[-1][1] - float(g[pre + "t%d_val_acc" % t])) <= one, (name, t, hf.trace)
            assert abs(hf.trace[-1][2] - float(g[pre + "t%d_threshold" % t])) <= one
            if t == 2 and name == "LWF":
                assert hf.attempts == int(g[pre + "t%d_attempts" % t]) and hf.hyperparams[key] == float(g[pre + "t%d_lambda" % t])
            tdir = os.path.join(out["manager"].parent_exp_dir, "task_%d" % t)
            grid = torch.load(os.path.join(tdir, "FT_LR_GRIDSEARCH", "grid_checkpoint.pth"), weights_only=False)["processed_lrs"]
            for lr in (1e-2, 3e-3):
                assert abs(grid[lr]["acc"][0] - float(g[pre + "t%d_lr%g" % (t, lr)][0])) <= one, (name, t, lr, grid[lr])
            mt = torch.load(os.path.join(tdir, "TASK_TRAINING", "best_model.pth.tar"), weights_only=False)
            cls = mt.model.classifier if hasattr(mt, "model") else mt.classifier
            assert len(cls._modules) == int(g[pre + "t%d_n_heads" % t])
            if name == "EBLL":
                assert len(mt.autoencoders._modules) == int(g[pre + "t%d_n_encoders" % t])
                assert [e[0].out_features for e in mt.autoencoders._modules.values()] == list(g[pre + "t%d_code_dim" % t])
        res = out["results"]
        for i in range(3):
            got, ref = np.array(res[i]["seq_res"][i]), g[pre + "seq_res%d" % i]
            assert got.shape == ref.shape and abs(got[0] - ref[0]) <= three, (name, i, got, ref)
            assert np.all((got >= 0) & (got <= 100))
        print("G18", name, "seq_res build:", {i: res[i]["seq_res"][i] for i in range(3)}, " reference:",
              {i: list(g[pre + "seq_res%d" % i]) for i in range(3)})


# --------------------------------------------------------------------------- teacher-forced importance weights (G19)
def _g19_model_and_task(g, root):
    from clsurvey_amd import models
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=3, classes_per_task=4, sizes=(160, 40, 40), hw=32,
                               noise=0.4, name="tiny3")
    m = models.parse_model_name("small_VGG9_cl_128_128", (32, 32), 4)
    with torch.no_grad():
        for i, p in enumerate(m.parameters()):
            p.copy_(torch.from_numpy(g["theta%d" % i]))
    return m, ds.get_task_dataset_path("1")


def _g19_compare(g, tag, model, tol):
    import numpy as np
    worst = 0.0
    for i, p in enumerate(model.parameters()):
        om = model.reg_params[p]["omega"].detach().double().cpu()
        ref_stats = g["%s_stats%d" % (tag, i)]
        flat = om.float().numpy().reshape(-1)
        if flat.size > (1 << 16):
            flat = flat[np.sort(np.random.RandomState(19).choice(flat.size, 8192, replace=False))]
        ref = g["%s_omega%d" % (tag, i)]
        scale = max(float(ref_stats[1]), 1e-30)
        err = float(np.abs(flat - ref).max()) / scale
        worst = max(worst, err)
        assert err <= tol, "%s Omega of tensor %d: %.3e of its maximum" % (tag, i, err)
        assert abs(float(om.sum()) - ref_stats[0]) <= tol * abs(ref_stats[0]) + 1e-12
        assert abs(float(om.pow(2).sum().sqrt()) - ref_stats[2]) <= tol * ref_stats[2] + 1e-12
        assert torch.equal(model.reg_params[p]["init_val"].cpu(), p.data.cpu())
    return worst


def test_teacher_forced_omega_matches_reference_g19(tmp_path, golden):
    """Omega tensors, element by element: the reference's accumulate_EWC_weights / accumulate_objective_based_weights
    were run (make_g19.py) on the first-task checkpoint its own framework trained; the same calls on the HIP path, started
    from that checkpoint's state_dict, must give the same tensors within north_star's 1e-3 (of each tensor's maximum)."""
    from clsurvey_amd.methods import ewc, mas
    g = golden("G19_teacher_forced_omega")
    for tag, fn in (("ewc", lambda m, p: ewc.accumulate_EWC_weights(None, [p], m, 40)),
                    ("mas", lambda m, p: mas.accumulate_objective_based_weights(None, [p], m, 40, "L2", test_set="train"))):
        m, task1 = _g19_model_and_task(g, str(tmp_path / tag))
        out = fn(m.to("cuda"), task1)
        worst = _g19_compare(g, tag, out, 1e-3)
        print("G19 %s: worst Omega element %.2e of its tensor's maximum" % (tag, worst))


def test_hat_alexnet_through_driver(tmp_path):
    """HAT on an AlexNet-structured base model (run.py:62-63 -> networks/alexnet_hat.py; narrow so the test stays small, 6x6
    feature map as the net asserts) on 224x224 tasks through the two-phase driver with --test: the trainer picks
    HatNetAlexnet, trains without the warm-up phase under nn.Dropout semantics (masks while training, none in the validation
    and test passes), and the saved wrapper reloads with its embeddings and plan."""
    from clsurvey_amd import models
    from clsurvey_amd.framework import driver
    from clsurvey_amd.framework.tasks import SyntheticTaskSequence
    from clsurvey_amd.methods import hat as HT
    from clsurvey_amd.methods import method as M
    root = str(tmp_path)
    ds = SyntheticTaskSequence(os.path.join(root, "data"), task_count=2, classes_per_task=4, sizes=(96, 24, 24), hw=224,
                               noise=0.4, name="tiny224")
    torch.manual_seed(0)
    m = models.AlexNet(num_classes=4, widths=(16, 24, 32, 32, 16), fc=128, feat_hw=6)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d)):
            torch.nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.1
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    torch.save(m, os.path.join(root, "models", "alexnet_scratch.pth.tar"))
    hat = M.parse("HAT")
    hat.hyperparams["smax"], hat.hyperparams["c"] = 50.0, 0.75
    out = driver.main(["alexnet_scratch", "--lr_grid", "1e-2,3e-3", "--num_epochs", "8", "--batch_size", "24", "--saving_freq", "100",
                       "--method_name", "HAT", "--results_root", root, "--test"], method=hat, dataset=ds)
    res = out["results"]
    assert sorted(res) == [0, 1]
    accs = [a for i in res for a in res[i]["seq_res"][i]]
    assert all(0.0 <= a <= 100.0 for a in accs)
    # (plumbing only: with seven gated layers at sigma(s e) ~ 0.5 early in every epoch this narrow net does not leave chance
    #  level on 96 samples in 8 epochs, and diverges at HAT's default LR of 0.05; the arithmetic of the step is pinned by G21)
    mt = torch.load(out["model_paths"][-1], weights_only=False)
    assert isinstance(mt, HT.HatNetAlexnet) and mt.enable_warmup is False and mt.smid == 6 and abs(mt.drop_p - 0.1) < 1e-9
    assert len(mt.conv_embs) == 5 and len(mt.fc_embs) == 2 and mt.conv_embs[0].weight.shape[0] == 2
