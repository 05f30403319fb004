"""CPU tests of the host logic: framework control flow (grid search, stability decay, eval dict
layout) with a fake Method — no tensors, no GPU."""
import copy
import os
from collections import OrderedDict

import torch

from clsurvey_amd.framework import driver


class FakeDataset:
    name = argname = test_results_dir = train_exp_results_dir = "fake"
    task_count = 3
    input_size = (32, 32)
    classes_per_task = OrderedDict((str(i), ["a", "b"]) for i in range(1, 4))

    def get_taskname(self, i):
        return str(i)

    def get_task_dataset_path(self, task_name=None, rnd_transform=False):
        return "ds_" + str(task_name)


class FakeMethod:
    """acc of phase 1 depends on lr; phase 2 acc rises as lambda decays."""
    name = eval_name = "EWC"
    hyperparams = OrderedDict({"lambda": 400})
    calls = []

    def grid_train(self, args, manager, lr):
        os.makedirs(manager.gridsearch_exp_dir, exist_ok=True)
        return None, {1e-2: 0.5, 5e-3: 0.8, 1e-3: 0.6}.get(lr, 0.1)

    def grid_poststep(self, args, manager):
        pass

    def train(self, args, manager, hyperparams):
        FakeMethod.calls.append((args.task_counter, hyperparams["lambda"], args.lr))
        os.makedirs(manager.heuristic_exp_dir, exist_ok=True)
        torch.save({"lam": hyperparams["lambda"]}, os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar"))
        return None, 0.8 - hyperparams["lambda"] / 1000.0      # 0.4, 0.6, 0.7 ...

    def inference_eval(self, args, manager):
        return 90.0 - 10.0 * (args.trained_model_idx - args.eval_dset_idx)

    def get_output(self, images, args):
        raise NotImplementedError


def test_two_phase_framework_and_eval(tmp_path):
    FakeMethod.calls = []
    m = FakeMethod()
    m.hyperparams = copy.deepcopy(FakeMethod.hyperparams)
    root = str(tmp_path)
    # task 1 model comes from the SI bootstrap path (main.py:226-233): create the file it expects
    si = os.path.join(root, "train", "fake", "SI", "small_VGG9_cl_128_128", "gridsearch", "first_task_basemodel",
                      "e=1_bs=200_lr=[0.001, 0.005, 0.01]", "task_1", "TASK_TRAINING")     # models/net.py:39-53 naming
    os.makedirs(si)
    torch.save({}, os.path.join(si, "best_model.pth.tar"))
    out = driver.main(["small_VGG9_cl_128_128", "--method_name", "EWC", "--results_root", root, "--lr_grid",
                       "1e-2,5e-3,1e-3", "--test", "--num_epochs", "1"], method=m, dataset=FakeDataset())
    # task 1 skipped (SI model), tasks 2 and 3: best lr 5e-3 (acc .8) -> threshold .64
    # lambda 400 -> .4 (<.64) decay 200 -> .6 (<.64) decay 100 -> .7 ok; task 3 starts from lambda 100
    assert FakeMethod.calls == [(2, 400, 5e-3), (2, 200.0, 5e-3), (2, 100.0, 5e-3), (3, 100.0, 5e-3)]
    hf2, hf3 = out["frameworks"][1], out["frameworks"][2]
    assert [round(t[2], 6) for t in hf2.trace] == [0.64, 0.64, 0.64]
    assert [t[0]["lambda"] for t in hf2.trace] == [400, 200.0, 100.0]
    assert len(hf3.trace) == 1
    tdir = os.path.join(out["manager"].parent_exp_dir, "task_2", "TASK_TRAINING")
    assert os.path.exists(os.path.join(tdir, "SUCCESS.FLAG"))
    st = torch.load(os.path.join(tdir, "hyperparams.pth.tar"), weights_only=False)
    assert st["state"]["attempts"] == 2 and st["state"]["hyperparams"]["lambda"] == 100.0
    res = out["results"]
    assert res[0]["seq_res"][0] == [90.0, 80.0, 70.0] and res[0]["seq_forgetting"][0] == [10.0, 20.0]
    assert res[2]["seq_res"][2] == [90.0]
    f = os.path.join(out["args"].out_path, "test_method_performancesEWC1.pth")
    assert "seq_head_acc" in torch.load(f, weights_only=False)["EWC"]


def test_multi_hyperparam_decay_order():
    """framework_train.py:176-184 worked example."""
    class M:
        hyperparams = OrderedDict([("lambda", 5.0), ("alpha", 2.0)])
    hf = driver.HyperparameterFramework(M())

    class A:
        decaying_factor = 0.5

    class Mgr:
        method = M()
    seq = []
    for _ in range(5):
        hf.hyperparamDecay(A(), Mgr())
        seq.append((hf.hyperparams["lambda"], hf.hyperparams["alpha"]))
    assert seq == [(2.5, 2.0), (5.0, 1.0), (2.5, 1.0), (1.25, 1.0), (2.5, 0.5)]


def test_set_hyperparams_grammar():
    from clsurvey_amd.methods import method as M
    m = M.parse("EWC")
    M.set_hyperparams(m, "200")
    assert m.hyperparams["lambda"] == 200.0
    M.set_hyperparams(m, "def")
    assert m.hyperparams["lambda"] == 200.0
    assert M.parse("MAS").hyperparams["lambda"] == 3 and M.parse("SI").hyperparams["lambda"] == 400


def test_device_loader_reproduces_dataloader_order():
    from torch.utils.data import DataLoader
    from clsurvey_amd.data import DeviceLoader, TensorTaskDataset
    ds = TensorTaskDataset(torch.arange(23).float().view(23, 1, 1, 1).expand(23, 3, 2, 2), torch.arange(23), list(range(5)))
    torch.manual_seed(7)
    ref = [b[1].tolist() for _ in range(2) for b in DataLoader(ds, batch_size=5, shuffle=True)]
    torch.manual_seed(7)
    dl = DeviceLoader(ds, 5, True, "cpu")
    assert [b[1].tolist() for _ in range(2) for b in dl] == ref and len(dl) == 5


def test_ebll_autoencoder_grid_host_logic(tmp_path, monkeypatch):
    """EBLL.prestep's autoencoder grid (method.py:842-908): every (dim, alpha, lr) node trained once, results checkpointed,
    finished nodes skipped on a re-run, the best node's directory kept and the others removed."""
    import types
    from clsurvey_amd.methods import method as M
    from clsurvey_amd.methods import ebll as E
    calls = []

    def fake_autoencoder(dataset_path, previous_task_model_path, exp_dir, batch_size, num_epochs, lr, alpha, last_layer_name,
                         auto_dim, device="cuda"):
        calls.append((auto_dim, alpha, lr))
        torch.save({"dim": auto_dim}, os.path.join(exp_dir, "best_model.pth.tar"))
        return None, {(100, 0.1): 0.55, (100, 0.01): 0.70, (300, 0.1): 0.62, (300, 0.01): 0.41}[(auto_dim, alpha)]

    monkeypatch.setattr(E, "fine_tune_Adam_Autoencoder", fake_autoencoder)
    ebll = M.parse("EBLL")
    assert list(ebll.hyperparams) == ["reg_lambda", "ebll_reg_alpha"] and ebll.extra_hyperparams_count == 2
    args = types.SimpleNamespace(task_counter=2, previous_task_dataset_path="d", batch_size=8, classifier_heads_starting_idx=4,
                                 presteps_elapsed_time=0)
    manager = types.SimpleNamespace(parent_exp_dir=str(tmp_path), previous_task_model_path="m")
    ebll.prestep(args, manager)
    enc = os.path.join(str(tmp_path), "task_1", "ENCODER_TRAINING")
    assert manager.autoencoder_model_path == os.path.join(enc, "dim=100_alpha=0.01_lr=0.01", "best_model.pth.tar")
    assert os.path.exists(manager.autoencoder_model_path)
    assert sorted(os.listdir(enc)) == ["dim=100_alpha=0.01_lr=0.01", "grid_checkpoint.pth"]
    assert len(calls) == 4 and args.presteps_elapsed_time >= 0
    ck = torch.load(os.path.join(enc, "grid_checkpoint.pth"), weights_only=False)
    assert ck[(100, 0.01, 0.01)] == 0.70 and ck["header"] == ("dim", "alpha", "lr")
    ebll.prestep(args, manager)                      # everything is in the checkpoint: nothing is retrained
    assert len(calls) == 4


def test_task_dataset_cache_passthrough_on_cpu(tmp_path):
    """load_task_datasets: dict passes through; without a device the file is simply unpickled (no HBM cache)."""
    from clsurvey_amd import data as D
    d = D.synthetic_task(8, 4, 4, 2, hw=8, seed=1)
    assert D.load_task_datasets(d) is d
    p = os.path.join(str(tmp_path), "t.pth.tar")
    torch.save(d, p)
    got = D.load_task_datasets(p)
    assert sorted(got) == ["test", "train", "val"] and torch.equal(got["train"].x, d["train"].x)


# --------------------------------------------------------------------------- G9: the PRODUCT's schedules vs the reference's
def test_product_set_lr_matches_reference_traces_g9():
    """train_common.set_lr driven exactly as train_model drives it (count of epochs without a new best validation
    accuracy) against the traces the reference's own set_lr produced (EWC/train_EWC.py:89-101 'gt', SI/train_SI.py:129-141
    'ge'): LR used and continue flag per epoch."""
    import numpy as np
    from clsurvey_amd.methods import train_common as tc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G9_schedules.npz"), allow_pickle=True)
    for tag, rule in (("ewc", "gt"), ("si", "ge")):
        for case in range(3):
            opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.01)
            lr, count, trace = 0.01, 0, []
            for ep, imp in enumerate(g["%s_c%d_improved" % (tag, case)]):
                opt, lr, cont = tc.set_lr(opt, lr, count, rule)
                trace.append((ep, lr, float(cont)))
                assert opt.param_groups[0]["lr"] == lr or count != 5
                if not cont:
                    break
                count = 0 if imp else count + 1
            ref = g["%s_c%d_trace" % (tag, case)]
            assert len(trace) == len(ref)
            for a, b in zip(trace, ref):
                assert a[0] == int(b[0]) and a[2] == b[2] and abs(a[1] - b[1]) <= 1e-12


def test_product_hyperparam_decay_matches_reference_g9():
    """driver.HyperparameterFramework.hyperparamDecay against the reference's framework_train.py:168-216 sequences for
    1 / 2 / 3 hyper-parameters (one at a time, then all together from the updated restore values) and for a method with
    its own decay_operator; shard.decayed_copy (speculative attempt k) must land on the same values without touching
    the original."""
    import collections
    import operator
    import types
    import numpy as np
    from clsurvey_amd.framework import driver, shard
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G9_schedules.npz"), allow_pickle=True)
    for tag, op in (("one", None), ("two", None), ("three", None), ("sub", operator.sub)):
        ref = g["decay_%s" % tag]
        keys = [str(k) for k in g["decay_%s_keys" % tag]]
        method = types.SimpleNamespace(hyperparams=collections.OrderedDict(zip(keys, ref[0].tolist())))
        if op is not None:
            method.decay_operator = op
        hf = driver.HyperparameterFramework(method)
        args = types.SimpleNamespace(decaying_factor=0.5)
        manager = types.SimpleNamespace(method=method)
        for k in range(1, len(ref)):
            twin = shard.decayed_copy(hf, args, manager, 1)
            hf_before = (dict(hf.hyperparams), dict(hf.hyperparams_backup), hf.hyperparam_idx, hf.attempts)
            assert [twin.hyperparams[key] for key in keys] == ref[k].tolist()
            assert hf_before == (dict(hf.hyperparams), dict(hf.hyperparams_backup), hf.hyperparam_idx, hf.attempts)
            hf.hyperparamDecay(args, manager)
            assert [hf.hyperparams[key] for key in keys] == ref[k].tolist(), (tag, k)
        hf0 = driver.HyperparameterFramework(types.SimpleNamespace(hyperparams=collections.OrderedDict(zip(keys, ref[0].tolist()))))
        method0 = types.SimpleNamespace(hyperparams=hf0.hyperparams)
        if op is not None:
            method0.decay_operator = op
        far = shard.decayed_copy(hf0, args, types.SimpleNamespace(method=method0), 7)
        assert [far.hyperparams[key] for key in keys] == ref[7].tolist()


def test_patience_plan_reference_sequences():
    """PatiencePlan against a literal transcription of the two inline schedules it replaces (HAT/approaches/hat.py:150-166
    with `patience <= 0`, hat_finetune.py:108-124 with `patience == 0`): same verdict and LR at every epoch."""
    import numpy as np
    from clsurvey_amd.methods.train_common import PatiencePlan
    rs = np.random.RandomState(3)
    for leq in (True, False):
        for trial in range(20):
            accs = np.round(rs.rand(60) * (0.5 + 0.5 * rs.rand()), 2)
            plan = PatiencePlan(0.05, 6, 2, stop_at_or_below_zero=leq)
            best, patience, lr = 0, 6, 0.05
            for a in accs:
                want = "hold"
                if a > best:
                    best, patience, want = a, 6, "best"
                else:
                    patience -= 1
                    if patience == 6 // 2:
                        lr /= 2
                        want = "decay"
                    elif (patience <= 0) if leq else (patience == 0):
                        want = "stop"
                got = plan.observe(a)
                assert got == want and plan.lr == lr and plan.patience == patience and plan.best == best
                if want == "stop" and not leq:
                    break


def test_traffic_json_names_the_kernel_bench_reports():
    """profiles/traffic.json holds PMC-measured HBM bytes of the dominant launch; bench.py copies it into
    roofline.traffic.  The entry must be for the kernel INSTANCE the library launches for that layer today (the tile
    geometry decides the halo re-reads), otherwise bench reports null instead of a stale number."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    # layer 2 of the bench model: forward / backward-data through the bf16-split kernel (large map: clhip_internal_bs_preferred) —
    # the entries must name THOSE instances; the Winograd forward / backward-data measurements of round 4 stay on file under 'wino_round4'
    inst = bench.bs_conv_instance(200, 32, 32, 64, 0, False)
    e = t["conv3x3_relu_pool_fwd 64x64@32 N=200"]
    assert e["instance"] == inst, (e["instance"], inst)
    assert bench.measured_traffic("conv3x3_relu_pool_fwd", "64x64@32", 200, inst) == e["hbm_bytes_per_launch"]
    assert bench.measured_traffic("conv3x3_relu_pool_fwd", "64x64@32", 200, "conv3x3_mfma_kernel<other>") is None
    assert e["wino_round4"]["instance"] == bench.wino_conv_instance(32, 0, False)
    assert e["wino_round4"]["direct_kernel_round2"]["instance"] == bench.conv_instance(64, 32, 32, 200, 64, 0, True)
    assert t["conv3x3_bwd_data_unpool 64x64@32 N=200"]["instance"] == bench.bs_conv_instance(200, 32, 32, 64, 1, True)
    assert t["conv3x3_bwd_data_unpool 64x64@32 N=200"]["wino_round4"]["instance"] == bench.wino_conv_instance(32, 1, True)
    assert "wino_conv16g_kernel<8, 2, 4, 1, true>" in bench.wino_conv_instance(32, 1, True)
    # round 6: the layer-2 weight gradient is the bf16-split kernel (csrc/bswgrad.hip); the Winograd kernel's values stay under 'wino_round5'
    w = t["conv3x3_bwd_weight_unpool 64x64@32 N=200"]
    assert w["instance"] == "bs_wgrad_kernel<true> (slabs + reduction)"
    assert w["wino_round5"]["instance"] == bench.wino_wgrad_instance(200, 64, 64, 32, 32, True)
    assert bench.measured_traffic("conv3x3_bwd_weight_unpool", "64x64@32", 200, w["instance"]) == w["hbm_bytes_per_launch"]


def test_bench_names_the_winograd_instances_the_library_launches():
    """bench.py's per-layer table names kernel instances by restating the dispatch rules of csrc/wino.hip (launch_wino,
    clhip_internal_wino_wgrad_partial); the names are what profiles/*kernel_stats*.csv rows are matched against."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # forward / backward-data: 16-tile kernel by map width; 8x8 maps with few units: the one-image-per-wave variant, 16 or 32 channels per wave
    assert bench.wino_conv_instance(32, 0, False, 200, 64).startswith("wino_conv16g_kernel<8, 2, 4, 0, false>")
    assert bench.wino_conv_instance(16, 1, True, 200, 64).startswith("wino_conv16g_kernel<8, 2, 4, 1, true>")
    assert bench.wino_conv_instance(8, 0, False, 200, 128).startswith("wino_conv16_kernel<0, false, 2>")      # 400 units of the 32-tile kernel
    assert bench.wino_conv_instance(8, 1, False, 200, 64).startswith("wino_conv16_kernel<1, false, 1>")       # <= 512 waves: 16-channel waves
    assert bench.wino_conv_instance(8, 0, False, 200, 512).startswith("wino_conv16g_kernel<4, 4, 4, 0, false>")
    # weight gradient: fewer than 16 sixteen-tile stages per 64x64-tile block -> the pixel-split kernel
    assert bench.wino_wgrad_instance(200, 64, 64, 32, 32, True).startswith("wino_wgrad_ps_kernel<8, 2, true, true>")     # 3200 stages / 256 splits
    assert bench.wino_wgrad_instance(200, 64, 64, 16, 16, False).startswith("wino_wgrad_ps_kernel<8, 2, false, true>")
    assert bench.wino_wgrad_instance(200, 128, 128, 8, 8, True).startswith("wino_wgrad_ps_kernel<4, 4, true, true>")
    assert bench.wino_wgrad_instance(200, 512, 512, 8, 8, False).startswith("wino_wgrad_kernel<4, 4, 1, false, true>")  # 200 stages / 4 splits
    assert bench.wino_wgrad_instance(200, 256, 256, 16, 16, False).startswith("wino_wgrad_kernel<8, 2, 1, false, true>")


def test_hat_alexnet_net_structure():
    """HatNetAlexnet (methods/HAT/networks/alexnet_hat.py:4-13 over vgg_hat.Net, vgg_hat.py:18-81): the parameter tree of the
    reference's dynamic construction over torchvision's AlexNet module tree, its single shared pool geometry and Dropout, and
    the plan view the executor is built over: Dropout IN FRONT of each gated Linear layer (first_drop), 3x3/2 pools behind
    convolutions 0, 1 and 4, the raw convolution geometries untouched."""
    import torch.nn as nn
    from clsurvey_amd import models
    from clsurvey_amd.methods import hat as HT
    net = HT.HatNetAlexnet(models.AlexNet(num_classes=7), (3, 224, 224), [(0, 7), (1, 7)])
    names = [n for n, _ in net.named_parameters()]
    assert names[:4] == ["convs.0.weight", "convs.0.bias", "convs.1.weight", "convs.1.bias"]
    assert [n for n in names if "embs" in n] == ["conv_embs.%d.weight" % i for i in range(5)] + ["fc_embs.0.weight", "fc_embs.1.weight"]
    assert names[-2:] == ["classifier.0.weight", "classifier.0.bias"]
    assert net.maxpool_idxs == [0, 1, 4] and net.pool_geometry == (3, 2) and net.drop_p == 0.5
    assert net.smid == 6 and net.enable_warmup is False and net.first_drop is True
    assert all(e.weight.shape == (2, c.out_channels) for e, c in zip(net.conv_embs, net.convs))
    view = net.plain_view()
    feats = list(view.features.children())
    assert [type(m).__name__ for m in feats] == ["Conv2d", "ReLU", "MaxPool2d", "Conv2d", "ReLU", "MaxPool2d", "Conv2d", "ReLU",
                                                  "Conv2d", "ReLU", "Conv2d", "ReLU", "MaxPool2d"]
    assert feats[0] is net.convs[0] and feats[0].kernel_size == (11, 11) and feats[0].stride == (4, 4) and feats[3].kernel_size == (5, 5)
    assert all(m.kernel_size == 3 and m.stride == 2 for m in feats if isinstance(m, nn.MaxPool2d))
    cls = list(view.classifier.children())
    assert [type(m).__name__ for m in cls] == ["Dropout", "Linear", "ReLU", "Dropout", "Linear", "ReLU", "Linear"]
    assert cls[1] is net.fcs[0] and cls[-1] is net.classifier[0]
    # the VGG variant keeps vgg_hat's order: Dropout (if the raw model has one) BEHIND relu(fc(x))
    vgg = HT.HatNet(models.parse_model_name("small_VGG9_cl_128_128_DROP", (64, 64), 5), (3, 64, 64), [(0, 5)])
    assert [type(m).__name__ for m in vgg.plain_view().classifier.children()] == ["Linear", "ReLU", "Dropout", "Linear", "ReLU", "Dropout", "Linear"]
    plain = HT.HatNet(models.parse_model_name("small_VGG9_cl_128_128", (64, 64), 5), (3, 64, 64), [(0, 5)])
    assert [type(m).__name__ for m in plain.plain_view().classifier.children()] == ["Linear", "ReLU", "Linear", "ReLU", "Linear"]
    assert all(m.kernel_size == 2 for m in plain.plain_view().features.children() if isinstance(m, nn.MaxPool2d))


def test_plugin_surface_matches_reference_g22():
    """The method objects the framework driver works with, against the reference's methods/method.py taken as DATA
    (fixture G22, tests/golden/make_g22.py): for every --method_name on this path the class, name / eval_name / category /
    extra_hyperparams_count, the hyper-parameter dict(s) in order with their defaults, the plain instance attributes and
    every hook the reference object offers; and set_hyperparams() on the override strings the reference accepts.
    Where the reference stops with a TypeError (a single bare value, or a 'def' placeholder: its parser drops the
    placeholder before it counts positions) the build applies the meaning the reference's docstring states; where the
    reference stores an empty list for 'def' inside a ';' list the build keeps the default — both deliberate, both asserted."""
    import json
    import pytest
    from clsurvey_amd.methods import method as M
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G22_method_table.json")) as f:
        table = json.load(f)
    pairs = lambda d: [[str(k), _plain(v)] for k, v in d.items()] if d is not None else None      # noqa: E731

    def _plain(v):
        if isinstance(v, dict):
            return pairs(v)
        if isinstance(v, (list, tuple)):
            return [_plain(x) for x in v]
        return v

    for name, ref in table["methods"].items():
        m = M.parse(name)
        assert type(m).__name__ == ref["class"], name
        assert (m.name, m.eval_name, m.extra_hyperparams_count) == (ref["name"], ref["eval_name"], ref["extra_hyperparams_count"]), name
        assert getattr(m.category, "name", str(m.category)) == ref["category"], name
        assert pairs(m.hyperparams) == ref["hyperparams"], name
        assert pairs(getattr(m, "static_hyperparams", None)) == ref["static_hyperparams"], name
        for k, v in ref["instance_attrs"].items():
            assert _plain(getattr(m, k)) == v, (name, k)
        for hook in ref["hooks"]:
            assert callable(getattr(m, hook, None)), (name, hook)
        for flag in ("start_scratch", "wrap_first_task_model", "no_framework", "grid_chkpt"):      # read with hasattr / getattr by the drivers
            if (name, flag) == ("finetuning", "no_framework"):
                # the reference's Finetune has neither a train() nor the no_framework switch: its phase 2 calls
                # manager.method.train, which does not exist, and the run ends in sys.exit(1) after the first grid
                # (framework_train.py:104-108).  The build runs the baseline grid-only, like IMM — deliberate.
                assert not ref["flags"].get(flag, False) and "train" not in ref["hooks"] and m.no_framework is True
                continue
            assert bool(getattr(m, flag, False)) == bool(ref["flags"].get(flag, False)), (name, flag)
        assert set(ref["flags"]) <= {"start_scratch", "wrap_first_task_model", "no_framework", "grid_chkpt"}, ref["flags"]
        for case in ref["overrides"]:
            mm = M.parse(name)
            M.set_hyperparams(mm, case["text"], static_params=case["static"])
            got = pairs(mm.static_hyperparams if case["static"] else mm.hyperparams)
            default = ref["static_hyperparams"] if case["static"] else ref["hyperparams"]
            if "result" in case and "def" not in case["text"]:
                assert got == case["result"], (name, case)
                assert pairs(mm.init_hyperparams) == case["init_hyperparams"], (name, case)
                continue
            # the repaired cases: positional meaning, 'def' / empty slots keep the default
            want = [list(kv) for kv in default]
            fields = [f for f in case["text"].split(";") if f.strip()]
            if len(fields) == 1:
                slots = [t.strip() for t in fields[0].split(",")]
                vals = [None if t in ("def", "") else float(t) for t in slots]
            else:
                vals = []
                for fld in fields:
                    nums = [float(t) for t in fld.split(",") if t.strip() not in ("def", "")]
                    vals.append(None if not nums else nums[0] if len(nums) == 1 else nums)
            for i, v in enumerate(vals[:len(want)]):
                if v is not None:
                    want[i][1] = v
            assert got == want, (name, case, got, want)
    for name, exc in table["unparseable"]:
        with pytest.raises(NotImplementedError):
            M.parse(name)
        assert exc == "NotImplementedError"


def test_lr_grid_decisions_match_reference_g23():
    """Phase 1 of the framework (framework/lr_grid_train.py:9-160) on 36 seeded accuracy tables (1-3 iterations per LR,
    ties, all-zero tables), three storage policies each, plus an interrupted run resumed from grid_checkpoint.pth: the
    build's lr_grid_single_task must take the reference's decisions — best LR, best (iteration-averaged) accuracy, the
    winning node directory, the directories left on disk, the checkpointed accuracy lists, the nodes trained again after
    the interruption (fixture G23: the reference's unchanged function over the same stand-in method)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g23_common as G
    from clsurvey_amd.framework import driver
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G23_lr_grid_decisions.json")) as f:
        ref = json.load(f)
    assert ref["lrs"] == G.LRS and ref["modes"] == G.MODES
    mine = json.loads(json.dumps(G.generate(driver.lr_grid_single_task)))          # tuples -> lists, as stored
    assert len(mine) == len(ref["tables"]) == 36
    for i, (a, b) in enumerate(zip(mine, ref["tables"])):
        assert a["acc"] == b["acc"] and a["iterations"] == b["iterations"]
        for mode in G.MODES:
            assert a["modes"][mode] == b["modes"][mode], (i, mode, a["modes"][mode], b["modes"][mode])
        assert a["resumed"] == b["resumed"], (i, a["resumed"], b["resumed"])


def test_stability_decay_matches_reference_g24():
    """Phase 2 of the framework (framework_train.py:76-166) through 24 scenarios — 1 / 2 / 3 hyper-parameters, a method with
    its own decay operator, thresholds met at once / after some decays / never — against the reference's unchanged
    HyperparameterFramework + Manager over the same stand-in method (fixture G24): the hyper-parameters of every training
    call, the framework state, hyperparams.pth.tar, SUCCESS.FLAG, the files left in TASK_TRAINING and the surviving model;
    a second run on the finished tree (skipped); a run that dies after k attempts (exit code 1, as the reference) and its
    continuation by fresh objects from the checkpoint."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g24_common as G
    from clsurvey_amd.framework import driver
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G24_stability_decay.json")) as f:
        ref = json.load(f)["tables"]
    mine = json.loads(json.dumps(G.generate(driver.HyperparameterFramework, driver.Manager)))
    assert len(mine) == len(ref) == 24
    for i, (a, b) in enumerate(zip(mine, ref)):
        assert a["scenario"] == b["scenario"]
        for key in ("fresh", "again", "interrupted", "resumed"):
            assert (key in a) == (key in b), (i, key)
            if key in a:
                assert a[key] == b[key], (i, key, a[key], b[key])


def test_eval_driver_and_names_match_reference_g25():
    """The evaluation driver (framework/eval.py:146-247) and the experiment / first-task-model names (utilities/utils.py
    get_exp_name, models/net.py get_init_modelname) against the reference's unchanged functions taken as data (fixture G25):
    which (task, model) pairs are evaluated with which paths and what the result files hold — whole sequence, task window,
    an evaluation failing on a later / the first model of a task, an existing result file with and without overwrite mode,
    debug mode — and the two name strings for six argument sets (weight decay, DROP / BN architectures, list-valued
    static hyper-parameters, a method without hyper-parameters)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g25_common as G
    from clsurvey_amd.framework import driver
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G25_eval_and_names.json")) as f:
        ref = json.load(f)
    assert G.names(driver.get_exp_name, driver.first_task_modelname) == ref["names"]
    from clsurvey_amd.methods import method as M
    assert G.adopt(M.parse("finetuning").grid_poststep) == ref["adopt"]          # the TASK_TRAINING link of the grid-only methods
    assert G.adopt(M.parse("meanIMM").grid_poststep) == ref["adopt"]
    mine = json.loads(json.dumps(G.evals(driver.eval_all_models_all_tasks, driver.get_perf_output_filename)))
    for a, b in zip(mine, ref["evals"]):
        assert a["tag"] == b["tag"]
        assert a["calls"] == b["calls"], (a["tag"], a["calls"], b["calls"])
        assert a["files"] == b["files"], (a["tag"], a["files"], b["files"])


def test_single_task_orchestration_matches_reference_g26():
    """framework_single_task (framework_train.py:219-292) with its two phases, through 24 scenarios (four hook sets x a later
    task, PackNet's storage policy, --save_models_FT_heuristic, a first task that is skipped / trained / wrapped), over a
    stand-in method that logs every hook call with the args / manager fields it sees — against the reference's unchanged
    function and Manager (fixture G26): same hooks in the same order seeing the same learning rate, storage policy,
    reg_sets, head index and directories, same model handed to the next task.
    One deliberate difference, asserted: wrapping the first task's model without an init_next_task hook ends in an
    AttributeError in the reference (manager.best_model_path is read before anything set it, framework_train.py:282); the
    build points it at task 1's TASK_TRAINING slot, which is what GEM.poststep (method.py:301-318) then writes."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g26_common as G
    from clsurvey_amd.framework import driver
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G26_single_task_trace.json")) as f:
        ref = json.load(f)["tables"]
    mine = json.loads(json.dumps(G.generate(driver.framework_single_task, driver.Manager)))
    assert len(mine) == len(ref) == 24
    repaired = 0
    for i, (a, b) in enumerate(zip(mine, ref)):
        assert a["scenario"] == b["scenario"]
        if b["ended"] == "AttributeError":
            repaired += 1
            assert b["scenario"]["wrap_first_task_model"] and b["scenario"]["task"] == 1
            assert a["ended"] == "returned" and [r["hook"] for r in a["log"]] == [r["hook"] for r in b["log"]]
            assert a["previous_task_model_path"] == os.path.join("task_1", "TASK_TRAINING", "best_model.pth.tar")
            continue
        if b["scenario"]["wrap_first_task_model"]:
            # (with an init_next_task hook the reference gets through; its poststep still sees no best_model_path)
            for rows in (a["log"], b["log"]):
                for r in rows:
                    r.pop("best_model_path"), r.pop("heuristic_exp_dir")
        assert a == b, (i, a, b)
    assert repaired == 3


def test_epoch_loops_match_reference_g27():
    """The shared epoch loop (methods/train_common.train_model) in the four configurations the methods use it in, against
    the reference's four train_model variants and LwF's train_model_lwf run on a scripted network (fixture G27): best accuracy returned, epochs run,
    learning rate per epoch (x0.1 after five epochs without a new best, stop after more than ten — ten for SI, which also
    runs one epoch more), NaN-loss abort (not in plain SGD), checkpoint every saving_freq epochs and what it holds, the
    best model's epoch, resume from epoch.pth.tar with fresh objects, save_models_mode off.  The engine is a stand-in that
    does what NetEngine.loss_step does to the loop: forward, batch-mean loss and hit count into the stats buffer."""
    import json
    import sys
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g27_common as G
    from clsurvey_amd.methods import train_common as tc
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G27_epoch_loops.json")) as f:
        ref = json.load(f)["runs"]

    class Engine:
        device = torch.device("cpu")

        def __init__(self, model):
            self.model = model

        def loss_step(self, x, y, kind, backward=True, stats=None):
            assert kind == "ce_mean"
            out = self.model(x)
            loss = F.cross_entropy(out, y)
            if backward:
                self.model.zero_grad()
                loss.backward()
            stats[0] += float(loss.detach())
            stats[1] += int((out.argmax(1) == y).sum())

    class LwfEngine(Engine):          # LwfEngine.step: task CE + hits into stats, backward of the whole objective
        def step(self, x, y, teacher_logits, T, reg_lambda, backward=True, stats=None):
            out = self.model(x)[-1]
            loss = F.cross_entropy(out, y)
            if backward:
                assert teacher_logits is not None and teacher_logits.shape == (x.shape[0], G.C)
                self.model.zero_grad()
                loss.backward()
            stats[0] += float(loss.detach())
            stats[1] += int((out.argmax(1) == y).sum())

    class Teacher:
        @staticmethod
        def logits(x):
            return torch.zeros(x.shape[0], G.C)

    def train(variant, model, opt, lr, loaders, sizes, num_epochs, exp_dir, resume, saving_freq, save_models_mode):
        if variant == "lwf":          # methods/lwf.py: fine_tune_SGD_LwF -> train_model_lwf
            from clsurvey_amd.methods import lwf
            return lwf.train_model_lwf(model, G.ScriptedTeacher(), opt, lr, loaders, sizes, num_epochs, exp_dir, resume,
                                       temperature=2, saving_freq=saving_freq, reg_lambda=1, engine=LwfEngine(model), teacher=Teacher)
        eng = Engine(model)
        if variant == "sgd":          # methods/finetune.py:59-61
            return tc.train_model(model, eng, opt, lr, loaders, sizes, num_epochs, exp_dir, resume, saving_freq=saving_freq,
                                  step_fn=opt.step, save_models_mode=save_models_mode, abort_on_bad_loss=False)
        if variant == "ebll":         # methods/ebll.py: fine_tune_SGD_EBLL -> train_model_ebll
            from clsurvey_amd.methods import ebll

            class EbllEngine(Engine):
                def step(self, x, y, target_logits, target_codes, T, reg_lambda, reg_alpha, backward=True, stats=None):
                    out = self.model(x)[0][-1]
                    loss = F.cross_entropy(out, y)
                    if backward:
                        assert target_logits is not None and target_codes is not None
                        self.model.zero_grad()
                        loss.backward()
                    stats[0] += float(loss.detach())
                    stats[1] += int((out.argmax(1) == y).sum())
                    return loss.detach(), torch.zeros(())

            class EbllTeacher:
                @staticmethod
                def targets(x):
                    return torch.zeros(x.shape[0], G.C), torch.zeros(x.shape[0], 3)

            return ebll.train_model_ebll(model, G.ScriptedEbllTeacher(), opt, lr, loaders, sizes, num_epochs, exp_dir, resume,
                                         temperature=2, reg_alpha=1e-6, saving_freq=saving_freq, reg_lambda=1,
                                         engine=EbllEngine(model), teacher=EbllTeacher)
        if variant == "imm":          # methods/imm.py:66-67
            return tc.train_model(model, eng, opt, lr, loaders, sizes, num_epochs, exp_dir, resume, saving_freq=saving_freq,
                                  abort_on_bad_loss=False)
        if variant == "si":           # methods/si.py:79-80
            return tc.train_model(model, eng, opt, lr, loaders, sizes, num_epochs, exp_dir, resume, saving_freq=saving_freq,
                                  early_stop="ge", extra_epoch=True)
        return tc.train_model(model, eng, opt, lr, loaders, sizes, num_epochs, exp_dir, resume, saving_freq=saving_freq)   # ewc.py:121, mas.py:96

    mine = json.loads(json.dumps(G.generate(train)))
    assert len(mine) == len(ref) == 36
    for a, b in zip(mine, ref):
        assert (a["tag"], a["variant"]) == (b["tag"], b["variant"])
        assert a == b, (a["tag"], a["variant"], a, b)


def test_trainer_calls_match_reference_g28():
    """What every method object hands to its trainers (the ARGUMENT MAPs of clsurvey_amd/methods/method.py) against the
    reference's methods/method.py taken as data (fixture G28): with all trainer entry points replaced by recorders, the hooks
    the framework calls run for tasks 1 and 2 of EWC, MAS, SI, LWF, EBLL, IMM, PackNet, HAT, GEM and plain finetuning over one
    fixed (args, manager) pair; per hook the same trainers must be called in the same order with the same effective
    arguments (call bound to the trainer's signature, defaults filled in; the dicts of the packnet / HAT / rehearsal mains
    item by item), and leave the same args / manager fields behind."""
    import contextlib
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g28_common as G
    from clsurvey_amd.framework import driver
    from clsurvey_amd.methods import method as M
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G28_trainer_calls.json")) as f:
        ref = json.load(f)["hooks"]
    phase2_label = {"EWC": "fine_tune_EWC_acuumelation", "MAS": "fine_tune_objective_based_acuumelation", "SI": "fine_tune_elastic",
                    "LWF": "fine_tune_SGD_LwF", "EBLL": "fine_tune_SGD_EBLL"}

    @contextlib.contextmanager
    def patches(log):
        saved2, saved1 = dict(M.PHASE2), dict(M.PHASE1)
        mains = {"_packnet": ("packnet.main", 0.5), "_hat": ("hat.main", (None, 0.5)), "_gem": ("rehearsal.main", (None, 0.5))}
        saved_mains = {k: getattr(M, k).main for k in mains}
        saved_ft, saved_compose = M._ft.fine_tune_SGD, M.compose_dataset
        try:
            for k, (fn, amap) in saved2.items():
                M.PHASE2[k] = (G.Recorder(log, phase2_label[k], fn, (None, 0.5)), amap)
            fn, amap = saved1["l2transfer"]
            M.PHASE1["l2transfer"] = (G.Recorder(log, "fine_tune_l2transfer", fn, (None, 0.5)), amap)
            for k, (label, res) in mains.items():
                getattr(M, k).main = G.Recorder(log, label, None, res)
            M._ft.fine_tune_SGD = G.Recorder(log, "fine_tune_SGD", saved_ft, (None, 0.5))
            M.compose_dataset = G.Recorder(log, "compose_dataset", None, ("<loaders>", "<sizes>", "<classes>"))
            yield
        finally:
            M.PHASE2.clear(), M.PHASE2.update(saved2), M.PHASE1.clear(), M.PHASE1.update(saved1)
            for k, v in saved_mains.items():
                getattr(M, k).main = v
            M._ft.fine_tune_SGD, M.compose_dataset = saved_ft, saved_compose

    exists = {G.ROOT + "/parent/task_2/TASK_TRAINING/best_model_PRUNED_final.pth.tar"}
    mine = json.loads(json.dumps(G.run(M.parse, driver.Manager, patches, exists)))
    assert list(mine) == list(ref)
    # arguments that exist on one side only: the reference's GPU switches / worker counts, the build's device plumbing
    ref_only = {"use_gpu", "num_workers", "print_freq"}
    mine_only = {"device", "engine_params", "batch_size", "cache"}
    for key in ref:
        a, b = mine[key], ref[key]
        assert a["ended"] == b["ended"], (key, a["ended"], b["ended"])
        assert [c["callee"] for c in a["calls"]] == [c["callee"] for c in b["calls"]], key
        for ca, cb in zip(a["calls"], b["calls"]):
            xa, xb = ca["arguments"], cb["arguments"]
            if "args" in xb:                                     # a main(overwrite_args [, nc_per_task]) or compose_dataset
                pa, pb = xa["args"], xb["args"]
                if ca["callee"] == "compose_dataset":
                    assert pa[:2] == pb[:2], (key, pa, pb)
                    continue
                da, db = dict(map(tuple, map(lambda kv: (kv[0], json.dumps(kv[1])), pa[0]))), dict(map(tuple, map(lambda kv: (kv[0], json.dumps(kv[1])), pb[0])))
                assert set(db) <= set(da), (key, ca["callee"], sorted(set(db) - set(da)))
                for k in db:
                    assert da[k] == db[k], (key, ca["callee"], k, da[k], db[k])
                assert pa[1:] == pb[1:], (key, ca["callee"], "positional")
                continue
            assert set(xb) - set(xa) <= ref_only, (key, ca["callee"], sorted(set(xb) - set(xa)))
            assert set(xa) - set(xb) <= mine_only, (key, ca["callee"], sorted(set(xa) - set(xb)))
            for k in set(xa) & set(xb):
                assert xa[k] == xb[k], (key, ca["callee"], k, xa[k], xb[k])
        assert a["args"] == b["args"], (key, a["args"], b["args"])
        assert a["manager"] == b["manager"], (key, a["manager"], b["manager"])


def test_hat_trainer_loops_match_reference_g29():
    """HatTrainer.train (methods/hat_main.py) against the epoch loops of the reference's two HAT trainers (approaches/hat.py
    joint training, hat_finetune.py phase-1 search) with train_epoch / eval replaced by table look-ups on both sides
    (fixture G29, 24 scenarios: first / later task, warm-up on / off, plateau / mixed / rising accuracies, resume): learning
    rate and lambda of every epoch (warm-up LR and lambda 0 for the first 11 epochs of a first task, the decay that follows
    a warm-up starting from the warm-up LR — the reference's arithmetic, kept), epochs run (patience 30, LR / 2 at 15 left,
    suspended below nepochs / 2 on the first task), best accuracy, checkpoint contents, the model kept.
    One asserted difference: the reference's phase-1 trainer cannot resume — it formats chkpt['warmup'], a key its own
    checkpoint does not have (hat_finetune.py:54) — the build's continues from the checkpoint."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g29_common as G
    from clsurvey_amd.methods import hat_main as HM
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G29_hat_trainer_loops.json")) as f:
        ref = json.load(f)["runs"]

    class Scripted(HM.HatTrainer):
        def __init__(self, model, exp_dir, args, joint):          # HatTrainer.__init__ minus the HatEngine (no device here)
            self.model, self.exp_dir, self.joint = model, exp_dir, joint
            self.sbatch, self.base_lr = args.batch_size, args.lr
            self.save_freq, self.weight_decay = args.save_freq, args.weight_decay
            self.smax, self.post_lamb = args.parameter
            self.lamb = None
            self.nepochs = args.nepochs + (0 if joint else HM.WARMUP_EPOCHS)
            self.optimizer = None
            self.mask_pre, self.mask_back = None, {}

        def init_masks(self, current_task, smax):
            return None, {}

    def make_trainer(joint, model, exp_dir, nepochs, args):
        args.nepochs = nepochs
        return Scripted(model, exp_dir, args, joint)

    mine = json.loads(json.dumps(G.generate(make_trainer)))
    assert len(mine) == len(ref) == 24
    unresumable = 0
    for a, b in zip(mine, ref):
        assert a["scenario"] == b["scenario"]
        if "raises" in b["run"]:
            unresumable += 1
            assert not b["scenario"]["joint"] and b["scenario"]["resume_after"] and b["run"]["raises"] == "KeyError"
            assert a["first"] == b["first"]
            r = a["run"]                                           # continues: first logged epoch follows the checkpoint's
            assert r["epochs"][0]["epoch"] == a["first"]["files"]["epoch.pth.tar"]["e"] + 1 and r["best_acc"] >= a["first"]["best_acc"]
            continue
        assert a == b, (a["scenario"], a, b)
    assert unresumable == 2


def test_packnet_session_loops_match_reference_g30():
    """packnet_main.Manager.train / prune against the reference's Manager (methods/packnet/main.py:234-339) with do_epoch /
    eval as table look-ups and the pruner a logger on both sides (fixture G30): learning rate per epoch (x0.1 in the epoch
    that follows five strictly worse validations — every epoch restarts from args.lr —, stop after more than ten), epoch
    numbering of a resumed session, what train() returns, the order pre-prune eval -> prune -> check -> post-prune eval ->
    retraining -> check, and every file with its fields (best / epoch checkpoints carry the counters the session STARTED
    with, as the reference's do; the .json error history)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g30_common as G
    from types import SimpleNamespace
    from clsurvey_amd.methods import packnet_main as PM
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G30_packnet_loops.json")) as f:
        ref = json.load(f)["runs"]

    def extra(m):        # what the build's Manager has besides the common fields: the plan executor (its arena feeds prune()'s optimizer)
        return {"_mode": lambda training: None, "_mask_arena": None,
                "engine": SimpleNamespace(arena=SimpleNamespace(params=list(m.model.parameters())))}

    mine = json.loads(json.dumps(G.generate(PM.Manager, extra, lambda m: torch.optim.SGD(m.model.parameters(), lr=m.args.lr, momentum=0.9))))
    assert [r["tag"] for r in mine] == [r["tag"] for r in ref]
    for a, b in zip(mine, ref):
        assert a == b, (a["tag"], a, b)


def test_rehearsal_epoch_loop_matches_reference_g31():
    """gem_main.train_model against the reference's rehearsal trainer loop (methods/rehearsal/train_rehearsal.py:57-199) over a
    scripted GEM wrapper (fixture G31), observe and observe_FT modes: learning rate per epoch, stop after more than ten
    epochs without a new best, exit on a NaN loss, the best model written when the loop ends (also with save_models_mode
    off), checkpoint every saving_freq epochs with its fields, resume by fresh objects."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g31_common as G
    from clsurvey_amd.methods import gem_main as GM
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G31_rehearsal_loop.json")) as f:
        ref = json.load(f)["runs"]
    mine = json.loads(json.dumps(G.generate(GM.train_model, with_paths=False)))
    assert len(mine) == len(ref) == 10
    for a, b in zip(mine, ref):
        assert a == b, (a["tag"], a["finetune"], a, b)


def test_ebll_autoencoder_grid_matches_reference_g32():
    """EBLL.prestep (the autoencoder grid on the previous task, method.py:835-908) against the reference's unchanged method over
    a stand-in autoencoder trainer (fixture G32), six accuracy tables: which nodes are trained with which arguments and in
    which directories, which directories are kept, the model path handed to phase 2, the grid checkpoint — fresh, again on
    the finished tree (nothing retrained), interrupted after two nodes, continued."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import g32_common as G
    from clsurvey_amd.methods import ebll as E
    from clsurvey_amd.methods import method as M
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G32_ebll_autoencoder_grid.json")) as f:
        ref = json.load(f)["tables"]

    def install(trainer):
        saved = E.fine_tune_Adam_Autoencoder
        E.fine_tune_Adam_Autoencoder = trainer
        return lambda: setattr(E, "fine_tune_Adam_Autoencoder", saved)

    mine = json.loads(json.dumps(G.generate(lambda: M.parse("EBLL"), install)))
    assert list(mine) == list(ref)
    for tag in ref:
        for stage in ("fresh", "again", "interrupted", "continued"):
            assert mine[tag][stage] == ref[tag][stage], (tag, stage, mine[tag][stage], ref[tag][stage])


def test_model_names_match_reference_factory_g34():
    """The regularised / deep model names (models/net.py:15-36, :133-175, VGGSlim.py:27-76): module tree, parameter order,
    head-surgery index and initialisation of the build's `models.parse_model_name` / `driver.BaseModel` against what the
    reference's own factory created (G34, tests/golden/make_g34.py)."""
    import numpy as np
    import tempfile
    from clsurvey_amd import models
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G34_model_names.npz"))
    names = sorted({k.split("__")[0] for k in g.files})
    assert names == sorted(["small_VGG9_cl_128_128_BN", "small_VGG9_cl_128_128_DROP", "small_VGG9_cl_128_128_DROP_BN",
                            "deep_VGG22_cl_512_512"])
    for name in names:
        torch.manual_seed(1)
        m = models.parse_model_name(name, (32, 32), 20)
        tree = ["%s:%s" % (n, type(mod).__name__) for n, mod in m.named_modules() if n]
        assert tree == [str(t) for t in g[name + "__tree"]], name
        assert [n for n, _ in m.named_parameters()] == [str(t) for t in g[name + "__param_names"]]
        with tempfile.TemporaryDirectory() as root:
            assert driver.BaseModel(root, name, (32, 32), 20).last_layer_idx == int(g[name + "__last_layer_idx"][0])
        for (pn, p), (mean, std, lo, hi) in zip(m.named_parameters(), g[name + "__init_stats"]):
            if std == 0.0:                                  # biases 0, BatchNorm weight 1: exact
                assert float(p.min()) == lo == float(p.max()) == hi, (name, pn)
            else:                                           # Kaiming fan-out / N(0, 0.01): same distribution, other draw
                assert abs(float(p.std()) - std) <= 0.1 * std and abs(float(p.mean())) <= 4 * std / p.numel() ** 0.5 + 1e-12, (name, pn)
        assert [type(mod).__name__ for mod in m.modules() if isinstance(mod, torch.nn.Dropout)] == ["Dropout"] * sum(
            1 for k in g.files if k.startswith(name + "__dropmask"))


def test_bench_last_line_is_compact_and_parseable():
    """The driver parses the LAST stdout line of bench.py and keeps 8 KB of tail: round 4's 23.7 KB line was lost.  The line
    built from the largest full record on file (profiles/r04_final_bench.json) must stay under bench.LINE_LIMIT, parse, and
    carry the contract's keys + roofline + cpu_baseline; a record ten times larger must still fit (optional parts dropped)."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(root, "profiles", "r04_final_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, "gpurun_out/bench_details.json")
    assert "\n" not in line and len(line) <= bench.LINE_LIMIT < 8192
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["value"] == float("%.6g" % full["value"]) and rec["config"]["workload"] == full["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert "per_layer" not in rec["roofline"] and "configs" not in rec and "sweep" not in rec
    # a bloated record: optional objects go, the contract stays
    full["grid"] = {"nodes": 8, "collectives_in_timed_region": ["x" * 3000] * 4, "fill_factor": 0.625}
    full["roofline"]["avg_launch_how"] = "y" * 5000
    line = bench.compact_line(full, None)
    assert len(line) <= bench.LINE_LIMIT
    rec = json.loads(line)
    assert rec["value"] == float("%.6g" % full["value"]) and "roofline" in rec and "cpu_baseline" in rec
