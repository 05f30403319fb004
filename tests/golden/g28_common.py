"""Shared by the G28 generator (reference run, dev container) and the CPU test of the build's method table: for every method
on the path, the framework-facing hooks are called with one fixed (args, manager) pair while every trainer entry point is
replaced by a recorder, and what each hook hands to which trainer — and what it changes on args / manager — is written
down.  Which attributes get patched on which side is the caller's business (`patches`); nothing here imports either."""
import collections
import inspect
import os
from types import SimpleNamespace

ROOT = "/exp"
METHODS = ["EWC", "MAS", "SI", "LWF", "EBLL", "meanIMM", "packnet", "HAT", "GEM", "finetuning"]


class DS:
    name = "standin_ds"
    task_count = 3
    classes_per_task = collections.OrderedDict([("1", list("abcd")), ("2", list("efghi")), ("3", list("jklmno"))])

    def get_taskname(self, i):
        return str(i)

    def get_task_dataset_path(self, task_name=None, rnd_transform=False):
        return "%s/data/%s.pth" % (ROOT, task_name)


def make_args(task):
    return SimpleNamespace(task_counter=task, task_name=str(task), weight_decay=1e-4, num_epochs=33, batch_size=64, lr=2e-3,
                           saving_freq=7, model_name="small_VGG9_cl_128_128", classifier_heads_starting_idx=6, init_model_path=None,
                           data_dir=None, train_bn=False, previous_task_dataset_path="%s/data/%d.pth" % (ROOT, task - 1),
                           previous_task_model_path="%s/prev_args.pth" % ROOT, presteps_elapsed_time=0, postprocess_time=0,
                           device="cuda")


def make_manager(Manager, method, task):
    base = SimpleNamespace(name="small_VGG9_cl_128_128", last_layer_idx=6, path="%s/models/base.pth" % ROOT)
    m = Manager(DS(), method, "%s/prev.pth" % ROOT, "%s/parent" % ROOT, base)
    t = "%s/parent/task_%d" % (ROOT, task)
    m.current_task_dataset_path = "%s/data/%d.pth" % (ROOT, task)
    m.ft_parent_exp_dir = t + "/FT_LR_GRIDSEARCH"
    m.gridsearch_exp_dir = t + "/FT_LR_GRIDSEARCH/lr=5.0E-03"
    m.best_exp_grid_node_dirname = t + "/FT_LR_GRIDSEARCH/lr=1.0E-03"
    m.heuristic_exp_dir = t + "/TASK_TRAINING"
    m.reg_sets = ["%s/data/%d.pth" % (ROOT, task - 1)]
    m.autoencoder_model_path = "%s/parent/task_%d/ENCODER_TRAINING/dim=100/best_model.pth.tar" % (ROOT, task - 1)
    m.best_finetuned_model_path = t + "/FT_LR_GRIDSEARCH/lr=1.0E-03/best_model.pth.tar"
    m.best_model_path = t + "/TASK_TRAINING/best_model.pth.tar"
    return m


def _plain(v):
    if isinstance(v, DS):
        return "<dataset>"
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, dict):
        return [[str(k), _plain(x)] for k, x in v.items()]
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return "<%s>" % type(v).__name__


class Recorder:
    """stands in for one trainer entry point; canonical record = the call bound to the ORIGINAL's signature with its
    defaults filled in (what the trainer would actually run with), or the argument dict of a main(overwrite_args)"""

    def __init__(self, log, label, original, result):
        self.log, self.label, self.original, self.result = log, label, original, result

    def __call__(self, *a, **kw):
        if self.original is None:
            bound = {"args": a, "kwargs": kw}
        else:
            b = inspect.signature(self.original).bind(*a, **kw)
            b.apply_defaults()
            bound = dict(b.arguments)
        self.log.append({"callee": self.label, "arguments": {k: _plain(v) for k, v in bound.items()}})
        return self.result


def _fields(obj, keys):
    return {k: _plain(getattr(obj, k, "<unset>")) for k in keys}


ARG_KEYS = ["lr", "init_model_path", "train_bn", "presteps_elapsed_time"]
MGR_KEYS = ["previous_task_model_path", "best_finetuned_model_path", "autoencoder_model_path", "dataset_name", "disable_pruning_mask",
            "best_model_path"]


def run(parse, Manager, patches, exists=()):
    """parse: name -> method object.  patches(log): context manager factory that installs the recorders.  exists: paths that
    os.path.exists should report as present."""
    out = collections.OrderedDict()
    real_exists = os.path.exists
    for name in METHODS:
        for task in (1, 2):
            method = parse(name)
            args, mgr = make_args(task), make_manager(Manager, method, task)
            log = []
            os.path.exists = lambda p, _r=real_exists: (p in exists) or (not str(p).startswith(ROOT) and _r(p))
            try:
                with patches(log):
                    if hasattr(method, "train_args_overwrite"):
                        method.train_args_overwrite(args)
                    for hook in ("grid_prestep", "grid_train", "grid_poststep", "train_init", "train", "poststep"):
                        fn = getattr(method, hook, None)
                        if fn is None:
                            continue
                        mark = len(log)
                        try:
                            if hook == "grid_train":
                                fn(args, mgr, 5e-3)
                            elif hook == "train":
                                fn(args, mgr, method.hyperparams)
                            else:
                                fn(args, mgr)
                            ended = "returned"
                        except Exception as e:          # some hooks touch the file system (symlinks): the type is the record
                            ended = type(e).__name__
                        out["%s/task%d/%s" % (name, task, hook)] = {"calls": log[mark:], "ended": ended, "args": _fields(args, ARG_KEYS),
                                                                    "manager": _fields(mgr, MGR_KEYS)}
            finally:
                os.path.exists = real_exists
    return out
