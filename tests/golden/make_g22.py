"""G22: the plugin surface of the reference's methods/method.py as DATA — for every --method_name on this path the parsed
object's class, name / eval_name / category / extra_hyperparams_count, its hyper-parameter dict(s) in order, its plain
instance attributes, the names of the hooks the framework driver probes for (train, grid_train, grid_prestep, ...), and
what set_hyperparams() makes of a list of override strings (or the exception type it stops with).  Dev container only:
    python tests/golden/make_g22.py
writes tests/golden/G22_method_table.json.  No reference source is stored: attribute values and names only."""
import copy
import json
import os
import sys
from collections import OrderedDict

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "harness"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import harness  # noqa: E402

harness.install()

NAMES = ["EWC", "MAS", "SI", "EBLL", "LWF", "GEM", "packnet", "HAT", "finetuning", "meanIMM", "modeIMM", "mean_IMM", "mode_IMM"]
HOOKS = ["train", "grid_train", "grid_prestep", "grid_poststep", "grid_datafetch", "prestep", "poststep", "train_args_overwrite",
         "train_init", "init_next_task", "get_output", "inference_eval", "eval_model_preprocessing", "set_mode", "get_dataset_name"]
OVERRIDES = ["0.5", "0.5,300", "def,7", "0.25,def", "0.1,0.2;5.2,300", "def;1,2", "3;", ""]


def plain(v):
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (dict, OrderedDict)):
        return [[str(k), plain(x)] for k, x in v.items()]
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    return "<%s>" % type(v).__name__


def describe(m):
    cat = m.category
    return {"class": type(m).__name__, "name": m.name, "eval_name": m.eval_name,
            "category": getattr(cat, "name", str(cat)), "extra_hyperparams_count": m.extra_hyperparams_count,
            "hyperparams": plain(m.hyperparams),
            "static_hyperparams": plain(getattr(m, "static_hyperparams", None)),
            "instance_attrs": {k: plain(v) for k, v in sorted(vars(m).items())},
            # class-level switches the drivers read with hasattr / getattr (start_scratch, wrap_first_task_model, no_framework,
            # grid_chkpt, ...): every public attribute of a plain type that is not one of the fields above
            "flags": {k: getattr(m, k) for k in sorted(dir(m))
                      if not k.startswith("_") and k not in ("name", "eval_name", "category", "extra_hyperparams_count", "hyperparams",
                                                             "static_hyperparams") and k not in vars(m)
                      and isinstance(getattr(m, k), (bool, int, float, str))},
            "hooks": [h for h in HOOKS if callable(getattr(m, h, None))]}


if __name__ == "__main__":
    import methods.method as RM
    # the hyper-parameter dicts are CLASS attributes in the reference (one dict per class, shared by its instances):
    # set_hyperparams() mutates them, so every experiment below starts from a restored copy
    pristine = {}
    for name in NAMES:
        cls = type(RM.parse(name))
        pristine[cls] = {k: copy.deepcopy(getattr(cls, k)) for k in ("hyperparams", "static_hyperparams") if isinstance(getattr(cls, k, None), dict)}

    def fresh(name):
        cls = type(RM.parse(name))
        for k, v in pristine[cls].items():
            setattr(cls, k, copy.deepcopy(v))
        return RM.parse(name)

    table = OrderedDict()
    for name in NAMES:
        entry = describe(fresh(name))
        entry["hyperparams_shared_by_class"] = bool(pristine[type(fresh(name))]) and "hyperparams" not in vars(fresh(name))
        entry["overrides"] = []
        for static in (False, True):
            for text in OVERRIDES:
                mm = fresh(name)
                if static and not hasattr(mm, "static_hyperparams"):
                    continue
                try:
                    RM.set_hyperparams(mm, text, static_params=static)
                    after = plain(mm.static_hyperparams if static else mm.hyperparams)
                    init = plain(getattr(mm, "init_hyperparams", None))
                    entry["overrides"].append({"text": text, "static": static, "result": after, "init_hyperparams": init})
                except Exception as e:      # the reference stops on some strings: keep the exception type as the observed behaviour
                    entry["overrides"].append({"text": text, "static": static, "raises": type(e).__name__})
        fresh(name)
        table[name] = entry
    unparseable = []
    for name in ["nonsense", "ewc", ""]:
        try:
            RM.parse(name)
        except Exception as e:
            unparseable.append([name, type(e).__name__])
    out = {"methods": table, "unparseable": unparseable, "hooks_probed": HOOKS, "override_strings": OVERRIDES}
    path = os.path.join(HERE, "G22_method_table.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=False)
    print("wrote", path, os.path.getsize(path), "bytes")
