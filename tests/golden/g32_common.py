"""Shared by the G32 generator (reference run, dev container) and the CPU test of the build's EBLL prestep (the autoencoder
grid, methods/method.py:842-908): accuracy tables for the (dim, alpha, lr) nodes, the stand-in autoencoder trainer, and the
routine that runs ONE implementation's prestep over them — fresh, again on the finished tree, and after an interruption."""
import os
import shutil
import tempfile
from types import SimpleNamespace

import torch

NODES = [(100, 0.1), (100, 0.01), (300, 0.1), (300, 0.01)]          # itertools.product(encoder_dims, encoder_alphas) x lr 0.01
TABLES = {
    "second_best": [0.55, 0.70, 0.62, 0.41],
    "first_best": [0.90, 0.70, 0.62, 0.41],
    "last_best": [0.45, 0.50, 0.62, 0.93],
    "tie": [0.60, 0.60, 0.60, 0.60],
    "all_weak": [0.10, 0.30, 0.20, 0.05],
    "rising": [0.41, 0.52, 0.63, 0.74],
}


class Interrupted(Exception):
    pass


def make_trainer(table, calls, fail_after=None):
    def fine_tune_Adam_Autoencoder(dataset_path, previous_task_model_path, exp_dir, batch_size, num_epochs, lr, alpha, last_layer_name,
                                   auto_dim, **kw):
        if fail_after is not None and len(calls) >= fail_after:
            raise Interrupted()
        calls.append([auto_dim, alpha, lr, os.path.basename(exp_dir), dataset_path, previous_task_model_path, batch_size, num_epochs,
                      last_layer_name])
        torch.save({"dim": auto_dim, "alpha": alpha}, os.path.join(exp_dir, "best_model.pth.tar"))
        return None, table[NODES.index((auto_dim, alpha))]
    return fine_tune_Adam_Autoencoder


def _snapshot(root, mgr, calls):
    enc = os.path.join(root, "task_1", "ENCODER_TRAINING")
    ck = torch.load(os.path.join(enc, "grid_checkpoint.pth"), weights_only=False) if os.path.exists(os.path.join(enc, "grid_checkpoint.pth")) else {}
    path = getattr(mgr, "autoencoder_model_path", None)
    return {"calls": list(calls), "dirs": sorted(os.listdir(enc)) if os.path.isdir(enc) else None,
            "autoencoder_model_path": os.path.relpath(path, root) if path else None,
            "kept": torch.load(path, weights_only=False) if path and os.path.exists(path) else None,
            "checkpoint": sorted([[str(k), v if not isinstance(v, tuple) else list(v)] for k, v in ck.items()])}


def generate(parse_ebll, install):
    """install(trainer) replaces the implementation's autoencoder trainer entry and returns a function that undoes it."""
    out = {}
    for tag, table in TABLES.items():
        root = tempfile.mkdtemp()
        try:
            def once(fail_after=None):
                calls = []
                undo = install(make_trainer(table, calls, fail_after))
                method = parse_ebll()
                args = SimpleNamespace(task_counter=2, previous_task_dataset_path="data/1.pth", batch_size=8, classifier_heads_starting_idx=4,
                                       presteps_elapsed_time=0)
                mgr = SimpleNamespace(parent_exp_dir=root, previous_task_model_path="models/prev.pth")
                try:
                    method.prestep(args, mgr)
                    ended = "returned"
                except Interrupted:
                    ended = "interrupted"
                finally:
                    undo()
                return dict(_snapshot(root, mgr, calls), ended=ended)
            entry = {"fresh": once(), "again": once()}
            shutil.rmtree(root)
            os.makedirs(root)
            entry["interrupted"] = once(fail_after=2)
            entry["continued"] = once()
            out[tag] = entry
        finally:
            shutil.rmtree(root, ignore_errors=True)
    return out
