"""Grid sharding of the Continual Hyperparameter Framework over the GPUs of one node.

What shards: the phase-1 LR grid — `for lr in args.lrs` x finetune_iterations
(framework/lr_grid_train.py:51,60) are independent trainings from the same start model, seeded by
the iteration index only — and, optionally, the phase-2 stability-decay attempts run speculatively
(attempt k uses lambda * decay^k; the smallest k meeting acc >= A_ft(1-p) wins, the same decision
rule as framework_train.py:100-136).  What does not: the task order and a single training run.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  Collectives carry models and metrics only:
  * broadcast of the flat parameter arena of the start / winning model (2.4-36 MB for the VGG9s)
  * all_gather of (node index, accuracy) scalars
There is no collective on the per-batch data path.
"""
import copy
import os

import torch
import torch.distributed as dist


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _dev():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def assign(n_nodes, world):
    """node i -> rank i % world (round-robin keeps the expensive low-LR / high-LR nodes apart)."""
    return [i % world for i in range(n_nodes)]


def gather_scalars(values):
    """values: {node_index: float} computed on this rank -> merged dict on every rank."""
    rank, world = rank_world()
    if world == 1:
        return dict(values)
    n = torch.tensor([len(values)], dtype=torch.int64, device=_dev())
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    m = max(int(c.item()) for c in counts)
    buf = torch.full((max(m, 1), 2), -1.0, dtype=torch.float64, device=_dev())
    for j, (k, v) in enumerate(sorted(values.items())):
        buf[j, 0], buf[j, 1] = float(k), float(v)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = {}
    for b in bufs:
        for k, v in b.cpu().tolist():
            if k >= 0:
                out[int(k)] = v
    return out


def broadcast_model(model, src=0):
    """Broadcast all parameters of `model` from rank src as ONE flat tensor (one RCCL broadcast)."""
    rank, world = rank_world()
    if world == 1:
        return model
    params = [p.data for p in model.parameters()]
    flat = torch.cat([p.reshape(-1).to(_dev(), torch.float32) for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    for p in params:
        n = p.numel()
        p.copy_(flat[off:off + n].view(p.shape).to(p.device))
        off += n
    return model


def sharded_grid_factory(node_dir_fn=None):
    """train_node factory for driver.main(): every rank trains the grid nodes assigned to it, the
    accuracies are all-gathered, and each rank then replays the reference's sequential
    selection rule over the complete table (identical decision on every rank)."""
    from . import driver

    def factory(args, manager):
        rank, world = rank_world()
        nodes = [(lr, it) for lr in args.lrs for it in range(args.finetune_iterations)]
        owner = assign(len(nodes), world)
        table = {}

        def run_all():
            mine = {}
            for i, (lr, it) in enumerate(nodes):
                if owner[i] != rank:
                    continue
                driver.set_random(it)
                d = "lr=" + driver.float_to_scientific_str(lr) + ("_it%d" % it if args.finetune_iterations > 1 else "")
                manager.gridsearch_exp_dir = os.path.join(manager.ft_parent_exp_dir, d)
                os.makedirs(manager.gridsearch_exp_dir, exist_ok=True)
                _, acc = manager.method.grid_train(args, manager, lr)
                mine[i] = acc
            table.update(gather_scalars(mine))

        def train_node(lr, it):
            if not table:
                run_all()
            return table[nodes.index((lr, it))]
        return train_node
    return factory


def speculative_decay(hf, args, manager, finetune_acc, max_parallel=None):
    """Run stability-decay attempts k = attempts .. attempts+world-1 concurrently (one per rank) and
    accept the smallest k with acc >= threshold.  Returns (k_accepted or None, {k: acc})."""
    rank, world = rank_world()
    lam0 = copy.deepcopy(dict(hf.hyperparams))
    k = rank
    hp = dict(lam0)
    for _ in range(k):
        for key in hp:
            hp[key] = hp[key] * args.decaying_factor
    base = manager.heuristic_exp_dir
    manager.heuristic_exp_dir = base + "_spec%d" % k
    _, acc = manager.method.train(args, manager, hp)
    manager.heuristic_exp_dir = base
    accs = gather_scalars({k: acc})
    thr = finetune_acc * args.inv_drop_margin
    ok = sorted(kk for kk, a in accs.items() if a >= thr)
    return (ok[0] if ok else None), accs
