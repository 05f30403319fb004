"""Sharding the Continual Hyperparameter Framework over the GPUs of one node.

What shards, all of it independent trainings — there is NO collective on the per-batch data path:
  * phase 1: the LR grid, `for lr in args.lrs` x finetune_iterations (framework/lr_grid_train.py:51,60): nodes start
    from the same model and are seeded by their iteration index only, so node i simply runs on rank i % world;
  * phase 2: the stability-decay attempts (framework_train.py:100-136) run speculatively, attempt `attempts + rank` on
    rank `rank`, each at the hyper-parameters the reference's decay schedule reaches after that many failures
    (one-at-a-time order for methods with several hyper-parameters, framework_train.py:168-216, replayed on a copy of
    the framework state); the smallest attempt index that meets acc >= A_ft (1 - p) is accepted, exactly the
    sequential rule;
  * evaluation: the (task, model) pairs of eval.py:146-247.
What does not: the task order, and a single training.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Every
rank executes the same host logic on ITS OWN results tree (`<results_root>/rank<r>` — no shared filesystem is assumed and
no two ranks ever write one path); what a rank needs from another travels through collectives, models and metrics only:
  * all_gather of (index, accuracy) scalars after a batch of trainings,
  * broadcast of the winner's saved model file(s) from the rank that trained it (2.4 - 36 MB for the VGG9s),
  * broadcast of a flat parameter arena (bench.py: start model / winner).
Sharded runs are reproducible for a fixed world size; against a 1-rank run they differ where the reference's own
results depend on the RNG state left by the previous training (it never re-seeds between decay attempts): every
speculative attempt is seeded by (task, attempt index) instead.
"""
import copy
import fnmatch
import os
import shutil

import torch
import torch.distributed as dist


# traffic of this process's collectives since import (bench.py --gpus N reports it: models and metrics only)
STATS = {"broadcast_calls": 0, "broadcast_bytes": 0, "broadcast_s": 0.0, "all_gather_calls": 0, "all_gather_bytes": 0,
         "all_gather_s": 0.0, "all_reduce_calls": 0, "all_reduce_s": 0.0,
         # seconds THIS rank spent training / evaluating inside each sharded stage (what filled its GPU), and the units it ran
         "grid_busy_s": 0.0, "grid_nodes": 0, "decay_busy_s": 0.0, "decay_attempts": 0, "eval_busy_s": 0.0, "eval_pairs": 0,
         # phase-2 trainings of ALL the ranks this process shards with (the same numbers on each of them): what ran, and what
         # the sequential rule would have run (attempts up to and including the accepted one) — the rest was speculation
         "decay_trainings_group": 0, "decay_trainings_useful": 0}


def _is_cuda(device):
    return device is not None and str(device).startswith("cuda") and torch.cuda.is_available()


class busy:
    """with shard.busy("grid", device=args.device): ...   adds the block's wall-clock to STATS["grid_busy_s"] and `units` to
    its unit counter.  The device queue is drained before the clock is read only when the RUN's device is a GPU: a CPU run
    (the oracle legs of bench.py's pair, --device cpu) must not create or synchronise a device context beside a GPU run that
    is being timed."""
    _unit = {"grid": "grid_nodes", "decay": "decay_attempts", "eval": "eval_pairs"}

    def __init__(self, stage, units=1, device=None):
        self.stage, self.units, self.sync = stage, units, _is_cuda(device)

    def __enter__(self):
        import time
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import time
        if self.sync:
            torch.cuda.synchronize()
        STATS[self.stage + "_busy_s"] += time.perf_counter() - self.t0
        STATS[self._unit[self.stage]] += self.units


class _timed:
    """Wall-clock of a collective as this rank sees it (waiting for the slowest rank included), into STATS[key]."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        import time
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import time
        if dist.is_initialized() and dist.get_backend() == "nccl":
            torch.cuda.synchronize()
        STATS[self.key] += time.perf_counter() - self.t0


# The ranks this process shards WITH.  None: the whole world.  `--methods A,B` (driver.main) splits the world into one group of
# disjoint ranks per method (BASELINE configs 3 and 5 are two methods on one 8-GPU node); inside a group every
# function of this module behaves as if the group were the world: ranks are group-relative, collectives run on the group's
# communicator (dist.new_group), `src` arguments are group ranks.
_G = {"group": None, "ranks": None}


def set_group(group, ranks):
    _G["group"], _G["ranks"] = group, (list(ranks) if ranks is not None else None)


def global_rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def rank_world():
    """(rank, world) of the ranks this process shards with: the method's group under --methods, else the world."""
    if dist.is_available() and dist.is_initialized():
        if _G["ranks"] is not None:
            return _G["ranks"].index(dist.get_rank()), len(_G["ranks"])
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _src(r):
    """group rank -> the global rank torch.distributed's `src` wants."""
    return _G["ranks"][r] if _G["ranks"] is not None else r


def split_ranks(n_groups, world):
    """Contiguous, near-equal blocks of the world's ranks, one per group (the first world % n_groups groups get one more):
    8 ranks, 2 methods -> [0..3], [4..7] (neighbouring GPUs share xGMI links either way: every pair is one hop)."""
    if n_groups > world:
        raise ValueError("%d methods need at least %d ranks, the world has %d" % (n_groups, n_groups, world))
    out, lo = [], 0
    for g in range(n_groups):
        n = world // n_groups + (1 if g < world % n_groups else 0)
        out.append(list(range(lo, lo + n)))
        lo += n
    return out


def enter_method_groups(n_groups):
    """Every rank of the world calls this with the same n_groups: creates ALL the groups' communicators (dist.new_group is
    collective over the world, same order everywhere) and enters this rank's.  Returns (group index, ranks of every group)."""
    rank, world = global_rank_world()
    blocks = split_ranks(n_groups, world)
    mine = None
    for g, ranks in enumerate(blocks):
        pg = dist.new_group(ranks=ranks)
        if rank in ranks:
            mine = (g, pg, ranks)
    set_group(mine[1], mine[2])
    return mine[0], blocks


def leave_method_groups():
    set_group(None, None)


def init_from_env(backend=None):
    """torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment.  Returns
    (rank, world); a single process initialises nothing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and _solo(1):
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=int(os.environ["RANK"]), world_size=world)
    return rank_world()


def _solo(world):
    """A one-rank world needs no collective — unless CLHIP_SHARD_FORCE_COLLECTIVES=1 asks for them anyway (tests/test_shard_gpu.py:
    every collective of this module issued over RCCL on a 1-GPU box, where a second rank cannot exist)."""
    return world == 1 and os.environ.get("CLHIP_SHARD_FORCE_COLLECTIVES") != "1"


def _dev():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def assign(n_nodes, world):
    """node i -> rank i % world (round-robin keeps the expensive low-LR / high-LR nodes apart)."""
    return [i % world for i in range(n_nodes)]


def fill_factor(n_nodes, world):
    """Busy fraction of the ranks while a batch of n_nodes equal trainings runs: n / (world * ceil(n / world))."""
    rounds = -(-n_nodes // world)
    return n_nodes / float(world * rounds) if n_nodes else 0.0


def gather_scalars(values):
    """values: {index: float} computed on this rank -> merged dict on every rank."""
    rank, world = rank_world()
    if _solo(world):
        return dict(values)
    n = torch.tensor([len(values)], dtype=torch.int64, device=_dev())
    counts = [torch.zeros_like(n) for _ in range(world)]
    with _timed("all_gather_s"):
        dist.all_gather(counts, n, group=_G["group"])
    STATS["all_gather_calls"] += 2
    m = max(int(c.item()) for c in counts)
    buf = torch.full((max(m, 1), 2), -1.0, dtype=torch.float64, device=_dev())
    for j, (k, v) in enumerate(sorted(values.items())):
        buf[j, 0], buf[j, 1] = float(k), float(v)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    with _timed("all_gather_s"):
        dist.all_gather(bufs, buf, group=_G["group"])
    STATS["all_gather_bytes"] += world * (8 + buf.numel() * 8)
    out = {}
    for b in bufs:
        for k, v in b.cpu().tolist():
            if k >= 0:
                out[int(k)] = v
    return out


def broadcast_model(model, src=0):
    """Broadcast all parameters of `model` from rank src as ONE flat tensor (one RCCL broadcast)."""
    rank, world = rank_world()
    if _solo(world):
        return model
    params = [p.data for p in model.parameters()]
    flat = torch.cat([p.reshape(-1).to(_dev(), torch.float32) for p in params])
    with _timed("broadcast_s"):
        dist.broadcast(flat, src=_src(src), group=_G["group"])
    STATS["broadcast_calls"] += 1
    STATS["broadcast_bytes"] += flat.numel() * 4
    off = 0
    for p in params:
        n = p.numel()
        p.copy_(flat[off:off + n].view(p.shape).to(p.device))
        off += n
    return model


def broadcast_bytes(payload, src):
    """bytes on rank src (anything on the others) -> the same bytes everywhere; length first, then one uint8 tensor."""
    rank, world = rank_world()
    if _solo(world):
        return payload
    n = torch.tensor([len(payload) if rank == src else 0], dtype=torch.int64, device=_dev())
    with _timed("broadcast_s"):
        dist.broadcast(n, src=_src(src), group=_G["group"])
    size = int(n.item())
    if rank == src:
        buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(_dev()) if size else torch.zeros(0, dtype=torch.uint8, device=_dev())
    else:
        buf = torch.zeros(size, dtype=torch.uint8, device=_dev())
    if size:
        with _timed("broadcast_s"):
            dist.broadcast(buf, src=_src(src), group=_G["group"])
    STATS["broadcast_calls"] += 2 if size else 1
    STATS["broadcast_bytes"] += 8 + size
    return bytes(buf.cpu().numpy().tobytes())


def broadcast_files(directory, src, patterns=("*.pth.tar", "*.pth", "*.FLAG")):
    """The model / checkpoint files directly under `directory` on rank src appear under the same relative name in
    every rank's `directory` (which is a different absolute tree per rank).  Returns the file names."""
    rank, world = rank_world()
    if _solo(world):
        return []
    names = []
    if rank == src and os.path.isdir(directory):
        # (no glob on the path: experiment directories carry the LR grid in brackets, a glob character class)
        names = sorted(n for n in os.listdir(directory) if os.path.isfile(os.path.join(directory, n))
                       and any(fnmatch.fnmatch(n, pat) for pat in patterns))
    names = broadcast_bytes("\n".join(names).encode(), src).decode().split("\n")
    # tokens ("*.FLAG" marks a directory as finished) travel LAST: a copy cut short by a kill is never marked done
    names = sorted((n for n in names if n), key=lambda n: (n.endswith(".FLAG"), n))
    if rank != src:
        os.makedirs(directory, exist_ok=True)
    for name in names:
        path = os.path.join(directory, name)
        data = b""
        if rank == src:
            with open(path, "rb") as f:
                data = f.read()
        data = broadcast_bytes(data, src)
        if rank != src:
            with open(path, "wb") as f:
                f.write(data)
    return names


def broadcast_object(obj, src=0):
    import pickle
    return pickle.loads(broadcast_bytes(pickle.dumps(obj), src))


def barrier():
    if not _solo(rank_world()[1]):
        dist.barrier(group=_G["group"])


def world_barrier():
    if global_rank_world()[1] > 1:
        dist.barrier()


def all_ok(ok, what=""):
    """Collective error check behind a sharded stage: every rank reports whether ITS part went through; if any did not,
    EVERY rank raises (a rank that failed alone would leave the others waiting in the next collective until the
    communicator times out, with no diagnostic)."""
    rank, world = rank_world()
    if not _solo(world):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_dev())
        with _timed("all_reduce_s"):
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_G["group"])
        STATS["all_reduce_calls"] += 1
        everyone = bool(int(flag.item()))
    else:
        everyone = bool(ok)
    if not everyone:
        raise RuntimeError("sharded stage failed on %s%s" % ("this rank" if not ok else "another rank", (": " + what) if what else ""))


# ------------------------------------------------------------------------------------------------ phase 1
def sharded_grid_factory():
    """train_node factory for driver.main(): every rank trains the grid nodes assigned to it, the accuracies are
    all-gathered, and each rank then replays the reference's sequential selection rule over the complete table (same
    decision everywhere).  After the selection the winner's node directory is copied from its owner to every rank
    (manager.after_grid), because the methods without a phase 2 adopt that model as the task's model."""
    from . import driver

    def factory(args, manager):
        rank, world = rank_world()
        nodes = [(lr, it) for lr in args.lrs for it in range(args.finetune_iterations)]
        owner = assign(len(nodes), world)
        table = {}

        def node_dir(lr, it):
            d = "lr=" + driver.float_to_scientific_str(lr) + ("_it%d" % it if args.finetune_iterations > 1 else "")
            return os.path.join(manager.ft_parent_exp_dir, d)

        def run_all():
            mine, err = {}, None
            try:
                for i, (lr, it) in enumerate(nodes):
                    if owner[i] != rank:
                        continue
                    driver.set_random(it)
                    manager.gridsearch_exp_dir = node_dir(lr, it)
                    os.makedirs(manager.gridsearch_exp_dir, exist_ok=True)
                    with busy("grid", device=getattr(args, "device", None)):
                        _, acc = manager.method.grid_train(args, manager, lr)
                    mine[i] = acc
            except Exception as e:                        # reported collectively below
                import traceback
                traceback.print_exc()
                err = e
            all_ok(err is None, "phase-1 grid node")
            table.update(gather_scalars(mine))

        def train_node(lr, it):
            if not table:
                run_all()
            return table[nodes.index((lr, it))]

        def after_grid(args_, manager_, best_lr):
            best_dir = manager_.best_exp_grid_node_dirname
            if best_dir is None:
                return
            src = next(owner[i] for i, (lr, it) in enumerate(nodes) if node_dir(lr, it) == best_dir)
            broadcast_files(best_dir, src)
        manager.after_grid = after_grid
        # every rank replays the grid over ONE table of finished nodes — rank 0's checkpoint — so that the decision to
        # train (a collective) is taken by all ranks or by none (a resumed run may hold different checkpoints per rank)
        manager.sync_processed = lambda processed: broadcast_object(processed, 0)
        manager.grid_fill_factor = fill_factor(len(nodes), world)
        return train_node
    return factory


# ------------------------------------------------------------------------------------------------ phase 2
def decayed_copy(hf, args, manager, k):
    """Framework state after k more failed attempts: a deep copy of `hf` advanced by the reference's own decay rule."""
    twin = copy.copy(hf)
    twin.hyperparams = copy.deepcopy(hf.hyperparams)
    twin.hyperparams_backup = copy.deepcopy(hf.hyperparams_backup)
    for _ in range(k):
        twin.hyperparamDecay(args, manager)
        twin.attempts += 1
    return twin


def speculative_round(hf, args, manager, finetune_acc):
    """One round of speculative stability decay: rank r trains attempt `hf.attempts + r` (skipped past
    max_attempts_per_task).  Returns (accs {attempt: acc}, accepted attempt or None); on acceptance every rank's
    heuristic_exp_dir holds the accepted attempt's files."""
    from . import driver
    rank, world = rank_world()
    k = hf.attempts + rank
    mine, err = {}, None
    if k < args.max_attempts_per_task:
        twin = decayed_copy(hf, args, manager, rank)
        driver.set_random(1000 * int(args.task_counter) + k)
        print(" => ATTEMPT {}/{} (speculative, rank {}): Hyperparams {}".format(k, args.max_attempts_per_task - 1, rank,
                                                                                 twin.hyperparams))
        shutil.rmtree(manager.heuristic_exp_dir, ignore_errors=True)
        os.makedirs(manager.heuristic_exp_dir, exist_ok=True)
        manager.method.hyperparams = twin.hyperparams
        try:
            with busy("decay", device=getattr(args, "device", None)):
                _, acc = manager.method.train(args, manager, twin.hyperparams)
            mine[k] = acc
        except Exception as e:
            import traceback
            traceback.print_exc()
            err = e
    all_ok(err is None, "phase-2 attempt %d" % k)
    accs = gather_scalars(mine)
    threshold = finetune_acc * args.inv_drop_margin
    ok = sorted(kk for kk, a in accs.items() if a >= threshold)
    last = args.max_attempts_per_task - 1
    accepted = ok[0] if ok else (last if last in accs else None)      # the final attempt is kept whatever it scored
    STATS["decay_trainings_group"] += len(accs)
    STATS["decay_trainings_useful"] += sum(1 for kk in accs if accepted is None or kk <= accepted)
    if accepted is not None:
        src = accepted - hf.attempts
        if rank != src:
            shutil.rmtree(manager.heuristic_exp_dir, ignore_errors=True)
        broadcast_files(manager.heuristic_exp_dir, src)
    return accs, accepted
