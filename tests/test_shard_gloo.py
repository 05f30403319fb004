"""world_size-2 gloo tests of the N>1 path (clsurvey_amd.framework.shard + the driver's --shard mode): grid-node
assignment, scalar all-gather, flat model / file broadcast, the same best-LR decision as the sequential rule with the
winner's model arriving on every rank, speculative stability decay that ends in the sequential loop's state for a method
with TWO hyper-parameters (one-at-a-time decay order, framework_train.py:168-216), and sharded evaluation."""
import copy
import os
import sys
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _DS:
    name = argname = test_results_dir = train_exp_results_dir = "fake"
    task_count = 1

    def get_taskname(self, i):
        return str(i)


_GRID_ACC = {1e-2: 0.3, 5e-3: 0.9, 1e-3: 0.7, 5e-4: 0.2, 1e-4: 0.1}


class _GridMethod:
    """Phase-1 stand-in: accuracy is a table of the LR; the 'model' it saves names the LR and the rank that trained it."""
    name = eval_name = "finetuning"
    hyperparams = {}

    def __init__(self, rank):
        self.rank, self.trained = rank, []

    def grid_train(self, args, manager, lr):
        self.trained.append(lr)
        torch.save({"lr": lr, "rank": self.rank}, os.path.join(manager.gridsearch_exp_dir, "best_model.pth.tar"))
        return None, _GRID_ACC[lr]


class _DecayMethod:
    """Phase-2 stand-in with two hyper-parameters: accuracy rises as they shrink; the saved 'model' is the pair used."""
    name = eval_name = "twoparam"

    def __init__(self):
        self.hyperparams = OrderedDict([("a", 8.0), ("b", 4.0)])
        self.calls = []

    def train(self, args, manager, hp):
        self.calls.append(dict(hp))
        os.makedirs(manager.heuristic_exp_dir, exist_ok=True)
        torch.save(dict(hp), os.path.join(manager.heuristic_exp_dir, "best_model.pth.tar"))
        return None, 1.0 / (1.0 + hp["a"] * hp["b"])


def _args(**kw):
    class A:
        task_counter = 1
        lrs = [1e-2, 5e-3, 1e-3, 5e-4, 1e-4]
        finetune_iterations = 1
        decaying_factor = 0.5
        max_attempts_per_task = 10
        inv_drop_margin = 0.8
    a = A()
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from clsurvey_amd.framework import driver, shard
    assert shard.init_from_env("gloo") == (rank, world)
    # --- collectives
    got = shard.gather_scalars({rank: 0.1 * (rank + 1), rank + 2: 0.5 + rank})
    assert got == {0: 0.1, 2: 0.5, 1: 0.2, 3: 1.5}, got
    torch.manual_seed(rank)
    m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    shard.broadcast_model(m, src=0)
    torch.manual_seed(0)
    m0 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    assert all(torch.equal(p, q) for p, q in zip(m.parameters(), m0.parameters()))
    assert shard.broadcast_object({"from": rank}, src=1) == {"from": 1}
    assert shard.broadcast_bytes(b"x" * (3 + rank), src=1) == b"xxxx"
    assert shard.fill_factor(7, 8) == 7 / 8 and shard.fill_factor(7, 2) == 7 / 8 and shard.fill_factor(5, 1) == 1.0

    # --- sharded grid == sequential decision; the winner's file reaches the rank that did not train it
    meth = _GridMethod(rank)
    mgr = driver.Manager(_DS(), meth, "prev", os.path.join(tmp, "rank%d" % rank, "exp_lr=[0.01, 0.005]"), None)   # brackets as in real runs
    args = _args()
    best_lr, best_acc = driver.lr_grid_single_task(args, mgr, "all", train_node=shard.sharded_grid_factory()(args, mgr))
    assert (best_lr, best_acc) == (5e-3, 0.9)
    assert meth.trained == ([1e-2, 1e-3, 1e-4] if rank == 0 else [5e-3, 5e-4]), meth.trained
    assert mgr.grid_fill_factor == 5 / 6
    won = torch.load(os.path.join(mgr.best_exp_grid_node_dirname, "best_model.pth.tar"), weights_only=False)
    assert won == {"lr": 5e-3, "rank": 1}, won              # trained on rank 1, present in BOTH trees
    assert mgr.best_exp_grid_node_dirname.startswith(os.path.join(tmp, "rank%d" % rank))

    # --- speculative phase 2 with two hyper-parameters == the sequential loop's outcome
    def phase2(speculative, sub):
        meth2 = _DecayMethod()
        mg = driver.Manager(_DS(), meth2, "prev", os.path.join(tmp, "rank%d" % rank, sub), None)
        mg.speculative = speculative
        hf = driver.HyperparameterFramework(meth2)
        a2 = _args()
        hf.stabilityDecay(a2, mg, 1e-3, finetune_acc=0.25)          # threshold 0.2 <=> a * b <= 4
        model = torch.load(os.path.join(mg.heuristic_exp_dir, "best_model.pth.tar"), weights_only=False)
        saved = torch.load(os.path.join(mg.heuristic_exp_dir, "hyperparams.pth.tar"), weights_only=False)
        assert os.path.exists(mg.get_success_token_path(mg.heuristic_exp_dir))
        return hf, meth2, model, saved

    seq_hf, seq_m, seq_model, seq_saved = phase2(False, "seq")
    spec_hf, spec_m, spec_model, spec_saved = phase2(True, "spec")
    # sequential reference: (8,4) (4,4) (8,2) (4,2) (2,4)->no: one-at-a-time = a, b, then both
    assert [c for c in seq_m.calls][:3] == [{"a": 8.0, "b": 4.0}, {"a": 4.0, "b": 4.0}, {"a": 8.0, "b": 2.0}]
    assert dict(spec_hf.hyperparams) == dict(seq_hf.hyperparams) and spec_hf.attempts == seq_hf.attempts
    assert spec_hf.hyperparam_idx == seq_hf.hyperparam_idx
    assert dict(spec_hf.hyperparams_backup) == dict(seq_hf.hyperparams_backup)
    assert spec_hf.trace == seq_hf.trace
    assert spec_model == seq_model == dict(seq_hf.hyperparams)      # the accepted attempt's model, on every rank
    assert spec_saved["val_acc"] == seq_saved["val_acc"] and spec_saved["state"]["attempts"] == seq_saved["state"]["attempts"]
    # each rank ran only its own attempts: attempt r, r + 2, ...
    assert len(spec_m.calls) <= (len(seq_m.calls) + 1) // 2 + 1
    assert spec_m.calls[0] == ({"a": 8.0, "b": 4.0} if rank == 0 else {"a": 4.0, "b": 4.0})

    # --- never accepted: the last attempt is kept, state as after max_attempts failures
    def phase2_exhausted(speculative, sub):
        meth3 = _DecayMethod()
        mg = driver.Manager(_DS(), meth3, "prev", os.path.join(tmp, "rank%d" % rank, sub), None)
        mg.speculative = speculative
        hf = driver.HyperparameterFramework(meth3)
        hf.stabilityDecay(_args(max_attempts_per_task=3), mg, 1e-3, finetune_acc=10.0)
        return hf, torch.load(os.path.join(mg.heuristic_exp_dir, "best_model.pth.tar"), weights_only=False)
    s_hf, s_model = phase2_exhausted(False, "seq_x")
    p_hf, p_model = phase2_exhausted(True, "spec_x")
    assert p_hf.attempts == s_hf.attempts == 3 and dict(p_hf.hyperparams) == dict(s_hf.hyperparams)
    assert p_hf.trace == s_hf.trace and p_model == s_model

    # --- --no_speculation: phase 2 runs on rank 0 ONLY, every rank ends with rank 0's model bytes and framework state
    meth4 = _DecayMethod()
    mg4 = driver.Manager(_DS(), meth4, "prev", os.path.join(tmp, "rank%d" % rank, "seq0"), None)
    mg4.speculative, mg4.sequential_on_rank0 = False, True
    hf4 = driver.HyperparameterFramework(meth4)
    hf4.stabilityDecay(_args(), mg4, 1e-3, finetune_acc=0.25)
    assert (len(meth4.calls) > 0) == (rank == 0), (rank, meth4.calls)
    with open(os.path.join(mg4.heuristic_exp_dir, "best_model.pth.tar"), "rb") as f:
        mine = f.read()
    assert shard.broadcast_bytes(mine, 0) == mine                   # identical best_model bytes on all ranks
    assert dict(hf4.hyperparams) == dict(seq_hf.hyperparams) and hf4.attempts == seq_hf.attempts and hf4.trace == seq_hf.trace
    assert os.path.exists(mg4.get_success_token_path(mg4.heuristic_exp_dir)) and mg4.best_model_path.startswith(os.path.join(tmp, "rank%d" % rank))

    # --- resumed phase 2 whose trees DISAGREE (run killed between rank 0's success token and the broadcast of its directory):
    #     rank 0 decides for everyone, its files arrive again, no rank trains and no rank is left in a collective alone
    for mode in ("sequential_on_rank0", "speculative"):
        meth5 = _DecayMethod()
        mg5 = driver.Manager(_DS(), meth5, "prev", os.path.join(tmp, "rank%d" % rank, "resume_" + mode), None)
        mg5.speculative, mg5.sequential_on_rank0 = mode == "speculative", mode == "sequential_on_rank0"
        hd = os.path.join(mg5.parent_exp_dir, "task_1", "TASK_TRAINING")
        if rank == 0:           # rank 0 finished the task before the kill ...
            os.makedirs(hd)
            torch.save({"a": 1.0, "b": 2.0}, os.path.join(hd, "best_model.pth.tar"))
            done = driver.HyperparameterFramework(_DecayMethod())
            done.attempts, done.hyperparams["a"], done.hyperparams["b"] = 3, 1.0, 2.0
            mg5.save_hyperparams(hd, {"acc_threshold": 0.2, "val_acc": 0.33, "state": done._get_state()})
            mg5.create_success_token(hd)
        hf5 = driver.HyperparameterFramework(meth5)          # ... rank 1's tree is empty
        hf5.stabilityDecay(_args(), mg5, 1e-3, finetune_acc=0.25)
        assert meth5.calls == [], (mode, rank, meth5.calls)
        assert hf5.attempts == 3 and dict(hf5.hyperparams) == {"a": 1.0, "b": 2.0}, (mode, rank, hf5.attempts, dict(hf5.hyperparams))
        assert torch.load(os.path.join(hd, "best_model.pth.tar"), weights_only=False) == {"a": 1.0, "b": 2.0}
        assert os.path.exists(mg5.get_success_token_path(hd)) and mg5.best_model_path == os.path.join(hd, "best_model.pth.tar")
    # the other way round: only rank 1 holds a (stale) token — rank 0 says "not done", both train, the stale token is gone first
    meth6 = _DecayMethod()
    mg6 = driver.Manager(_DS(), meth6, "prev", os.path.join(tmp, "rank%d" % rank, "resume_stale"), None)
    mg6.speculative = True
    hd6 = os.path.join(mg6.parent_exp_dir, "task_1", "TASK_TRAINING")
    if rank == 1:
        os.makedirs(hd6)
        torch.save({"stale": True}, os.path.join(hd6, "best_model.pth.tar"))
        # ... and a stale checkpoint of a run killed half-way (other hyper-parameters, 5 attempts in): must not be resumed from
        stale = driver.HyperparameterFramework(_DecayMethod())
        stale.attempts, stale.hyperparams["a"], stale.hyperparams["b"] = 5, 0.125, 0.25
        mg6.save_hyperparams(hd6, {"acc_threshold": 0.2, "val_acc": 0.1, "state": stale._get_state()})
        mg6.create_success_token(hd6)
    hf6 = driver.HyperparameterFramework(meth6)
    hf6.stabilityDecay(_args(), mg6, 1e-3, finetune_acc=0.25)
    assert len(meth6.calls) > 0 and hf6.trace == seq_hf.trace
    assert torch.load(os.path.join(hd6, "best_model.pth.tar"), weights_only=False) == dict(seq_hf.hyperparams)

    # --- a failure on ONE rank ends the stage on EVERY rank (no rank is left waiting in the next collective)
    class _Failing(_GridMethod):
        def grid_train(self, args, manager, lr):
            if self.rank == 1:
                raise ValueError("node failed on rank 1")
            return super().grid_train(args, manager, lr)
    mf = _Failing(rank)
    mgf = driver.Manager(_DS(), mf, "prev", os.path.join(tmp, "rank%d" % rank, "fail"), None)
    af = _args()
    try:
        driver.lr_grid_single_task(af, mgf, "all", train_node=shard.sharded_grid_factory()(af, mgf))
        raise AssertionError("the failing grid went through on rank %d" % rank)
    except RuntimeError as e:
        assert "sharded stage failed" in str(e)

    # --- resumed run with different checkpoints per rank: rank 0's table decides for everyone (no one-sided collective)
    mr = _GridMethod(rank)
    mgr2 = driver.Manager(_DS(), mr, "prev", os.path.join(tmp, "rank%d" % rank, "resume"), None)
    ar = _args()
    ft = os.path.join(mgr2.parent_exp_dir, "task_1", "FT_LR_GRIDSEARCH")
    os.makedirs(ft)
    if rank == 1:       # rank 1 believes the whole grid is done; rank 0 has nothing
        torch.save({"processed_lrs": {lr: {"acc": [a]} for lr, a in _GRID_ACC.items()}}, os.path.join(ft, "grid_checkpoint.pth"))
    best_lr, best_acc = driver.lr_grid_single_task(ar, mgr2, "all", train_node=shard.sharded_grid_factory()(ar, mgr2))
    assert (best_lr, best_acc) == (5e-3, 0.9) and mr.trained == ([1e-2, 1e-3, 1e-4] if rank == 0 else [5e-3, 5e-4])

    # --- sharded evaluation: pairs split over the ranks, same table everywhere
    class EvalMethod:
        name = eval_name = "evalfake"

        def __init__(self):
            self.seen = []

        def inference_eval(self, a, manager):
            self.seen.append((a.eval_dset_idx, a.trained_model_idx))
            return 100.0 - 10 * a.eval_dset_idx - a.trained_model_idx

    em = EvalMethod()
    mg = driver.Manager(_DS(), em, "prev", os.path.join(tmp, "rank%d" % rank, "ev"), None)
    mg.speculative = True
    ea = _args(test_starting_task_count=1, test_max_task_count=3, out_path=os.path.join(tmp, "rank%d" % rank, "ev_out"))
    res = driver.eval_all_models_all_tasks(ea, mg, ["d0", "d1", "d2"], ["m0", "m1", "m2"])
    assert res[0]["seq_res"][0] == [100.0, 99.0, 98.0] and res[0]["seq_forgetting"][0] == [1.0, 2.0]
    assert res[1]["seq_res"][1] == [89.0, 88.0] and res[2]["seq_res"][2] == [78.0]
    assert len(em.seen) == 3 and set(em.seen) == {p for n, p in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]) if n % 2 == rank}
    dist.barrier()
    dist.destroy_process_group()


def test_grid_shard_world2(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


# ------------------------------------------------------------------------------------------------ method groups (--methods)
class _GridMethodB(_GridMethod):
    """A second method with another accuracy table: its block must reach ITS OWN decision."""
    name = eval_name = "other"
    TABLE = {1e-2: 0.2, 5e-3: 0.3, 1e-3: 0.4, 5e-4: 0.95, 1e-4: 0.6}

    def grid_train(self, args, manager, lr):
        self.trained.append(lr)
        torch.save({"lr": lr, "rank": self.rank, "method": "other"}, os.path.join(manager.gridsearch_exp_dir, "best_model.pth.tar"))
        return None, self.TABLE[lr]


def _worker_groups(rank, world, port, tmp):
    """World 4 split into two method blocks of two ranks (two methods side by side on one node)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from clsurvey_amd.framework import driver, shard
    assert shard.init_from_env("gloo") == (rank, world)
    assert shard.split_ranks(2, 8) == [[0, 1, 2, 3], [4, 5, 6, 7]] and shard.split_ranks(3, 8) == [[0, 1, 2], [3, 4, 5], [6, 7]]
    gi, blocks = shard.enter_method_groups(2)
    assert blocks == [[0, 1], [2, 3]] and gi == rank // 2
    brank = rank % 2
    assert shard.rank_world() == (brank, 2) and shard.global_rank_world() == (rank, 4)
    # collectives stay inside the block; `src` is a block rank
    got = shard.gather_scalars({brank: 10.0 * gi + brank})
    assert got == {0: 10.0 * gi, 1: 10.0 * gi + 1}, got
    assert shard.broadcast_object({"from": rank}, src=1) == {"from": blocks[gi][1]}
    shard.all_ok(True)
    # each block's sharded grid takes ITS method's decision, with the block's two ranks splitting the nodes
    meth = (_GridMethod if gi == 0 else _GridMethodB)(rank)
    mgr = driver.Manager(_DS(), meth, "prev", os.path.join(tmp, "rank%d" % rank, "grid"), None)
    args = _args()
    best_lr, best_acc = driver.lr_grid_single_task(args, mgr, "all", train_node=shard.sharded_grid_factory()(args, mgr))
    # (the reference's rule, lr_grid_train.py: LRs in grid order, a later LR replaces the best only when strictly better)
    assert (best_lr, best_acc) == ((5e-3, 0.9) if gi == 0 else (5e-4, 0.95))
    assert meth.trained == ([1e-2, 1e-3, 1e-4] if brank == 0 else [5e-3, 5e-4]), meth.trained
    won = torch.load(os.path.join(mgr.best_exp_grid_node_dirname, "best_model.pth.tar"), weights_only=False)
    assert won["rank"] == blocks[gi][1] and won["lr"] == best_lr, won          # trained on the block's rank 1, present on both

    # speculative phase 2 per block (different start values per block) == that block's sequential loop
    def phase2(speculative, sub):
        m2 = _DecayMethod()
        if gi == 1:
            m2.hyperparams = OrderedDict([("a", 16.0), ("b", 2.0)])
        mg = driver.Manager(_DS(), m2, "prev", os.path.join(tmp, "rank%d" % rank, sub), None)
        mg.speculative = speculative
        hf = driver.HyperparameterFramework(m2)
        hf.stabilityDecay(_args(), mg, 1e-3, finetune_acc=0.25)
        return hf, m2, torch.load(os.path.join(mg.heuristic_exp_dir, "best_model.pth.tar"), weights_only=False)

    before = dict(shard.STATS)
    seq_hf, seq_m, seq_model = phase2(False, "seq")
    spec_hf, spec_m, spec_model = phase2(True, "spec")
    assert spec_hf.trace == seq_hf.trace and dict(spec_hf.hyperparams) == dict(seq_hf.hyperparams) and spec_model == seq_model
    assert spec_hf.attempts == seq_hf.attempts
    ran = shard.STATS["decay_trainings_group"] - before["decay_trainings_group"]
    useful = shard.STATS["decay_trainings_useful"] - before["decay_trainings_useful"]
    assert useful == len(seq_hf.trace) and useful <= ran <= useful + 1, (ran, useful)     # two in flight: at most one speculative extra
    shard.barrier()
    shard.leave_method_groups()
    assert shard.rank_world() == (rank, world)
    shard.world_barrier()
    dist.destroy_process_group()


def test_method_groups_world4(tmp_path):
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_groups, args=(4, port, str(tmp_path)), nprocs=4, join=True)


def _worker_forced_world1(rank, world, port, tmp):
    """CLHIP_SHARD_FORCE_COLLECTIVES=1: a one-rank world issues every collective instead of short-cutting it (the hook behind
    tests/test_shard_gpu.py::test_driver_shard_one_rank_over_rccl, here over gloo) — and the sharded grid still decides as the
    sequential rule does."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", CLHIP_SHARD_FORCE_COLLECTIVES="1")
    from clsurvey_amd.framework import driver, shard
    assert not shard._solo(1)
    assert shard.init_from_env("gloo") == (0, 1) and dist.is_initialized()
    before = dict(shard.STATS)
    assert shard.gather_scalars({3: 0.25, 1: 0.5}) == {1: 0.5, 3: 0.25}
    assert shard.broadcast_bytes(b"abc" * 100, 0) == b"abc" * 100
    assert shard.broadcast_object({"k": [1, 2]}, src=0) == {"k": [1, 2]}
    shard.all_ok(True, "forced")
    shard.barrier()
    assert shard.STATS["all_gather_calls"] > before["all_gather_calls"] and shard.STATS["broadcast_calls"] > before["broadcast_calls"]
    assert shard.STATS["all_reduce_calls"] > before["all_reduce_calls"]
    meth = _GridMethod(0)
    mgr = driver.Manager(_DS(), meth, "prev", os.path.join(tmp, "rank0", "exp"), None)
    args = _args()
    best_lr, best_acc = driver.lr_grid_single_task(args, mgr, "all", train_node=shard.sharded_grid_factory()(args, mgr))
    assert (best_lr, best_acc) == (5e-3, 0.9) and meth.trained == [1e-2, 5e-3, 1e-3, 5e-4, 1e-4]
    dist.destroy_process_group()
    del os.environ["CLHIP_SHARD_FORCE_COLLECTIVES"]
    assert shard._solo(1)


def test_forced_collectives_world1(tmp_path):
    port = 29500 + (os.getpid() * 13 + 5) % 2000
    mp.spawn(_worker_forced_world1, args=(1, port, str(tmp_path)), nprocs=1, join=True)
