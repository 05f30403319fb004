"""world_size-2 gloo test of the N>1 path: grid-node assignment, scalar all-gather, flat model
broadcast, and that both ranks take the same best-LR decision as the sequential rule."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clsurvey_amd.framework import driver, shard
    # --- collectives
    got = shard.gather_scalars({rank: 0.1 * (rank + 1), rank + 2: 0.5 + rank})
    assert got == {0: 0.1, 2: 0.5, 1: 0.2, 3: 1.5}, got
    torch.manual_seed(rank)
    m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref = [p.detach().clone() for p in m.parameters()]
    shard.broadcast_model(m, src=0)
    torch.manual_seed(0)
    m0 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    for p, q in zip(m.parameters(), m0.parameters()):
        assert torch.equal(p, q)
    if rank == 1:
        assert not all(torch.equal(a, b) for a, b in zip(ref, m0.parameters()))

    # --- sharded grid == sequential decision
    class DS:
        name = argname = test_results_dir = train_exp_results_dir = "fake"
        task_count = 1

        def get_taskname(self, i):
            return str(i)

    trained = []

    class M:
        name = eval_name = "finetuning"
        hyperparams = {}

        def grid_train(self, args, manager, lr):
            trained.append(lr)
            return None, {1e-2: 0.3, 5e-3: 0.9, 1e-3: 0.7, 5e-4: 0.2, 1e-4: 0.1}[lr]

    class Args:
        task_counter = 1
        lrs = [1e-2, 5e-3, 1e-3, 5e-4, 1e-4]
        finetune_iterations = 1

    mgr = driver.Manager(DS(), M(), "prev", os.path.join(tmp, "exp%d" % rank), None)
    args = Args()
    factory = shard.sharded_grid_factory()
    mgr.ft_parent_exp_dir = os.path.join(mgr.parent_exp_dir, "task_1", "FT_LR_GRIDSEARCH")
    best_lr, best_acc = driver.lr_grid_single_task(args, mgr, "keep_none", train_node=factory(args, mgr))
    assert (best_lr, best_acc) == (5e-3, 0.9)
    assert trained == ([1e-2, 1e-3, 1e-4] if rank == 0 else [5e-3, 5e-4]), trained
    dist.barrier()
    dist.destroy_process_group()


def test_grid_shard_world2(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
