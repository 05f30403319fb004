"""The driver's --shard mode end to end on the GPU: two ranks (RCCL when the box has two GPUs, otherwise both ranks on
cuda:0 with gloo carrying the metrics / model files) run the SI first-task dump and a 2-task EWC sequence with --test.
Both ranks must end with the same decisions, the same saved models (bitwise) and the same result tables, each in its
own results tree."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp, ndev):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank % ndev), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank % ndev)
    from clsurvey_amd.framework import driver, shard
    from clsurvey_amd.methods import method as M
    assert shard.init_from_env("nccl" if ndev >= world else "gloo") == (rank, world)
    if ndev >= 2:                 # a box with two devices MUST carry this test over RCCL (one rank per GPU), never over gloo
        assert torch.distributed.get_backend() == "nccl" and torch.cuda.current_device() == rank
    common = ["small_VGG9_cl_128_128", "--lr_grid", "1e-2,3e-3,1e-3", "--num_epochs", "6", "--batch_size", "40",
              "--saving_freq", "100", "--results_root", tmp, "--synthetic", "2,4,160,40,40,32", "--shard",
              "--device", "cuda:%d" % (rank % ndev)]
    driver.main(common + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"))
    out = driver.main(common + ["--method_name", "EWC", "--test", "--drop_margin", "0.05"], method=M.parse("EWC"))
    mgr = out["manager"]
    assert mgr.previous_task_model_path.startswith(os.path.join(tmp, "rank%d" % rank))
    assert 0 < mgr.grid_fill_factor <= 1.0
    model = torch.load(out["model_paths"][-1], weights_only=False)
    digest = [float(p.detach().double().sum().cpu()) for p in model.parameters()]
    omega = [float(model.reg_params[p]["omega"].double().sum().cpu()) for p in model.parameters() if p in model.reg_params]
    mine = dict(results=out["results"], grid=mgr.grid_trace, trace=out["frameworks"][-1].trace,
                attempts=out["frameworks"][-1].attempts, digest=digest, omega=omega)
    other = shard.broadcast_object(mine, src=0)
    assert other["grid"] == mine["grid"] and other["trace"] == mine["trace"] and other["attempts"] == mine["attempts"]
    assert other["digest"] == mine["digest"] and other["omega"] == mine["omega"]
    assert other["results"] == mine["results"]
    assert sorted(mine["results"]) == [0, 1] and len(mine["results"][0]["seq_res"][0]) == 2
    assert len(mine["grid"]) == 3 and len(mine["trace"]) >= 1
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_driver_shard_two_ranks(tmp_path):
    ndev = torch.cuda.device_count()
    port = 29500 + (os.getpid() * 7) % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path), ndev), nprocs=2, join=True)


def _worker_rccl_one_rank(rank, world, port, tmp):
    """World size 1 over the REAL backend: every collective of the sharded driver (scalar gathers on device float64, model files as
    device uint8, the ok-flag all_reduce(MIN), barriers, object broadcasts) goes through ProcessGroupNCCL = RCCL, with device tensors
    on cuda:0 — what a 1-GPU box can check of the N > 1 path's use of the library (dtypes, devices, eager communicator set-up)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", CLHIP_SHARD_FORCE_COLLECTIVES="1")
    torch.cuda.set_device(0)
    import torch.distributed as dist
    from clsurvey_amd.framework import driver, shard
    from clsurvey_amd.methods import method as M
    assert shard.init_from_env("nccl") == (0, 1)
    assert dist.get_backend() == "nccl"
    # the module's collectives one by one
    assert shard.gather_scalars({3: 0.25, 1: 0.5}) == {1: 0.5, 3: 0.25}
    assert shard.broadcast_bytes(b"model-bytes" * 1000, 0) == b"model-bytes" * 1000
    assert shard.broadcast_object({"lr": 1e-3, "ok": True}, src=0) == {"lr": 1e-3, "ok": True}
    shard.all_ok(True, "smoke")
    shard.barrier()
    t = torch.arange(8, dtype=torch.float32, device="cuda:0")
    dist.broadcast(t, src=0)                                     # bench.py's arena broadcast
    table = [torch.zeros(2, dtype=torch.float64, device="cuda:0")]
    dist.all_gather(table, torch.tensor([0.5, 0.0], dtype=torch.float64, device="cuda:0"))      # bench.py's node metrics
    assert float(table[0][0]) == 0.5
    # ... and the driver itself, sharded over that one rank
    common = ["small_VGG9_cl_128_128", "--lr_grid", "1e-2,3e-3", "--num_epochs", "4", "--batch_size", "40",
              "--saving_freq", "100", "--results_root", tmp, "--synthetic", "2,4,160,40,40,32", "--shard", "--device", "cuda:0"]
    driver.main(common + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"))
    out = driver.main(common + ["--method_name", "EWC", "--test", "--drop_margin", "0.05"], method=M.parse("EWC"))
    assert out["manager"].previous_task_model_path.startswith(os.path.join(tmp, "rank0"))
    assert sorted(out["results"]) == [0, 1] and len(out["manager"].grid_trace) == 2
    assert shard.STATS.get("broadcast_calls", 0) > 0 and shard.STATS.get("all_gather_calls", 0) > 0
    shard.barrier()
    dist.destroy_process_group()


def test_driver_shard_one_rank_over_rccl(tmp_path):
    port = 29500 + (os.getpid() * 11 + 3) % 2000
    mp.spawn(_worker_rccl_one_rank, args=(1, port, str(tmp_path)), nprocs=1, join=True)


def _worker_methods(rank, world, port, tmp, ndev):
    """`--shard --methods EWC,MAS` on 4 ranks: two blocks of two ranks, one method each."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank % ndev), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank % ndev)
    from clsurvey_amd.framework import driver, shard
    from clsurvey_amd.methods import method as M
    assert shard.init_from_env("nccl" if ndev >= world else "gloo") == (rank, world)
    common = ["small_VGG9_cl_128_128", "--lr_grid", "1e-2,3e-3,1e-3", "--num_epochs", "5", "--batch_size", "40",
              "--saving_freq", "100", "--results_root", tmp, "--synthetic", "2,4,160,40,40,32", "--shard",
              "--device", "cuda:%d" % (rank % ndev)]
    driver.main(common + ["--method_name", "SI", "--runmode", "first_task_basemodel_dump"], method=M.parse("SI"))   # the whole world
    res = driver.main(common + ["--methods", "EWC,MAS", "--test", "--drop_margin", "0.05"])
    assert res["blocks"] == {"EWC": [0, 1], "MAS": [2, 3]} and res["method"] == ("EWC" if rank < 2 else "MAS")
    assert shard.rank_world() == (rank, world)                       # back in the world
    out = res["out"]
    mgr = out["manager"]
    assert mgr.method.name == res["method"] and "/%s/" % res["method"] in mgr.parent_exp_dir
    assert mgr.previous_task_model_path.startswith(os.path.join(tmp, "rank%d" % rank))        # trees are per GLOBAL rank
    model = torch.load(out["model_paths"][-1], weights_only=False)
    digest = [float(p.detach().double().sum().cpu()) for p in model.parameters()]
    omega = [float(model.reg_params[p]["omega"].double().sum().cpu()) for p in model.parameters() if p in model.reg_params]
    mine = dict(method=res["method"], results=out["results"], grid=mgr.grid_trace, trace=out["frameworks"][-1].trace, digest=digest, omega=omega)
    table = [None] * world
    torch.distributed.all_gather_object(table, mine)
    for a, b in ((0, 1), (2, 3)):                                    # the two ranks of a block end in the same state
        for k in ("method", "results", "grid", "trace", "digest", "omega"):
            assert table[a][k] == table[b][k], (a, b, k)
    assert table[0]["omega"] != table[2]["omega"]                    # Fisher vs MAS importance: the blocks really ran different methods
    assert table[0]["grid"] == table[2]["grid"]                      # ... whose phase 1 (plain finetuning from the same model) coincides
    shard.world_barrier()
    torch.distributed.destroy_process_group()


def test_driver_methods_side_by_side(tmp_path):
    ndev = torch.cuda.device_count()
    port = 27500 + (os.getpid() * 11) % 2000
    mp.spawn(_worker_methods, args=(4, port, str(tmp_path), ndev), nprocs=4, join=True)
