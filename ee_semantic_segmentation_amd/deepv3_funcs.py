"""Train / evaluate orchestration of the reference (deepv3_funcs.py:19-197 ``train_deepv3``,
:200-279 ``eval_deepv3``) over the HIP path: same ``dts_info`` keys, SGD parameter groups
(base_model @ base_lr, branches @ lr, classifier @ 1.1*lr; momentum .9, wd 5e-4), poly LR
schedule, best-checkpoint dict, per-epoch tracker CSV, final test mIoU row appended to
``./mIoU_{n}_branches_results.csv``.  Fixes: B-1 (``-t`` honoured), B-10 (NameError on a
second batch size)."""
import datetime as dttm
import os
from collections import defaultdict

import numpy as np
import torch
from pandas import DataFrame
from torch import optim, utils

from . import from_deepv3_new as dv3
from .eval_mIoU import mIoU_evaluator
from .optim import SGD
from .parallel import ShardSampler, dp_info, eval_shard, init_data_parallel
from .train_funcs import train

get_metric = {"mIoU": mIoU_evaluator}          # module_variables.py:112


def _say(msg, use_file):
    if torch.distributed.is_initialized() and torch.distributed.get_rank() != 0:
        return                                   # data parallel: rank 0 keeps the log
    if use_file:
        with open(use_file, "a") as f:
            f.write(msg)
    else:
        print(msg)


def train_deepv3(net, num_epochs, kwargs):
    net_id = kwargs.get("name", kwargs.get("net_id"))
    train_set, val_loader = kwargs["train_set"], kwargs["val_loader"]
    num_epochs = kwargs["num_epochs"]
    device = kwargs.get("device", torch.device("cpu"))
    use_file, res_dir = kwargs.get("use_file"), kwargs["mod_dir"]
    batch_size = kwargs["batch_sizes"]
    lr, min_lr, base_lr = kwargs["lr"], kwargs.get("min_lr", 0), kwargs.get("base_lr")
    freeze_backbone, freeze_from = kwargs.get("freeze_backbone", False), kwargs.get("freeze_from", False)
    weighted_lr = kwargs.get("weighted_lr", False)
    patience, loss = kwargs.get("patience"), kwargs["loss"]
    metrics = [(i, get_metric[i]) for i in kwargs["metrics"]]
    train_metrics = [(i, get_metric[i]) for i in kwargs["metrics"][:2]]
    use_scheduler = kwargs.get("use_scheduler")
    start_from = kwargs.get("start_from")
    if start_from:
        start_from = os.path.join(kwargs["main_dir"], start_from)
    minimize = kwargs.get("minimize", True)
    n_branches = getattr(net, "n_branches", None)

    net.to(device)
    params = []
    if n_branches and base_lr:                                   # deepv3_funcs.py:74-99
        if freeze_backbone:
            for p in net.base_model.parameters():
                p.requires_grad = False
            if freeze_from:
                for p in net.branches[freeze_from:].parameters():
                    p.requires_grad = False
        else:
            params.append({"params": net.base_model.parameters(), "lr": base_lr})
        if weighted_lr:
            weights = np.linspace(1, 1.2, num=n_branches)
            params.extend({"params": net.branches[i].parameters(), "lr": lr * weights[i]}
                          for i in range(len(weights) - 1))
            params.append({"params": net.classifier.parameters(), "lr": lr * weights[-1]})
        elif freeze_backbone:
            br = net.branches[:freeze_from] if freeze_from else net.branches
            params.append({"params": br.parameters(), "lr": lr})
            params.append({"params": net.classifier.parameters(), "lr": lr})
        else:
            params.append({"params": net.branches.parameters(), "lr": lr})
            params.append({"params": net.classifier.parameters(), "lr": lr * 1.1})
        optimizer = SGD(params, lr=lr, momentum=.9, weight_decay=5e-4)
    else:
        optimizer = SGD(net.parameters(), lr=lr, momentum=.9, weight_decay=5e-4)
    if hasattr(net, "enable_grad_arena") and device.type == "cuda":
        # (also with a frozen backbone: autograd then never enters the frozen sections' backward, their arena slices stay zero and
        # the data-parallel reducer sends them with the buckets it launches at the end of the step - parallel.ArenaReducer.finish)
        net.enable_grad_arena()
        net.fused_outputs = True

    _say(f"--> Started training {net_id} (time: {dttm.datetime.now().strftime('%m/%d %H:%M:%S')})\n", use_file)
    saveat = os.path.join(res_dir, f"{net_id}.pth")
    save_model = kwargs.get("save_model", saveat[:-4] + "final.pth")
    net_res = None
    for phase, b_size in enumerate(batch_size if isinstance(batch_size, list) else [batch_size]):
        _say(f"<< {net_id} progress update >> B. Size: {b_size}; time: {dttm.datetime.now().strftime('%H:%M:%S')}\n",
             use_file)
        num_workers = kwargs["def_nworkers"](b_size) if "def_nworkers" in kwargs else 0
        p_factor = kwargs["def_prefetch"](b_size) if "def_prefetch" in kwargs and num_workers else None
        scheduler, ret_lr = None, False
        # deepv3_funcs.py:53-57: 's_patience', overridden by int(patience * .5) whenever early stopping has a patience
        # ('scheduler_patience' is kept as an alias of this package's round-2 key)
        sp = kwargs.get("s_patience", kwargs.get("scheduler_patience")) if use_scheduler else None
        if use_scheduler and patience:
            sp = int(patience * .5)
        if use_scheduler and sp:                                 # deepv3_funcs.py:139-146
            floors = lr * .01 if not kwargs.get("base_lr") else \
                [kwargs["base_lr"] * .01 for _ in range(len(optimizer.param_groups) - 1)] + [lr * .01]
            scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=.75, mode="min" if minimize else "max",
                                                             patience=sp, eps=1e-6, min_lr=floors)
            ret_lr = True
        elif use_scheduler:                                      # poly schedule, deepv3_funcs.py:148-153
            if min_lr:
                w = (min_lr / lr) ** (1 / .9)
                N_0 = num_epochs * w / (1 - w)
                scheduler = optim.lr_scheduler.LambdaLR(
                    optimizer, lr_lambda=lambda k: (1 - k / (num_epochs + N_0)) ** .9)
            else:
                scheduler = optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda k: (1 - k / num_epochs) ** .9)
            ret_lr = True
        # `b_size` is the GLOBAL batch (the reference's single device sees all of it, main_bradeepv3.py:119); under data
        # parallelism rank r loads the r-th b_size/world slice of every global batch (parallel.ShardSampler; the ragged
        # last batch is dropped there when world > 1, kept like the reference's drop_last=False otherwise)
        world, rank = dp_info(net)
        # the order follows torch.manual_seed like DataLoader(shuffle=True) does (every rank holds the same initial seed: the
        # entry points seed before building the network), and every batch-size phase draws its own permutations
        seed = kwargs.get("seed")
        if seed is None:
            seed = torch.initial_seed() % (2 ** 31)
            if world > 1:                                        # never seeded: the ranks' initial seeds differ - take rank 0's
                import torch.distributed as dist
                from .comm import host_group
                box = [seed]
                dist.broadcast_object_list(box, src=0, group=host_group(getattr(net.cfg, "group", None)))
                seed = box[0]
        sampler = ShardSampler(len(train_set), b_size, world, rank, seed=seed + 7919 * phase)
        train_loader = utils.data.DataLoader(train_set, batch_size=b_size // world, sampler=sampler,
                                             num_workers=num_workers, drop_last=False, prefetch_factor=p_factor,
                                             pin_memory=True)
        aux = train(net, train_loader, loss, val_iter=val_loader, num_epochs=num_epochs, updater=optimizer,
                    patience=patience, saveat=saveat, start_from=start_from or None, device=device,
                    use_file=use_file, verbose=True, metrics=train_metrics, name=net_id, scheduler=scheduler,
                    min_lr=min_lr, ret_lr=ret_lr, minimize=minimize, n_branches=n_branches,
                    nout_channels=kwargs["nout_channels"], use_graph=kwargs.get("use_graph", True),
                    save_last=kwargs.get("save_last"))
        net_res = {k: v + aux[k] for k, v in net_res.items()} if net_res else aux      # B-10 fixed
    if dp_info(net)[1] == 0:
        DataFrame.from_dict({k: v for k, v in net_res.items()}).to_csv(os.path.join(res_dir, f"{net_id}_tr.csv"),
                                                                       index=False)
    save_dict = torch.load(saveat, weights_only=True)
    net.load_state_dict(save_dict["model_state_dict"])
    kwargs["best_epoch"] = save_dict.get("epoch")      # which epoch's weights the final model holds (identical on every rank)
    kwargs["tracker"] = {k: list(v) for k, v in net_res.items()}
    if dp_info(net)[0] > 1:
        torch.distributed.barrier()               # every rank has read the best checkpoint before rank 0 replaces the file
    if dp_info(net)[1] == 0:
        torch.save(net.state_dict(), save_model)       # state_dict, not a pickled module (safe to reload)
    if dp_info(net)[0] > 1:
        torch.distributed.barrier()
    _say(f"--> Finished training {net_id} (time: {dttm.datetime.now().strftime('%m/%d %H:%M:%S')})\n", use_file)
    return save_model


def eval_deepv3(kwargs):
    res_dir, device = kwargs["res_dir"], kwargs["device"]
    use_file, name = kwargs.get("use_file"), kwargs["name"]
    saveat = os.path.join(res_dir, name)
    kwargs["mod_dir"] = saveat
    os.makedirs(saveat, exist_ok=True)
    n_branches = kwargs["n_branches"]
    btype = kwargs.get("type", "resnet101")                      # B-1: the reference ignores -t
    C = kwargs["nout_channels"]
    fine_tune = kwargs.get("fine_tune")
    net = dv3.branchyDeepv3(fine_tune or None, f"deeplabv3_{btype}", n_branches, kwargs["input_dim"],
                            count_branches=kwargs["count_branches"], skip=kwargs["skip"],
                            branch_params=kwargs.get("branch_params"), num_classes=C,
                            compute_dtype=kwargs.get("compute_dtype", torch.float32))
    net.to(device)
    # data parallel: under torch.distributed.run (main_bradeepv3 initialises the process group) every rank
    # builds the same network, takes rank 0's weights, BatchNorm becomes SyncBN and the collectives go over RCCL
    if torch.distributed.is_initialized() and device.type == "cuda":
        init_data_parallel(net, sync_bn=True, transport=kwargs.get("dp_transport"))
    world, rank = dp_info(net)
    if n_branches and n_branches != net.n_branches:
        n_branches = net.n_branches
        kwargs["loss"].update_n(n_branches)
        kwargs["n_branches"] = n_branches
        _say(f"<< {name} progress update >> Number of branches is different then antecipated: {n_branches} "
             f"branches\n", use_file)
    final_model = os.path.join(saveat, name + ".pth")
    if kwargs.get("num_epochs", 0):
        val_loader = utils.data.DataLoader(eval_shard(kwargs["val_set"], world, rank), batch_size=5, shuffle=False,
                                           num_workers=0, drop_last=False)
        kwargs |= {"val_loader": val_loader, "save_model": final_model}
        final_model = train_deepv3(net, kwargs["num_epochs"], kwargs)
        net.load_state_dict(torch.load(final_model, weights_only=True))
    else:
        if rank == 0:
            torch.save(net.state_dict(), final_model)
    net.to(device)
    net.eval()
    test_loader = utils.data.DataLoader(eval_shard(kwargs["test_set"], world, rank), batch_size=5, shuffle=False,
                                        num_workers=0, drop_last=False)
    aux_res = mIoU_evaluator(net, n_branches + 1, C, test_loader, device)      # counters summed over the ranks
    kwargs["test_result"] = dict(aux_res)
    if getattr(net.cfg, "comm", None) is not None:      # collective teardown; the training graphs died with train()
        net.cfg.comm.close()
        net.cfg.comm = None
    if rank != 0:
        return final_model
    res = defaultdict(list)
    res["net_id"].append(name)
    for key, val in aux_res.items():
        res[key].append(val)
    mIoU_res = f"./mIoU_{n_branches}_branches_results.csv"
    DataFrame.from_dict(res).set_index("net_id").to_csv(mIoU_res, mode="a", header=not os.path.exists(mIoU_res))
    return final_model
