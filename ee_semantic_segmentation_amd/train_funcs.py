"""Training loop of the reference (train_funcs.py:12-33 ``train_epoch``, :60-269 ``train``)
over the HIP path.  Same signature / tracker / checkpoint dict; differences:
B-2 fixed (runs ``num_epochs`` epochs), validation runs in eval() mode (B-3), the
per-step work is a HIP-graph replay when the optimizer is this package's SGD and the
network has a gradient arena."""
import os
import re
import time
from collections import defaultdict
from copy import deepcopy

import numpy as np
import torch

from .parallel import ArenaReducer, GraphedTrainStep, dp_info


def train_epoch(net, train_iter, loss, updater, device=torch.device("cpu"), runner=None, reducer=None):
    if isinstance(net, torch.nn.Module):
        net.train()
    last = None
    for X, y in train_iter:
        X, y = X.to(device, non_blocking=True), y.to(device, non_blocking=True)
        if runner is not None and (runner.X is None or (X.shape == runner.X.shape and y.shape == runner.y.shape)):
            last = runner(X, y)
            continue
        y_hat = net(X)
        l = loss(y_hat, y)
        updater.zero_grad()
        l.mean().backward()
        if hasattr(net, "cfg"):
            net.cfg.run_deferred()
            net.cfg.join_side()
        if reducer is not None:
            reducer.finish()                     # data parallel: the step waits for the averaged gradients
        updater.step()
        if hasattr(net, "cfg"):
            net.cfg.end_step()
        last = l.detach()
    if hasattr(net, "cfg") and str(getattr(device, "type", device)).startswith("cuda"):
        # once per epoch (a host read): a cooperative kernel whose grid was not co-resident has given up with wrong results
        from . import kernels as K
        if K.coop_timeouts():
            raise RuntimeError("a cooperative kernel (BatchNorm backward / weight-gradient combine) found its grid not co-resident: "
                               "the gradients of this epoch are wrong - lower EESEG_OPT_CONV_CUS (parallel.ArenaReducer reserve_cus)")
    return last


def train(net, train_iter, loss, num_epochs, updater, val_iter=None, metrics=None, patience=None, saveat=None,
          start_from=None, verbose=False, device="cpu", scheduler=None, use_file=None, up_updater=False,
          ret_lr=False, name=None, minimize=True, start_counting=0, use_graph=True, save_last=None, **kwargs):
    # data parallel (the reference's commented-out nn.DataParallel sits at train_funcs.py:72-74): every rank
    # runs this loop on its shard of each global batch; gradients are averaged by the ArenaReducer, validation counters
    # are summed over the ranks (mIoU_evaluator), so every rank takes the same early-stopping / scheduler decisions;
    # only rank 0 writes messages and checkpoints
    world, rank = dp_info(net)

    def say(msg):
        if not verbose or rank != 0:
            return
        if use_file:
            with open(use_file, "a") as f:
                f.write(msg + "\n")
        else:
            print(msg)

    follow = f"val_{metrics[0][0]}" if metrics else "val"
    tracker = defaultdict(list)
    net.to(device)
    name = name or "unspecified"
    counter = 0
    best_val = np.inf if minimize else 0.0
    saveat = saveat or os.path.join(".", "model.pth")
    say(f"<< {name} progress update >> Earlystopping " +
        (f"will follow {follow} with patience set to {patience}." if patience else "not set."))
    if start_from:
        save_dict = torch.load(start_from, weights_only=True)
        net.load_state_dict(save_dict["model_state_dict"])
        if up_updater:
            lr_aux = updater.param_groups[0]["lr"]
            updater.load_state_dict(save_dict["opt_state_dict"])
            updater.param_groups[0]["lr"] = lr_aux
        if patience and follow in save_dict:
            best_val = save_dict[follow]
    branchy = bool(kwargs.get("n_branches"))
    runner = reducer = None
    if world > 1 or (hasattr(net, "cfg") and net.cfg.dp_active()):
        if not hasattr(net, "cfg") or net.cfg.arena is None:
            raise RuntimeError("data-parallel training needs net.enable_grad_arena() (gradient buckets are arena slices)")
        reducer = ArenaReducer(net, group=net.cfg.group)
        use_graph = use_graph and net.cfg.collective is None        # the gloo test transport stages through the host
    if hasattr(net, "cfg") and net.cfg.arena is not None and hasattr(updater, "static_grads"):
        runner = GraphedTrainStep(net, loss, updater, reducer, use_graph=use_graph)
    epoch, last_lr = 0, 0
    have_ckpt = os.path.exists(saveat)       # read once, before anybody writes: every rank must take the same branch below
    num_epochs = num_epochs or np.inf
    while epoch < num_epochs:
        epoch += 1
        cur_lr = updater.state_dict()["param_groups"][-1 if branchy else 0]["lr"]
        start = time.perf_counter()
        say(f"<< {name} progress update >> starting #{epoch} training epoch; lr = {cur_lr}, "
            f"no updates since {counter} epochs")
        if hasattr(getattr(train_iter, "sampler", None), "set_epoch"):
            train_iter.sampler.set_epoch(epoch)
        train_epoch(net, train_iter, loss, updater, device, runner, reducer)
        end = time.perf_counter() - start
        say(f"<< {name} progress update >> finished #{epoch} training epoch after {end // 60} mins and "
            f"{end - 60 * (end // 60):.2f} s")
        if val_iter:
            was_training = net.training
            net.eval()                                       # B-3: the reference validates in train() mode
            with torch.no_grad():
                for met, f in metrics:
                    cur_res = f(net, net.n_branches + 1 if branchy else 1, kwargs["nout_channels"], val_iter, device)
                    if branchy:
                        for key, value in cur_res.items():
                            tracker[f"val_{met}_{key}"].append(value)
                    else:
                        tracker[f"val_{met}"].append(cur_res["mIoU"])
            net.train(was_training)
        if ret_lr or scheduler:
            tracker["lr"].append(cur_lr)
        if branchy:
            branch_val = [tracker[key][-1] for key in tracker if re.search(follow, key)]
            cur_val = float(np.average(branch_val)) if branch_val else 0.0
        else:
            cur_val = tracker[follow][-1] if tracker[follow] else 0.0
        if scheduler:
            # the reference calls step() without the metric (train_funcs.py:200-201), which ReduceLROnPlateau rejects;
            # a plateau scheduler gets the value early stopping follows
            if isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                scheduler.step(cur_val)
            else:
                scheduler.step()
            if hasattr(updater, "sync_lr"):
                updater.sync_lr()
        if save_last and rank == 0:          # optional (not in the reference): the weights after the epoch just finished
            torch.save({"model_state_dict": net.state_dict(), "epoch": epoch}, save_last.format(epoch=epoch))
        improved = (best_val > cur_val) if minimize else (best_val < cur_val)
        if patience and counter >= patience and epoch > start_counting:
            break
        if improved or not have_ckpt:
            save_dict = {"model_state_dict": deepcopy(net.state_dict()),
                         "opt_state_dict": deepcopy(updater.state_dict()), "epoch": epoch}
            for k in list(tracker.keys()):
                if k.startswith("val_"):
                    save_dict[k] = tracker[k][-1]
            if rank == 0:
                torch.save(save_dict, saveat)
            have_ckpt = True
            if improved:
                best_val = cur_val
                counter = 0
                msg = f"<< {name} progress update >> saved @ {epoch} epoch. Best score: {best_val:.5g}"
                if branchy:
                    msg += "\nFor each branch:\n\t" + "\n\t".join(f"b{i + 1} = {v:.5g}" for i, v in enumerate(branch_val))
                say(msg)
        elif "lr" in tracker and last_lr != cur_lr:
            counter = 1
            last_lr = cur_lr
        else:
            counter += 1
    if world > 1 and torch.distributed.is_initialized():
        torch.distributed.barrier()          # rank 0's last checkpoint is on disk before any rank reloads it
    return tracker
