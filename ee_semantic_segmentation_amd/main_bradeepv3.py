#!/usr/bin/env python3
"""Train + test entry point with the reference's flags (main_bradeepv3.py:24-37 Lovasz loss,
main_bradeepv3_ce.py:121 cross entropy).  ``python -m ee_semantic_segmentation_amd.main_bradeepv3
-t resnet50 -n 1 -e 2 [--loss ce|lovasz] [--dim 256] [--classes 21] [--bf16]``"""
import argparse
import errno
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes), before the HIP runtime starts

import torch

from . import branchy_seg_losses as BSL
from .deepv3_funcs import eval_deepv3
from .get_seg_datasets import LoadDataset
from .my_pixelwise_xentropy import BrXEntropyLoss


def build_parser(default_loss):
    p = argparse.ArgumentParser(description="Evaluate branched deepv3.")
    p.add_argument("-t", "--type", type=str, default="resnet101")
    p.add_argument("-n", "--n_branches", type=int, default=0)
    p.add_argument("-N", "--Name", type=str, default="deep_v3_resnet101")
    p.add_argument("-p", "--print_file", type=str, default=None)
    p.add_argument("-e", "--num_epochs", type=int, default=0)
    p.add_argument("-l", "--lr", type=float, default=.01)
    p.add_argument("-m", "--min_lr", type=float, default=.0)
    p.add_argument("-L", "--base_lr", type=float, default=0)
    p.add_argument("-c", "--count_branches", action="store_true")
    p.add_argument("-s", "--skip", type=int, default=0)
    p.add_argument("-f", "--fine_tune", type=str, default="")
    # additions (not in the reference)
    p.add_argument("--loss", choices=["ce", "lovasz"], default=default_loss)
    p.add_argument("--dim", type=int, default=256)
    p.add_argument("--classes", type=int, default=21)
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--bf16", action="store_true")
    p.set_defaults(count_branches=False)
    return p


def init_distributed():
    """Under `python -m torch.distributed.run --nproc-per-node N -m ee_semantic_segmentation_amd.main_bradeepv3 ...`: bind
    this process to its GPU and join the job BEFORE any GPU call.  torch.distributed is the rendezvous (gloo); the
    collectives of the training step go over RCCL through libeeseg (parallel.init_data_parallel, called by eval_deepv3).
    -> (world, rank, local_rank)."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    return world, rank, local_rank


def main(default_loss="lovasz", argv=None):
    args = build_parser(default_loss).parse_args(argv)
    world, rank, local_rank = init_distributed()
    n_branches, lr = args.n_branches, args.lr
    base_lr = args.base_lr or (lr if n_branches else 0)
    dataset = "voc_seg"
    use_file = args.print_file or f"{dataset}_deepv3_msgs.txt"
    og_dir = os.getcwd()
    r_dir = os.path.join(og_dir, f"{dataset}_results")
    try:
        os.makedirs(r_dir, exist_ok=True)
    except OSError as err:
        if err.errno != errno.EEXIST:
            raise
    C = args.classes
    train_set, val_set, test_set = LoadDataset(args.dim, None, num_classes=C,
                                               sizes=(max(64, 2 * args.batch), 10, 10)).get_dataset(None, dataset)
    if args.loss == "ce":
        loss = BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=n_branches + 1)
    else:
        loss = BSL.LovaszSoftmax(classes="present", ignore=C, n_branches=n_branches)
    dts_info = {
        "device": torch.device("cuda", local_rank),
        "name": args.Name, "main_dir": og_dir, "n_procs": 1, "n_rep": 1, "res_dir": r_dir,
        "input_dim": args.dim, "train_set": train_set, "val_set": val_set, "test_set": test_set,
        "use_file": use_file, "def_prefetch": lambda x: 2, "def_nworkers": lambda x: 0,
        "metrics": ["mIoU"], "ch_es": None, "minimize": False, "n_branches": n_branches,
        "count_branches": args.count_branches, "lr": lr, "min_lr": args.min_lr, "base_lr": base_lr,
        "num_epochs": args.num_epochs, "batch_sizes": args.batch, "loss": loss, "use_scheduler": True,
        "nout_channels": C, "skip": args.skip, "fine_tune": args.fine_tune,
        "freeze_backbone": bool(args.fine_tune), "freeze_from": None, "weighted_lr": False,
        "branch_params": None, "type": args.type,
        "compute_dtype": torch.bfloat16 if args.bf16 else torch.float32,
    }
    ret = eval_deepv3(dts_info)
    msg = f"Finished training. model is saved @ {ret}"
    if rank == 0:
        with open(use_file, "a") as f:
            f.write(msg + "\n" + "-" * 20 + "\n")
        print(msg)
    if world > 1:
        torch.distributed.barrier()
    return ret


if __name__ == "__main__":
    main("lovasz")
