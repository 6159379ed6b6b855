"""What a collective's resident workgroups cost the training step, on ONE GPU: a stand-in kernel (scripts/probes/cu_hog.hip: G
workgroups of 256 threads that hold their slots for T ms, like an RCCL kernel's channels waiting on their peers) runs on a side
stream beside the HIP-graph replay of the step, with the 256-tile conv kernels planned for all 256 CUs or for 256 - R
(EESEG_OPT_CONV_CUS + the weight-gradient grid: what ArenaReducer(reserve_cus=R) sets while buckets are in flight).

    python3 scripts/contention_probe.py --batch 4 --hog-ms 12

Prints one line per (planned CUs, hog workgroups): step ms.  The measured question: is there a cliff (a launch planned for 256
one-block-per-CU slots needing a second round) when a few CUs are held, and does planning for fewer CUs remove it."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def hog_lib():
    src = os.path.join(ROOT, "scripts", "probes", "cu_hog.hip")
    out = os.path.join(ROOT, "gpurun_out", "cu_hog.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", out])
    lib = C.CDLL(out)
    lib.hog_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    lib.hog_launch.restype = C.c_int
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--hog-ms", type=float, default=12.0)
    ap.add_argument("--hog-threads", type=int, default=256)
    ap.add_argument("--hog-lds-kb", type=int, default=0,
                    help="dynamic LDS of a hog workgroup: 0 = it fits beside a 256-tile conv block (136 KiB of 160), 32 = it does not")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cus", type=str, default="256,224")
    ap.add_argument("--hogs", type=str, default="0,8,16,32")
    a = ap.parse_args()
    from ee_semantic_segmentation_amd._lib import lib
    hog = hog_lib()
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    for cus in [int(v) for v in a.cus.split(",")]:
        assert lib().eeseg_set_option(8, cus) == 0
        assert lib().eeseg_set_wgrad_big_grid(cus, 8) == 0
        args = types.SimpleNamespace(overlap_wgrad=0, dp_transport="rccl", reserve_cus=0, no_graph=False)
        run = bench.Run("resnet101", 2, 19, 513, a.batch, "bf16", "ce", False, 1, 0, dev, args)
        for _ in range(5):
            run.step()
        torch.cuda.synchronize()
        for g in [int(v) for v in a.hogs.split(",")]:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
            for i in range(a.steps):
                evs[i].record()
                if g > 0:
                    side.wait_stream(torch.cuda.current_stream())
                    rc = hog.hog_launch(g, a.hog_threads, a.hog_lds_kb << 10, a.hog_ms, C.c_void_p(side.cuda_stream))
                    assert rc == 0, rc
                run.step()
                if g > 0:
                    torch.cuda.current_stream().wait_stream(side)
            evs[a.steps].record()
            torch.cuda.synchronize()
            per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
            print(f"B={a.batch} planned CUs {cus} hog {g:3d} workgroups ({a.hog_lds_kb} KiB LDS) x {a.hog_ms} ms: step median {per[len(per) // 2]:.2f} ms "
                  f"(min {per[0]:.2f})", flush=True)
        del run
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
