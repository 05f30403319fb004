"""Per-layer timing of the bf16-split conv kernels (csrc/bsconv.hip) next to the Winograd f32 kernels on the 3x3 layers of the three
VGG9 widths at N = 200 (HIP events, 20 launches, best of 3): forward (+ fused ReLU / pool where the net has it), backward-data plain
and from the pooled gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops  # noqa: E402

LAYERS = [  # C, K, H, pool          (small / base / wide VGG9 at 64x64 inputs)
    (64, 64, 32, True), (64, 64, 16, False), (64, 64, 16, True), (64, 128, 8, False), (128, 128, 8, True),
    (64, 128, 16, False), (128, 128, 16, True), (128, 256, 8, False), (256, 256, 8, True),
    (64, 128, 32, True), (128, 256, 16, False), (256, 256, 16, True), (256, 512, 8, False), (512, 512, 8, True),
]


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    print("N = %d   (us: wino / bs; floor = bf16 pipe at 2.5 PF x 1/6)" % N)
    tot = {"wino": 0.0, "bs": 0.0}
    for C, K, H, pool in LAYERS:
        x = torch.randn(N, C, H, H, device=dev).relu_()
        w = torch.randn(K, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
        b = torch.randn(K, device=dev) * 0.1
        fl = 2.0 * 9 * C * K * H * H * N
        floor = fl * 6 / 2.5e15 * 1e6
        tw = timed(lambda: ops.conv3x3_wino_fwd(x, w, b, True, pool=pool))
        tb = timed(lambda: ops.conv3x3_bs_fwd(x, w, b, True, pool=pool))
        if pool:
            yp, idx = ops.conv3x3_bs_fwd(x, w, b, True, pool=True)
            dy = torch.randn_like(yp)
            dw_ = timed(lambda: ops.conv3x3_wino_bwd_data(dy, w, None, idx))
            db_ = timed(lambda: ops.conv3x3_bs_bwd_data(dy, w, None, idx))
        else:
            dy = torch.randn(N, K, H, H, device=dev)
            dw_ = timed(lambda: ops.conv3x3_wino_bwd_data(dy, w, x))
            db_ = timed(lambda: ops.conv3x3_bs_bwd_data(dy, w, x))
        tot["wino"] += tw + dw_
        tot["bs"] += tb + db_
        print("%3dx%3d@%2d %s  fwd %6.1f / %6.1f   bwd-data %6.1f / %6.1f   floor %5.1f   bs frac of pipe %.2f / %.2f"
              % (C, K, H, "pool" if pool else "    ", tw, tb, dw_, db_, floor, floor / tb, floor / db_))
    print("sum: wino %.1f us, bs %.1f us" % (tot["wino"], tot["bs"]))


if __name__ == "__main__":
    main()
