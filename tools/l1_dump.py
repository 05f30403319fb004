"""dump the first-layer fused forward of the library selected by CLHIP_LIB (bitwise A/B of kernel variants). usage: l1_dump.py out.pt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsurvey_amd import ops
out = {}
for shape in ((200, 3, 64, 64, 64), (5, 3, 70, 12, 64), (3, 3, 64, 64, 128)):
    N, C, K, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (0.2 * torch.randn(K, C, 3, 3, generator=g)).cuda()
    b = (0.1 * torch.randn(K, generator=g)).cuda()
    y, i = ops.conv3x3_relu_pool_fwd(x, w, b)
    dyp = torch.randn(y.shape, generator=g).cuda() * (y > 0)
    dw, db = ops.conv3x3_bwd_weight_unpool(x, dyp, i)
    out[shape] = (y.cpu(), i.cpu(), dw.cpu(), db.cpu())
torch.save(out, sys.argv[1])
