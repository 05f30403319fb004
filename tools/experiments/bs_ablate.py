"""Build timing-only ablation variants of csrc/bsconv.hip (BS_ABL bit mask, see the file) as libclhip_bsablN.so — run on the build
box (no GPU needed); tools/experiments/r05_d.sh times them on the GPU box through CLHIP_LIB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clsurvey_amd import build  # noqa: E402

for abl in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 6, 14, 46, 47]:
    print(build.build_variant("bsabl%d" % abl, "bsconv.hip", ["BS_ABL=%d" % abl], verbose=False))
