"""Non-default sides of the switches added in the second session of round 6, each in a process of its own (the library reads them
once): CLHIP_FC_TAIL_ROWS (rows per workgroup of fc_tail_kernel: results must not depend on it — the bitwise test against the
per-layer classifier path runs at 32 and 8 rows as it does at the default 16) and CLHIP_WG_SMALLC_BLOCKS (block count of the first
layer's weight-gradient launch, a tuning switch: another split of the stage list, same tolerance against the unfused kernels)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(env_extra, target, keyword):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, target), "-q", "-x", "-p", "no:cacheprovider", "-k", keyword],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-2500:], r.stderr[-1500:])


@pytest.mark.timeout(1000)
@pytest.mark.parametrize("rows", ["32", "8"])
def test_fc_tail_rows_per_workgroup(rows):
    _run({"CLHIP_FC_TAIL_ROWS": rows}, "test_gpu_fc_tail.py", "bitwise_the_per_layer")


@pytest.mark.timeout(1000)
def test_first_layer_weight_gradient_block_count():
    _run({"CLHIP_WG_SMALLC_BLOCKS": "768"}, "test_gpu_parity.py", "wgrad_fused_unpool")
