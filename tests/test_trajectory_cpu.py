"""The trajectory-separation machinery of tests/trajectory.py on the oracle alone (no GPU): shapes, determinism of the shared
problem, and the property the GPU test relies on — fp32 runs leave the fp64 run at rounding level and stay together."""
import torch


def test_oracle_runners_separate_at_rounding_level():
    import trajectory as T
    prob = T.make_problem(steps=20)
    again = T.make_problem(steps=20)
    assert all(torch.equal(a, b) for a, b in zip(prob["star"], again["star"]))
    assert all(torch.equal(a, b) for a, b in zip(prob["batches"], again["batches"]))
    r64 = T.run_oracle(prob, torch.float64, 2, 40.0, 1e-2, every=10)
    r32 = T.run_oracle(prob, torch.float32, 1, 40.0, 1e-2, every=10)
    r32p = T.run_oracle(prob, torch.float32, 2, 40.0, 1e-2, every=10, perturb=1e-7)
    assert [r["step"] for r in r64] == [1, 10, 20]
    s32, s32p = T.separation(r32, r64), T.separation(r32p, r64)
    assert 0 < s32[0] < 1e-6 and 0 < s32[1] < 1e-4 and 0 < s32p[1] < 1e-3 and s32[2] < 1e-2, (s32, s32p)
    text, seps = T.table({"fp64": r64, "a": r32, "b": r32p})
    assert seps["fp64"] == [0.0, 0.0, 0.0] and len(text.splitlines()) == 4
