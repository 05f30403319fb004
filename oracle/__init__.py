"""CPU oracle for the CLsurvey training + importance-weight hot path.

TEST INFRASTRUCTURE ONLY. Nothing in the product package (clsurvey_amd/) may
import this. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import it — as the checker (or the timed CPU baseline), never as the thing
shipped or measured as the GPU path.

The reference (Mattdl/CLsurvey) is 100 % Python on torch, so the restatement is
written with torch-CPU fp32 tensor ops (float work) and numpy (uint8 mask / index
work). Every function cites the reference file:line it restates.

Pinning: tests/golden/*.npz were produced by importing the real reference in the
dev container (tests/golden/make_golden.py); tests/test_oracle_golden.py checks
this oracle against them. GEM's QP (quadprog==0.1.6, not vendored in the
reference) is restated from the published Goldfarb-Idnani algorithm and is
"parity unpinned" (checked by KKT residuals only) — see oracle/gem_ref.py.
"""
