"""Test infrastructure shared by the GPU parity tests: reading the executor's non-linear decisions back; the test-only
device library."""
import ctypes as C
import os

_DBG = None


def dbg_lib():
    """tests/libclhip_dbg.so (naive triage convolutions + MFMA fragment probe, tests/csrc/debug_naive.hip)."""
    global _DBG
    if _DBG is None:
        from clsurvey_amd import _lib, build
        _lib.lib()                                   # maps torch's HIP runtime first (see _lib.lib)
        path = build.TEST_LIB if os.path.exists(build.TEST_LIB) else build.build_test_lib(verbose=False)
        h = C.CDLL(os.path.abspath(path))
        p, i = C.c_void_p, C.c_int
        for name, args in (("clhip_dbg_conv3x3_fwd", [p, p, p, p, i, i, i, i, i, i, p]),
                           ("clhip_dbg_conv3x3_bwd_data", [p, p, p, p, i, i, i, i, i, p]),
                           ("clhip_dbg_conv3x3_bwd_weight", [p, p, p, p, i, i, i, i, i, p]),
                           ("clhip_dbg_mfma_probe", [p, p])):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = i, args
        _DBG = h
    return _DBG


def live_codes(idx):
    """Arg-max bytes of the fused conv + ReLU + pool kernels -> window positions: code 4 (CLHIP_POOL_DEAD, the window's
    maximum after ReLU is not positive, no gradient goes through it) reads as position 0, the arg-max ATen reports for an
    all-zero window; the ReLU mask of the same block switches the window off in the forced-branch oracles."""
    return idx.where(idx < 4, idx.new_zeros(()))


def engine_decisions(eng, n):
    """The ReLU masks and max-pool arg-max codes of the executor's last forward, one dict per plan layer that has a
    ReLU, shaped for oracle.alexnet_ref.forward_forced.  (The saved activation behind a Dropout is already masked: a
    dropped element reads as 'off', which is the same product.)"""
    out = []
    c, h, w = eng.in_shape
    for li, (kind, m, relu, pool) in enumerate(eng.layers):
        if kind == "conv":
            ks, st, pd = m.kernel_size[0], m.stride[0], m.padding[0]
            c, h, w = m.out_channels, (h + 2 * pd - ks) // st + 1, (w + 2 * pd - ks) // st + 1
            if pool:
                pk, ps = pool if isinstance(pool, tuple) else (2, 2)
                h, w = (h - pk) // ps + 1, (w - pk) // ps + 1
            shape = (n, c, h, w)
        else:
            c, h, w = m.out_features, 1, 1
            shape = (n, c)
        if not relu:
            assert li == len(eng.layers) - 1
            continue
        d = {"mask": (eng.layer_input(li + 1, n).view(shape) > 0).cpu()}
        if li + 1 in eng._masks:              # a Dropout sits between this block and the next layer
            d["dropped"] = (eng._masks[li + 1].cpu() == 0).expand(n, -1).reshape(shape)
        if pool:
            d["idx"] = eng.pool_idx(li, n).view(shape).cpu().long()
            if (pool if isinstance(pool, tuple) else (2, 2)) == (2, 2):     # only the fused 2x2 kernels write the dead code
                d["idx"] = live_codes(d["idx"])
        out.append(d)
    return out
