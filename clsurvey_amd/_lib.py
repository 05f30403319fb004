"""ctypes binding of libclhip.so (include/clhip.h).  The product path has NO CPU fallback:
if the library is missing or a call fails this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclhip.so")

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_z = C.c_size_t
_l = C.c_long


class HatParam(C.Structure):                       # clhip_hat_param
    _fields_ = [("theta", _p), ("grad", _p), ("buf", _p), ("mask_back", _p), ("n", _z), ("is_emb", _i), ("reserved", _i)]


class HatGateJob(C.Structure):                     # clhip_hat_gate_job
    _fields_ = [("emb_row", _p), ("gate", _p), ("mask_pre", _p), ("n", _i), ("reserved", _i)]


class HatLayer(C.Structure):                       # clhip_hat_layer
    _fields_ = [("w", _p), ("gate_in", _p), ("out", _p), ("K", _z), ("C", _z), ("R", _z)]


class HatWgradJob(C.Structure):                    # clhip_hat_wgrad_job
    _fields_ = [("g", _p), ("w", _p), ("gate_in", _p), ("dgate_in", _p), ("K", _i), ("C", _i), ("R", _i), ("reserved", _i)]


class HatEmbJob(C.Structure):                      # clhip_hat_emb_job
    _fields_ = [("dgate", _p), ("gate", _p), ("mask_pre", _p), ("demb", _p), ("n", _i), ("rows", _i), ("t", _i), ("reserved", _i)]


class LayerDesc(C.Structure):
    _fields_ = [("type", _i), ("cin", _i), ("cout", _i), ("relu", _i), ("pool", _i),
                ("w_off", _l), ("b_off", _l), ("ksize", _i), ("stride", _i), ("pad", _i), ("pool_k", _i), ("pool_s", _i),
                ("bn", _i), ("bn_w_off", _l), ("bn_b_off", _l), ("has_drop", _i)]


# name -> (restype, argtypes); mirrors include/clhip.h one to one
SIGNATURES = {
    "clhip_version": (_i, []),
    "clhip_arch": (C.c_char_p, []),
    "clhip_conv3x3_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "clhip_conv3x3_relu_pool_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_conv3x3_bwd_weight_unpool": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv3x3_bwd_data": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_conv3x3_bwd_data_unpool": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_conv3x3_bwd_weight_ws": (_z, [_i, _i, _i, _i, _i]),
    "clhip_conv3x3_bwd_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv3x3_bwd_weight_slabs": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, C.POINTER(_i), _p]),
    "clhip_conv3x3_bwd_weight_reduce": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clhip_conv3x3_wino_ws": (_z, [_i, _i]),
    "clhip_conv3x3_wino_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv3x3_wino_bwd_data": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv5x5_bs_ws": (_z, [_i, _i]),
    "clhip_conv5x5_bs_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv5x5_bs_bwd_data": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv3x3_bs_ws": (_z, [_i, _i]),
    "clhip_conv3x3_bs_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv3x3_bs_bwd_data": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv3x3_wino_bwd_weight_ws": (_z, [_i, _i, _i, _i, _i]),
    "clhip_conv3x3_wino_bwd_weight": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_conv3x3_wino_bwd_ws": (_z, [_i, _i, _i, _i, _i]),
    "clhip_conv3x3_wino_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_maxpool2_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clhip_maxpool2_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clhip_fc_ws": (_z, [_i, _i, _i]),
    "clhip_fc_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_fc_bwd_data": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "clhip_fc_bwd_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "clhip_relu_bwd": (_i, [_p, _p, _p, _z, _p]),
    "clhip_softmax_ce": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "clhip_mse_zero_sum": (_i, [_p, _z, _p, _p, _p]),
    "clhip_reg_sgd_step": (_i, [_p, _p, _p, _p, _p, _z, _f, _f, _f, _f, _i, _p]),
    "clhip_fisher_accum": (_i, [_p, _p, _z, _f, _p]),
    "clhip_mas_accum": (_i, [_p, _p, _z, _f, _f, _p]),
    "clhip_si_step": (_i, [_p, _p, _p, _p, _p, _p, _z, _f, _f, _f, _f, _i, _p]),
    "clhip_si_consolidate": (_i, [_p, _p, _p, _p, _z, _f, _p]),
    "clhip_maxpool_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_maxpool_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clhip_conv2d_bwd_weight_ws": (_z, [_i] * 9),
    "clhip_conv2d_fwd": (_i, [_p, _p, _p, _p] + [_i] * 10 + [_p]),
    "clhip_conv2d_bwd_data": (_i, [_p, _p, _p, _p] + [_i] * 9 + [_p]),
    "clhip_conv2d_bwd_weight": (_i, [_p, _p, _p, _p] + [_i] * 9 + [_p, _z, _p]),
    "clhip_conv3x3_bs_bwd_weight_ws": (_z, [_i] * 5),
    "clhip_conv3x3_bs_bwd_weight": (_i, [_p] * 5 + [_i] * 5 + [_p, _z, _p]),
    "clhip_conv5x5_bs_bwd_weight_ws": (_z, [_i] * 5),
    "clhip_conv5x5_bs_bwd_weight": (_i, [_p] * 4 + [_i] * 5 + [_p, _z, _p]),
    "clhip_conv2d_s2d_ws": (_z, [_i] * 8),
    "clhip_conv2d_s2d_fwd": (_i, [_p, _p, _p, _p] + [_i] * 9 + [_p, _z, _p]),
    "clhip_conv2d_s2d_bwd_weight": (_i, [_p, _p, _p, _p] + [_i] * 8 + [_p, _z, _p]),
    "clhip_imm_merge": (_i, [_p, _p, _p, _i, _z, _p, _p]),
    "clhip_lwf_loss": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p, _p, _p, _p]),
    "clhip_packnet_finetune_mask": (_i, [_p, _z, _i, _p]),
    "clhip_packnet_kth_ws": (_z, []),
    "clhip_packnet_kth_abs": (_i, [_p, _p, _z, _i, _z, _p, _p, _z, _p]),
    "clhip_packnet_prune": (_i, [_p, _p, _z, _i, _p, _p]),
    "clhip_mask_grad_zero": (_i, [_p, _p, _z, _i, _p]),
    "clhip_mask_weight_zero": (_i, [_p, _p, _z, _i, _i, _p]),
    "clhip_packnet_sgd_step": (_i, [_p, _p, _p, _p, _z, _i, _f, _f, _f, _i, _p]),
    "clhip_hat_gate": (_i, [_p, _i, _f, _p, _p]),
    "clhip_hat_scale_weight": (_i, [_p, _p, _p, _z, _z, _z, _p]),
    "clhip_hat_weight_grad": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "clhip_hat_emb_grad": (_i, [_p, _p, _p, _i, _f, _f, _p, _p]),
    "clhip_hat_reg_sums": (_i, [_p, _p, _i, _p, _p]),
    "clhip_hat_backmask": (_i, [_p, _p, _p, _z, _z, _z, _p]),
    "clhip_hat_sgd_ws": (_z, []),
    "clhip_hat_sgd_step": (_i, [_p, _p, _p, _p, _z, _f, _f, _f, _i, _i, _f, _f, _f, _f, _i, _p, _z, _p]),
    "clhip_clamp": (_i, [_p, _z, _f, _f, _p]),
    "clhip_hat_sgd_multi_ws": (_z, [_i]),
    "clhip_hat_sgd_step_multi": (_i, [_p, _i, _f, _f, _f, _i, _f, _f, _f, _f, _f, _i, _p, _z, _p]),
    "clhip_hat_gates_multi": (_i, [_p, _i, _f, _p, _p]),
    "clhip_hat_scale_weights_multi": (_i, [_p, _i, _p]),
    "clhip_hat_weight_grads_multi": (_i, [_p, _i, _p]),
    "clhip_hat_emb_grads_multi": (_i, [_p, _i, _f, _f, _f, _p, _p]),
    "clhip_softmax_ce_slice": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "clhip_net_loss_step_slice": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "clhip_axpy": (_i, [_p, _p, _z, _f, _i, _p]),
    "clhip_gem_gram_ws": (_z, [_i]),
    "clhip_gem_gram": (_i, [_p, _z, C.POINTER(_i), _i, _z, _p, _p, _z, _p]),
    "clhip_gem_project": (_i, [_p, _z, C.POINTER(_i), C.POINTER(_f), _i, _p, _p, _z, _p]),
    "clhip_gem_qp": (_i, [_p, _i, C.c_double, C.c_double, _p, _p, _p]),
    "clhip_gem_project_dev": (_i, [_p, _z, C.POINTER(_i), _p, _p, _i, _p, _p, _z, _p]),
    "clhip_net_create": (_i, [C.POINTER(LayerDesc), _i, _i, _i, _i, _i, C.POINTER(_p)]),
    "clhip_net_set_dropout": (_i, [_p, _i, _p, _l]),
    "clhip_net_set_bn": (_i, [_p, _i, _p, _p, _f, _f]),
    "clhip_net_set_training": (_i, [_p, _i]),
    "clhip_net_probe": (_i, [_p, _i]),
    "clhip_net_probe_read": (_i, [_p, C.POINTER(_f), C.POINTER(_i)]),
    "clhip_net_layer_input": (_i, [_p, _i, _p, _p]),
    "clhip_net_layer_pool_idx": (_i, [_p, _i, _p, _p]),
    "clhip_net_layer_paths": (_i, [_p, _i]),
    "clhip_net_probe_kind": (_i, [_p, _i, _i]),
    "clhip_net_set_input_grad": (_i, [_p, _i, _p]),
    "clhip_sigmoid_fwd": (_i, [_p, _p, _z, _p]),
    "clhip_sigmoid_bwd": (_i, [_p, _p, _p, _z, _p]),
    "clhip_mse_mean": (_i, [_p, _p, _z, _f, _p, _p, _p]),
    "clhip_adadelta_step": (_i, [_p, _p, _p, _p, _z, _f, _f, _f, _f, _p]),
    "clhip_bn_ws": (_z, [_i]),
    "clhip_bn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p, _z, _p]),
    "clhip_bn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "clhip_net_destroy": (None, [_p]),
    "clhip_net_workspace_bytes": (_z, [_p]),
    "clhip_net_num_classes": (_i, [_p]),
    "clhip_net_forward": (_i, [_p, _p, _p, _i, _p, _p, _p]),
    "clhip_net_backward": (_i, [_p, _p, _p, _p, _i, _p, _p, _p]),
    "clhip_net_loss_step": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle. Raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        path = os.environ.get("CLHIP_LIB", LIB_PATH)     # CLHIP_LIB: experimental variants (tools/)
        if not os.path.exists(path):
            raise RuntimeError(
                "clsurvey_amd: %s not found. Build it with `python clsurvey_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        # torch's ROCm wheel bundles its own libamdhip64; it MUST be the one (and only) HIP runtime in
        # the process, so make sure it is mapped before libclhip's DT_NEEDED libamdhip64.so.7 resolves
        # (otherwise /opt/rocm's copy is loaded as a second runtime and launches fail with
        # hipErrorNoDevice on pointers owned by torch's runtime).
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        h = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)      # AttributeError => symbol missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


class ClhipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        if rc > 0:
            raise ClhipError("%s: HIP error %d" % (what, rc))
        names = {-1: "CLHIP_EINVAL", -2: "CLHIP_ENOSPC", -3: "CLHIP_ENOTSUP"}
        raise ClhipError("%s: %s" % (what, names.get(rc, rc)))
