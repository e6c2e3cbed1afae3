"""ctypes binding of libshapegan_hip.so — the only door from Python into the HIP kernels — and of libshapegan_cpu.so, the
plain-C++ twin of the same C ABI (`sg_<name>_cpu`, csrc_cpu/shapegan_cpu.cpp; SURVEY.md 8b, BASELINE configs[0] "on CPU").

Dispatch is by the DEVICE OF THE TENSORS of a call and nothing else: GPU tensors go to the HIP library, CPU tensors to the
twin, a call that mixes them raises.  There is no fallback in either direction: a missing libshapegan_hip.so raises as soon
as a GPU tensor (or any size query) needs it, a missing twin raises when a CPU tensor shows up; neither library ever stands
in for the other, and no eager-PyTorch path exists.  Signatures mirror include/shapegan_hip.h one to one.
"""
import ctypes
import os
import threading
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHAPEGAN_HIP_LIB") or os.path.join(_HERE, "libshapegan_hip.so")  # override: A/B builds

ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4

_P, _I, _L, _F, _Z, _D = c_void_p, c_int, c_long, c_float, c_size_t, c_double

# name -> (restype, argtypes); every symbol include/shapegan_hip.h declares
SIGNATURES = {
    "sg_abi_version": (c_int, []),
    "sg_last_error": (c_char_p, []),
    "sg_conv3d_k4s2p1_fwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "sg_conv3d_k4s2p1_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _P]),
    "sg_conv3d_k4s2p1_fwd_keep": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _I, _P]),
    "sg_conv3d_k4s2p1_pack_images": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "sg_conv3d_k4s2p1_image_layout": (ctypes.c_longlong, [_I, _P, _Z]),
    "sg_conv3d_k4s2p1_fwd_impl": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _I, _I, _P]),
    "sg_conv3d_k4s2p1_dgrad_workspace_bytes": (_Z, [_I, _I]),
    "sg_conv3d_k4s2p1_dgrad_workspace_bytes_for": (_Z, [_I, _I, _I, _I, _I, _I]),
    "sg_conv3d_k4s2p1_dgrad": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _P]),
    "sg_conv3d_k4s2p1_dgrad_keep": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _I, _P]),
    "sg_conv3d_k4s2p1_dgrad_impl": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _I, _P]),
    "sg_conv3d_k4s2p1_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "sg_conv3d_k4s2p1_wgrad_impl": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _I, _P]),
    "sg_conv3d_k4s2p1_wgrad": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "sg_conv3d_k4s2p1_wgrad_act_eligible": (c_int, [_I, _I, _I, _I, _I, _I, _I]),
    "sg_conv3d_k4s2p1_wgrad_act": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _P]),
    "sg_convT3d_k4s2p1_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _P]),
    "sg_convT3d_k4s2p1_to1_pre_eligible": (c_int, [_I, _I, _I, _I, _I]),
    "sg_convT3d_k4s2p1_to1_pre": (c_int, [_P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _I, _I, _I, _F, _P]),
    "sg_convT3d_k4s2p1_to1_pre_impl": (c_int, [_P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "sg_convT3d_k4s2p1_to1_pre_grouped": (c_int, [_P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _I, _I, _I, _F, _I, _L, _P]),
    "sg_convT3d_k4s2p1_dgrad": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "sg_convT3d_k4s2p1_wgrad": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "sg_gemm_workspace_bytes": (_Z, [_I, _I]),
    "sg_gemm": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _F, _P, _Z, _P]),
    "sg_colsum": (c_int, [_P, _P, _I, _I, _L, _P]),
    "sg_rowsum": (c_int, [_P, _P, _L, _L, _L, _P]),
    "sg_rowsum_multi": (c_int, [_P, _P, _P, _I, _L, _L, _L, _P]),
    "sg_segsum": (c_int, [_P, _P, _L, _L, _P, _L, _P]),
    "sg_bn_workspace_bytes": (_Z, [_I]),
    "sg_bn_train_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _F, _F, _I, _F, _P, _Z, _P]),
    "sg_bn_train_stats": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _F, _F, _P, _Z, _P]),
    "sg_bn_train_fwd_grouped": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _F, _F, _I, _F, _P, _Z, _P]),
    "sg_bn_train_stats_grouped": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _F, _F, _P, _Z, _P]),
    "sg_bn_eval_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _F, _I, _F, _P]),
    "sg_bn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _F, _P, _Z, _P]),
    "sg_act_fwd": (c_int, [_P, _P, _L, _I, _F, _P]),
    "sg_act_bwd": (c_int, [_P, _P, _P, _L, _I, _F, _P]),
    "sg_act_bwd_rowsum": (c_int, [_P, _P, _P, _P, _L, _L, _I, _F, _P]),
    "sg_sdfnet_packed_floats": (_Z, [_I]),
    "sg_sdfnet_acts_floats": (_Z, [_L]),
    "sg_sdfnet_pack": (c_int, [_P, _I, _I, _P, _P]),
    "sg_sdfnet_fwd": (c_int, [_P, _L, _P, _P, _I, _P, _I, _P, _P, _L, _P, _P, _P, _L, _L, _P]),
    "sg_sdfnet_bwd_blocks": (c_long, [_L]),
    "sg_sdfnet_bwd_tile_start": (c_long, [_L, _L]),
    "sg_sdfnet_shape_bias": (c_int, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    "sg_sdfnet_pack_shape_bias": (c_int, [_P, _I, _P, _P, _L, _P, _P, _P]),
    "sg_sdfnet_shape_bias_bwd": (c_int, [_P, _P, _L, _P, _I, _P, _P, _P, _P, _P, _P, _F, _P]),
    "sg_sdfgen_acts_floats": (_Z, [_L]),
    "sg_sdfgen_packed_norm_offset": (_L, [_I]),
    "sg_sdfgen_pack": (c_int, [_P, _P, _P, _P]),
    "sg_sdfgen_fwd": (c_int, [_P, _P, _P, _P, _L, _P, _F, _P, _P, _L, _L, _P]),
    "sg_sdfgen_bwd_blocks": (_L, [_L]),
    "sg_sdfgen_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _P]),
    "sg_sdfgen_bwd_finish_workspace_bytes": (_Z, [_L]),
    "sg_sdfgen_bwd_finish": (c_int, [_P, _P, _L, _L, _P, _P, _P, _P, _L, _P, _L, _P, _P, _L, _P, _P, _P, _Z, _P, _P]),
    "sg_sdfnet_bwd_finish_workspace_bytes": (_Z, [_L]),
    "sg_sdfnet_bwd_finish": (c_int, [_P, _P, _L, _L, _I, _P, _P, _P, _P, _L, _P, _L, _P, _L, _P, _P, _P, _Z, _P, _P]),
    "sg_sdfnet_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _I, _L, _L, _P]),
    "sg_axpby": (c_int, [_P, _P, _P, _L, _F, _F, _P]),
    "sg_reduce_workspace_bytes": (_Z, []),
    "sg_reduce_sum": (c_int, [_P, _P, _L, _F, _P, _Z, _P]),
    "sg_gather_rows": (c_int, [_P, _P, _P, _L, _I, _P]),
    "sg_scatter_add_rows": (c_int, [_P, _L, _P, _P, _L, _I, _P]),
    "sg_sdf_batch_sort_max_shapes": (c_int, []),
    "sg_sdf_batch_sort_workspace_bytes": (_Z, [_L, _L]),
    "sg_sdf_batch_sort": (c_int, [_P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _Z, _P]),
    "sg_rmsprop_step": (c_int, [_P, _P, _P, _L, _F, _F, _F, _F, _F, _P]),
    "sg_adam_step": (c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _L, _F, _P]),
    "sg_adam_step_dev": (c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P, _F, _P]),
    "sg_adam_step_guarded": (c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _L, _F, _P, _P]),
    "sg_adam_step_dev_guarded": (c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P, _F, _P, _P]),
    "sg_adam_step_dev_multi": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sg_clamp": (c_int, [_P, _L, _F, _F, _P]),
    "sg_clamp_multi": (c_int, [_P, _P, _I, _F, _F, _P]),
    "sg_voxel_prepare": (c_int, [_P, _P, _L, _F, _F, _P]),
    "sg_gemm_nt_workspace_bytes": (_Z, [_I, _I, _L]),
    "sg_gemm_nt": (c_int, [_P, _L, _P, _L, _P, _L, _I, _I, _L, _P, _Z, _P]),
    "sg_gemm_nt_batched_workspace_bytes": (_Z, [_I, _I, _I, _L]),
    "sg_gemm_nt_batched": (c_int, [_P, _P, _L, _P, _P, _L, _P, _P, _P, _I, _I, _I, _L, _P, _Z, _P]),
    "sg_gemm_nt_batched_lnrelu": (c_int, [_P, _P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _P, _Z, _P]),
    "sg_layernorm_fwd": (c_int, [_P, _L, _P, _L, _P, _P, _P, _L, _P, _P, _L, _I, _F, _I, _P]),
    "sg_layernorm_bwd_workspace_bytes": (_Z, [_L, _I]),
    "sg_layernorm_bwd": (c_int, [_P, _L, _P, _L, _P, _P, _L, _P, _L, _P, _P, _P, _L, _P, _P, _L, _I, _I, _P, _Z, _P]),
    "sg_colsum_tall_workspace_bytes": (_Z, [_L, _L, _I]),
    "sg_colsum_tall": (c_int, [_P, _P, _L, _L, _L, _I, _L, _P, _Z, _P]),
    "sg_pointnet_packed_floats": (_Z, []),
    "sg_pointnet_pack": (c_int, [_P, _P, _P]),
    "sg_pointnet_select_workspace_bytes": (_Z, [_L, _L]),
    "sg_pointnet_select": (c_int, [_P, _P, _P, _L, _L, _P, _P, _P, _Z, _P]),
    "sg_segmax_workspace_bytes": (_Z, [_L, _L, _I]),
    "sg_segmax_fwd": (c_int, [_P, _P, _P, _L, _L, _I, _P, _Z, _P]),
    "sg_segmax_scatter": (c_int, [_P, _P, _P, _L, _L, _I, _P]),
    "sg_segmax_gather": (c_int, [_P, _P, _P, _L, _L, _I, _P]),
    "sg_scatter_rows_grouped": (c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "sg_rowdot": (c_int, [_P, _P, _P, _P, _L, _I, _I, _P]),
    "sg_rowscale": (c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "sg_rowouter": (c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "sg_loss_workspace_bytes": (_Z, []),
    "sg_loss_weighted_l1_fwd": (c_int, [_P, _P, _L, _F, _P, _P, _Z, _P]),
    "sg_loss_weighted_l1_bwd": (c_int, [_P, _P, _P, _P, _L, _F, _P]),
    "sg_loss_mean_split_fwd": (c_int, [_P, _L, _L, _F, _F, _P, _P, _P]),
    "sg_loss_mean_split_bwd": (c_int, [_P, _P, _L, _L, _F, _F, _P]),
    "sg_loss_bce_fwd": (c_int, [_P, _L, _F, _P, _P]),
    "sg_loss_bce_bwd": (c_int, [_P, _P, _P, _L, _F, _P]),
    "sg_loss_neg_mean_log_fwd": (c_int, [_P, _L, _P, _P]),
    "sg_loss_neg_mean_log_bwd": (c_int, [_P, _P, _P, _L, _P]),
    "sg_vae_reparam_fwd": (c_int, [_P, _P, _P, _P, _L, _P]),
    "sg_vae_reparam_bwd": (c_int, [_P, _P, _P, _P, _L, _P]),
    "sg_head_dot_fwd": (c_int, [_P, _P, _P, _P, _I, _L, _I, _F, _P]),
    "sg_head_dot_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "sg_conv3d_k4s2p1_wgrad_dy_image": (c_int, [_I, _I, _I, _I, _I, _I, _Z, _P, _P]),
    "sg_act_bwd_rowsum_pack8": (c_int, [_P, _P, _P, _P, _P, _L, _I, _L, _I, _F, _P]),
    "sg_conv3d_k4s2p1_wgrad_prepacked": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "sg_loss_kld_fwd": (c_int, [_P, _P, _L, _P, _P, _Z, _P]),
    "sg_loss_kld_bwd": (c_int, [_P, _P, _P, _P, _P, _L, _P]),
    "sg_loss_meansq_fwd": (c_int, [_P, _P, _L, _I, _D, _P, _P, _Z, _P]),
    "sg_loss_meansq_bwd": (c_int, [_P, _P, _P, _P, _L, _I, _D, _P]),
    "sg_loss_deepsdf_fwd": (c_int, [_P, _P, _L, _P, _P, _L, _I, _D, _P, _P, _Z, _P]),
    "sg_loss_deepsdf_bwd": (c_int, [_P, _P, _L, _P, _P, _L, _I, _D, _P, _P, _P, _P]),
    "sg_loss_deepsdf_fused": (c_int, [_P, _P, _L, _P, _P, _L, _I, _D, _P, _P, _P, _P, _Z, _P, _P]),
    "sg_count_sign_mismatch": (c_int, [_P, _P, _L, _P, _P, _Z, _P]),
    "sg_gradient_penalty_fwd": (c_int, [_P, _L, _L, _F, _P, _P, _P]),
    "sg_gradient_penalty_bwd": (c_int, [_P, _P, _P, _P, _L, _L, _F, _P]),
    "sg_lerp_rows": (c_int, [_P, _P, _P, _P, _L, _L, _P]),
    "sg_fade_blend": (c_int, [_P, _P, _P, _L, _I, _L, _F, _F, _P]),
    "sg_channel0": (c_int, [_P, _P, _L, _I, _L, _F, _P]),
    "sg_subsample2": (c_int, [_P, _P, _L, _I, _P]),
    "sg_subsample2_adjoint": (c_int, [_P, _P, _L, _I, _P]),
    "sg_act_bwd_dy": (c_int, [_P, _P, _P, _P, _L, _I, _P]),
    "sg_scatter_max_workspace_bytes": (_Z, [_L, _I]),
    "sg_scatter_max_fwd": (c_int, [_P, _P, _P, _P, _L, _L, _I, _P, _Z, _P]),
    "sg_scatter_max_scatter": (c_int, [_P, _P, _P, _L, _L, _I, _P]),
    "sg_scatter_max_gather": (c_int, [_P, _P, _P, _L, _L, _I, _P]),
}

# libshapegan_comm.so (RCCL gradient exchange; loaded only by data-parallel runs that ask for it)
COMM_PATH = os.path.join(_HERE, "libshapegan_comm.so")
COMM_SIGNATURES = {
    "sg_comm_last_error": (c_char_p, []),
    "sg_comm_bind": (c_int, [c_char_p]),
    "sg_comm_versions": (c_int, [_P, _P]),
    "sg_allreduce_unique_id_bytes": (_Z, []),
    "sg_allreduce_unique_id": (c_int, [_P, _Z]),
    "sg_allreduce_init": (c_int, [_P, _I, _I, _P, _Z, _I]),
    "sg_allreduce_info": (c_int, [_P, _P, _P, _P, _P]),
    "sg_allreduce_launch": (c_int, [_P, _P, _L, _P]),
    "sg_allreduce_wait": (c_int, [_P, _P]),
    "sg_allreduce_destroy": (c_int, [_P]),
}
_comm = None


def rccl_path():
    """The RCCL of THIS process: the one PyTorch ships and its "nccl" backend maps (torch/lib/librccl.so).  The C-ABI exchange binds
    the same file, so that both communicators of a rank live in one RCCL instance of one version (the library used to be
    linked against the ROCm toolchain's librccl with a RUNPATH into /opt/rocm-7.2.0 and took whichever librccl.so.1 the loader
    found first).  SG_RCCL_PATH overrides; None if torch carries none (the loader's default search is used then)."""
    override = os.environ.get("SG_RCCL_PATH")
    if override:
        return override
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else None


COMM_INFO = {}      # what load_comm() bound: {"rccl_path", "rccl_header_version", "rccl_runtime_version"}


def _version_string(code):
    return "%d.%d.%d" % (code // 10000, code // 100 % 100, code % 100) if code >= 10000 else str(code)


def load_comm():
    """Loads libshapegan_comm.so (once) and binds it to this process's RCCL (rccl_path()).  Raises if it has not been built, if the
    RCCL cannot be opened, or if its major version differs from the header the library was compiled against."""
    global _comm
    if _comm is None:
        if not os.path.exists(COMM_PATH):
            raise RuntimeError("libshapegan_comm.so is missing at %s — build it with `python -m shapegan_amd.build`" % COMM_PATH)
        lib = ctypes.CDLL(COMM_PATH)
        for name, (res, args) in COMM_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        path = rccl_path()
        if lib.sg_comm_bind(path.encode() if path else None) != 0:
            msg = lib.sg_comm_last_error()
            raise RuntimeError("shapegan_comm: cannot bind RCCL (%s): %s" % (path or "loader default", msg.decode() if msg else "?"))
        hv, rv = c_int(0), c_int(0)
        lib.sg_comm_versions(ctypes.byref(hv), ctypes.byref(rv))
        COMM_INFO.update(rccl_path=path or "librccl.so.1 (loader default)", rccl_header_version=_version_string(hv.value),
                         rccl_runtime_version=_version_string(rv.value))
        _comm = lib
    return _comm


def check_comm(rc, what=""):
    if rc != 0:
        msg = load_comm().sg_comm_last_error()
        raise RuntimeError("shapegan_comm %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


# Entry points WITHOUT a twin: size queries and layout helpers are host code of libshapegan_hip.so (callable without a GPU; the
# twin keeps its opaque buffers within those sizes), the *_impl variants force a particular HIP kernel (tests / tuning).
NO_TWIN = {n for n in SIGNATURES if n.endswith("_workspace_bytes") or n.endswith("_workspace_bytes_for") or n.endswith("_impl")} | {
    "sg_sdfgen_acts_floats", "sg_sdfgen_packed_norm_offset", "sg_sdfgen_bwd_blocks", "sg_pointnet_packed_floats",
    "sg_abi_version", "sg_last_error", "sg_sdfnet_packed_floats", "sg_sdfnet_acts_floats", "sg_sdfnet_bwd_blocks", "sg_sdfnet_bwd_tile_start", "sg_sdf_batch_sort_max_shapes",
    "sg_conv3d_k4s2p1_wgrad_act_eligible", "sg_convT3d_k4s2p1_to1_pre_eligible", "sg_conv3d_k4s2p1_wgrad_dy_image",
    "sg_conv3d_k4s2p1_image_layout"}
CPU_PATH = os.path.join(_HERE, "libshapegan_cpu.so")

_hip = None
_cpu = None
_lib = None


def _load_hip():
    global _hip
    if _hip is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libshapegan_hip.so is missing at %s — build it with `python -m shapegan_amd.build` "
                "(there is no CPU / eager fallback for the shapegan_amd hot path)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.sg_abi_version() != 8:
            raise RuntimeError("libshapegan_hip.so ABI version mismatch")
        _hip = lib
    return _hip


def load_cpu():
    """Loads libshapegan_cpu.so (once): `sg_<name>_cpu` for every entry point outside NO_TWIN, same argument lists."""
    global _cpu
    if _cpu is None:
        if not os.path.exists(CPU_PATH):
            raise RuntimeError("libshapegan_cpu.so is missing at %s — build it with `python -m shapegan_amd.build`; CPU tensors "
                               "are computed by the C++ twin only, never by an eager-PyTorch path" % CPU_PATH)
        lib = ctypes.CDLL(CPU_PATH)
        for name, (res, args) in SIGNATURES.items():
            if name not in NO_TWIN:
                fn = getattr(lib, name + "_cpu")
                fn.restype = res
                fn.argtypes = args
        lib.sg_cpu_last_error.restype = c_char_p
        _cpu = lib
    return _cpu


class _DeviceOfCallState(threading.local):
    """Which library the call being assembled ON THIS THREAD belongs to: set by ptr() / note_device() while the arguments are
    evaluated, consumed by the dispatcher when the call is made.  Thread-local: autograd runs GPU-node backwards on device worker
    threads while CPU-twin nodes and the main thread run at the same time, and nn.DataParallel
    (train_hybrid_progressive_gan.py:62-68) calls replicas from one Python thread per device; ctypes releases the GIL inside
    every library call."""
    kind = None


_DeviceOfCall = _DeviceOfCallState()


def _note(is_cuda):
    kind = "cuda" if is_cuda else "cpu"
    if _DeviceOfCall.kind is not None and _DeviceOfCall.kind != kind:
        _DeviceOfCall.kind = None
        raise RuntimeError("shapegan_amd: one call received both GPU and CPU tensors")
    _DeviceOfCall.kind = kind


def note_device(t):
    """For callers that pass `t.data_ptr()` arithmetic instead of ptr(t)."""
    _note(t.is_cuda)


def reset_call_state():
    """Drops a half-assembled call (an exception between ptr() and the library call would otherwise leave its device behind for
    the next call on this thread)."""
    _DeviceOfCall.kind = None


class _Dispatch(object):
    """`lib.sg_foo(...)`: the HIP entry for GPU tensors, `sg_foo_cpu` of the twin for CPU tensors."""

    def __getattr__(self, name):
        hip_fn = getattr(_load_hip(), name)
        if name in NO_TWIN:
            fn = hip_fn
        else:
            def fn(*args):
                kind, _DeviceOfCall.kind = _DeviceOfCall.kind, None
                if kind is None:
                    raise RuntimeError("shapegan_amd: %s was called without a tensor argument that names its device" % name)
                if kind == "cuda":
                    return hip_fn(*args)
                return getattr(load_cpu(), name + "_cpu")(*args)
        setattr(self, name, fn)
        return fn


def load():
    """The dispatching view of the two libraries.  Raises RuntimeError if libshapegan_hip.so has not been built."""
    global _lib
    if _lib is None:
        _load_hip()
        _lib = _Dispatch()
    return _lib


def check(rc, what=""):
    if rc != 0:
        msgs = [_load_hip().sg_last_error()]
        if _cpu is not None:
            msgs.append(_cpu.sg_cpu_last_error())
        raise RuntimeError("shapegan %s failed (%d): %s" % (what, rc, " | ".join(m.decode() for m in msgs if m)))


def ptr(t):
    """Address of a contiguous fp32 / int tensor (None -> NULL); records the tensor's device for the dispatcher."""
    if t is None:
        return None
    if not t.is_contiguous():
        _DeviceOfCall.kind = None      # the call being assembled is abandoned: leave nothing behind for the next one
        raise RuntimeError("shapegan_amd kernels need contiguous tensors")
    _note(t.is_cuda)
    return t.data_ptr()


def stream():
    """The current HIP stream handle (None for a call on CPU tensors)."""
    if _DeviceOfCall.kind == "cpu" or not torch.cuda.is_available():
        return None
    return torch.cuda.current_stream().cuda_stream


_workspaces = {}


def workspace(name, nbytes, device):
    """Caller-owned scratch, cached per (device, stream, name) and grown on demand."""
    if device.type != "cuda":
        key = ("cpu", threading.get_ident(), name)      # CPU-twin calls run synchronously on the calling thread
    else:
        key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, name)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


_tickets = {}


def tickets(name, device, count=16):
    """`count` zeroed 32-bit words, cached per (device, stream, name): the arrival counters of kernels that finish their
    reduction in the last workgroup to arrive.  Zeroed once; every such kernel leaves them zero."""
    if device.type != "cuda":
        key = ("cpu", threading.get_ident(), name)
    else:
        key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, name)
    buf = _tickets.get(key)
    if (buf is None or buf.numel() < count) and device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # a graph under capture runs on a stream of its own: take the words an eager call of the same kernel family made on this
        # device (a trainer's warm-up steps) instead of zeroing new ones inside the graph — that was a fill launch per replay.
        # Replays are ordered with the eager work of the stream they are launched on; the kernels leave the words at zero.
        for (dev_index, _, nm), cand in list(_tickets.items()):
            if dev_index == key[0] and nm == name and cand.numel() >= count:
                return cand
    if buf is None or buf.numel() < count:
        buf = torch.zeros(count, dtype=torch.int32, device=device)
        _tickets[key] = buf
    return buf


def f32c(t):
    """fp32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# Kernels that update parameters through raw pointers do not bump tensor._version; anything that caches a derived image of
# parameters (the SDFNet MFMA weight pack, a kept ConvTranspose weight image) keys on an epoch as well.  PARAM_EPOCH moves with
# every event that may have rewritten ANY parameter (graph replays, load_state_dict, clip_weights, a new optimizer); a
# flat-buffer optimizer's step moves only the epoch of ITS buffer (param_epoch_of), so the critic's five updates per WGAN unit
# do not invalidate the generator's images.  Every new value comes from one counter: an epoch is never seen twice.
PARAM_EPOCH = 0
_EPOCH_COUNTER = 0
_PARAM_RANGES = []      # [start, end, epoch] of every live flat parameter buffer


def bump_param_epoch(param_range=None):
    """param_range: what register_param_range returned (the stepping optimizer's buffer); None: every parameter."""
    global PARAM_EPOCH, _EPOCH_COUNTER
    _EPOCH_COUNTER += 1
    if param_range is None:
        PARAM_EPOCH = _EPOCH_COUNTER
    else:
        param_range[2] = _EPOCH_COUNTER


def register_param_range(start, nbytes):
    global _EPOCH_COUNTER
    _EPOCH_COUNTER += 1
    r = [start, start + nbytes, _EPOCH_COUNTER]
    _PARAM_RANGES.append(r)
    return r


def unregister_param_range(r):
    try:
        _PARAM_RANGES.remove(r)
    except ValueError:
        pass


def param_epoch_of_ptrs(ptrs):
    """The newest epoch that may have changed any of the parameters at the addresses `ptrs`: the global one, or that of a flat
    buffer holding one of them."""
    e = PARAM_EPOCH
    for r in _PARAM_RANGES:
        if r[2] > e:
            lo, hi = r[0], r[1]
            for a in ptrs:
                if lo <= a < hi:
                    e = r[2]
                    break
    return e


def param_epoch_of(*tensors):
    return param_epoch_of_ptrs([t.data_ptr() for t in tensors])


def writers_known(*tensors):
    """True when every tensor lives in a registered flat parameter buffer (shapegan_amd.optim): then everything that writes it is
    ours and announces itself — the optimizer's step, clip_weights, load_state_dict, broadcast_parameters and graph replays all
    move a parameter epoch.  A weight outside such a buffer may be written through `.data` (the reference's own idiom:
    `parameter.data.clamp_(...)`, model/gan.py:67-69; EMA copies; manual loading), which moves neither `tensor._version` nor an
    epoch — images derived from it must not be reused.  A `.data` writer of a flat-buffer parameter calls
    `ops.invalidate_weight_images()`."""
    for t in tensors:
        a = t.data_ptr()
        for r in _PARAM_RANGES:
            if r[0] <= a < r[1]:
                break
        else:
            return False
    return True


# ---- gradient destinations ---------------------------------------------------------------------------------------------
# A flat-buffer optimizer (shapegan_amd.optim) registers, per parameter, the slice of its flat gradient buffer that
# belongs to it.  In a plain backward (no create_graph) the weight-gradient kernels write straight into that slice and hand
# autograd a view of it: AccumulateGrad adopts the view as p.grad (no `p.grad += g` launch, no copy), post-accumulate hooks
# fire as usual.  Only the FIRST contribution of a zero_grad() cycle may do this (a second write would clobber the first);
# later contributions come back as ordinary tensors and autograd adds them in place — into the same slice.
import weakref  # noqa: E402


class GradSlot(object):
    __slots__ = ("param", "flat_grad", "offset", "numel", "written")

    def __init__(self, param, flat_grad, offset):
        self.param = weakref.ref(param)
        self.flat_grad, self.offset, self.numel = flat_grad, offset, param.numel()
        self.written = False


GRAD_SLOTS = {}


def register_grad_slot(param, flat_grad, offset):
    slot = GradSlot(param, flat_grad, offset)
    GRAD_SLOTS[param.data_ptr()] = slot
    return slot


def unregister_grad_slots(slots):
    """Called when the flat buffer that owns `slots` goes away (optim._Flat.__del__): the registry must not keep a discarded
    optimizer's gradient buffer alive until its addresses happen to be reused."""
    for key in [k for k, v in GRAD_SLOTS.items() if any(v is s for s in slots)]:
        GRAD_SLOTS.pop(key, None)


_ONES = {}
_direct_write_depth = 0      # > 0 while a backward started through `backward()` below is running (process-wide: the engine
                             # runs GPU nodes on its own worker threads, so this cannot be thread-local)


def backward(loss, **kwargs):
    """`loss.backward(**kwargs)` with direct gradient writes enabled: inside it, weight-gradient kernels may store straight into
    the flat-gradient slices registered by shapegan_amd.optim and autograd adopts those views as p.grad.  The trainers call
    this.  A backward started any other way — a plain `loss.backward()`, and in particular `torch.autograd.grad(loss, params)`,
    which RETURNS gradients to a caller who may keep them — always receives ordinary tensors, as in torch (a
    returned view of the flat slice would be overwritten by the next backward)."""
    global _direct_write_depth
    if "gradient" not in kwargs and loss.dim() == 0:
        # the engine would create the implicit d loss / d loss = 1 with a fill launch on every call: keep one per device / dtype
        key = (loss.device, loss.dtype)
        one = _ONES.get(key)
        if one is None:
            one = _ONES[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
        kwargs["gradient"] = one
    _direct_write_depth += 1
    try:
        loss.backward(**kwargs)
    finally:
        _direct_write_depth -= 1


def is_unit_gradient(g):
    """True when `g` is the constant 1 that `backward()` above seeds a scalar loss with (so d loss / d loss = 1 exactly)."""
    one = _ONES.get((g.device, g.dtype))
    return one is not None and g.dim() == 0 and g.data_ptr() == one.data_ptr()


def grad_destination(param_tensor, shape):
    """A view (of `shape`) of the flat-gradient slice of the parameter whose storage `param_tensor` is, if this backward
    may write its gradient there directly; else None."""
    if _direct_write_depth == 0:
        return None
    slot = GRAD_SLOTS.get(param_tensor.data_ptr())
    if slot is None or slot.written or torch.is_grad_enabled():
        return None
    p = slot.param()
    if p is None or p.data_ptr() != param_tensor.data_ptr():
        GRAD_SLOTS.pop(param_tensor.data_ptr(), None)      # the parameter is gone (its address was reused)
        return None
    n = 1
    for d in shape:
        n *= d
    if p.grad is not None or n != slot.numel:
        return None
    slot.written = True
    return slot.flat_grad[slot.offset:slot.offset + n].view(shape)
