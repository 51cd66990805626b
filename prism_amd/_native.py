"""ctypes binding of ``libprism_hip.so`` (C ABI in ``include/prism_hip.h``).

There is deliberately NO fallback: if the shared library is missing or a symbol does not resolve
the import of the product path fails loudly (``NativeLibraryError``) — nothing here ever routes to a
CPU or eager-PyTorch implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRISM_HIP_LIB") or os.path.join(_HERE, "libprism_hip.so")   # (override: kernel experiments)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "prism_hip.h")

PRISM_OK, PRISM_ERR_INVALID, PRISM_ERR_HIP, PRISM_ERR_UNSUPPORTED = 0, -1, -2, -3
PRISM_MAX_NSTEP = 15
FLAG_DONE, FLAG_TRUNC, FLAG_HAS_NEXT = 1, 2, 4
STATUS_NONPOSITIVE_PSUM, STATUS_NONPOSITIVE_PMIN = 1, 2
WS_STATUS_WORD, WS_STATUS_BARRIER_TIMEOUT, WS_STATUS_COLLECTIVE_TIMEOUT = 7, 1, 2
GEMM_MODES = {"auto": 0, "fp32": 1, "bf16x3": 2}
ACT_WEIGHTS_CURRENT = 1

c_i32, c_i64, c_u64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_void_p


class NativeLibraryError(RuntimeError):
    pass


class PrismError(RuntimeError):
    pass


class ReplayDesc(ctypes.Structure):
    _fields_ = [("capacity", c_i64), ("tree_capacity", c_i64), ("obs_elems", c_i32), ("n_step", c_i32),
                ("obs", c_vp), ("succ_obs", c_vp), ("reward", c_vp), ("action", c_vp), ("flags", c_vp),
                ("link", c_vp), ("back", c_vp), ("tree", c_vp), ("per_state", c_vp),
                ("status", c_vp), ("gammas", ctypes.c_double * (PRISM_MAX_NSTEP + 1))]


class ModelDims(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in (
        "in_channels", "n_actions", "embed_dim", "use_iqn", "n_basis", "iqn_layers", "iqn_width", "n_tau",
        "n_tau_next", "use_layer_norm", "n_heads", "head_layers", "head_width", "has_target", "double_q",
        "propagate_grad")] + [(n, c_f32) for n in ("huber_k", "dist_loss_weight", "q_loss_weight", "theil_coef")] + \
        [("squish_fn", c_i32)]


class ParamOffsets(ctypes.Structure):
    _fields_ = [(n, c_i64) for n in (
        "n_params", "conv_w", "conv_b", "phi_w", "phi_b", "iqn_ln1_g", "iqn_ln1_b", "iqn_w1", "iqn_b1",
        "iqn_ln2_g", "iqn_ln2_b", "iqn_w2", "iqn_b2", "head_base", "head_stride", "h_ln1_g", "h_ln1_b", "h_w1",
        "h_b1", "h_ln2_g", "h_ln2_b", "h_w2", "h_b2")]


class AdamHyper(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("lr", "beta1", "beta2", "eps")] + \
               [(n, c_f32) for n in ("max_grad_norm", "grad_scale")]


class LearnerDesc(ctypes.Structure):
    _fields_ = [("dims", ModelDims), ("off", ParamOffsets), ("batch", c_i32), ("embed_done", c_i32),
                ("params", c_vp), ("target_params", c_vp), ("grads", c_vp), ("adam_m", c_vp), ("adam_v", c_vp),
                ("adam_step", c_vp),
                ("obs", c_vp), ("next_obs", c_vp), ("reward", c_vp), ("nonterminal", c_vp), ("gamma", c_vp),
                ("action", c_vp), ("per_weights", c_vp),
                ("tau_cur", c_vp), ("tau_next_online", c_vp), ("tau_next_target", c_vp), ("tau_out", c_vp),
                ("seed", c_u64), ("offset", c_u64), ("rng_counters", c_vp),
                ("fused_replay", c_vp), ("fused_index", c_vp), ("fused_alpha", c_f32), ("fused_eps", c_f32),
                ("fuse_tail", c_i32), ("act_flags", c_i32), ("gemm_mode", c_i32),
                ("out_dist_loss", c_vp), ("out_q_loss", c_vp), ("out_td", c_vp), ("out_scalars", c_vp),
                ("dbg_z", c_vp), ("dbg_stamps", c_vp), ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
                ("hyper", AdamHyper), ("host_status", c_vp)]


MAX_PEERS = 8
DIRECT_FLAG_WORDS, IPC_HANDLE_BYTES = MAX_PEERS + 2, 64


class DirectDesc(ctypes.Structure):
    _fields_ = [("world", c_i32), ("rank", c_i32), ("bufs", c_vp * MAX_PEERS), ("flags", c_vp * MAX_PEERS), ("n", c_i64),
                ("poison", c_vp), ("host_status", c_vp), ("wait_seconds", ctypes.c_double)]


_P = ctypes.POINTER
# name -> (restype, argtypes); must list every symbol declared in include/prism_hip.h
SIGNATURES = {
    "prism_last_error": (ctypes.c_char_p, []),
    "prism_abi_version": (ctypes.c_int, []),
    "prism_device_info": (ctypes.c_int, [ctypes.c_int, _P(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]),
    "prism_replay_init": (ctypes.c_int, [_P(ReplayDesc), c_vp]),
    "prism_replay_insert": (ctypes.c_int, [_P(ReplayDesc), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_f32, c_f32, c_vp]),
    "prism_per_sample": (ctypes.c_int, [_P(ReplayDesc), c_i64, c_i32, c_vp, c_u64, c_u64, c_f32, c_vp, c_vp,
                                         c_vp]),
    "prism_uniform_sample": (ctypes.c_int, [c_i64, c_i32, c_u64, c_u64, c_vp, c_vp]),
    "prism_replay_gather": (ctypes.c_int, [_P(ReplayDesc), c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_vp]),
    "prism_per_update": (ctypes.c_int, [_P(ReplayDesc), c_vp, c_vp, c_i32, c_f32, c_f32, c_i32, c_vp]),
    "prism_per_rebuild": (ctypes.c_int, [_P(ReplayDesc), c_vp]),
    "prism_per_query": (ctypes.c_int, [_P(ReplayDesc), c_i64, c_vp, c_vp]),
    "prism_learner_workspace_bytes": (ctypes.c_size_t, [_P(ModelDims), c_i32]),
    "prism_learner_supported": (ctypes.c_int, [_P(ModelDims), c_i32]),
    "prism_learner_fwd_bwd": (ctypes.c_int, [_P(LearnerDesc), c_vp]),
    "prism_learner_clip_adam": (ctypes.c_int, [_P(LearnerDesc), c_vp]),
    "prism_step_front": (ctypes.c_int, [_P(LearnerDesc), _P(ReplayDesc), c_i64, c_vp, c_u64, c_u64, c_f32, c_vp, c_vp,
                                         c_vp]),
    "prism_step_back": (ctypes.c_int, [_P(LearnerDesc), _P(ReplayDesc), c_vp, c_f32, c_f32, c_vp]),
    "prism_act_forward": (ctypes.c_int, [_P(LearnerDesc), c_vp, c_i32, c_i32, c_vp, c_u64, c_u64, c_vp, c_vp, c_vp]),
    "prism_ids_select": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_i32, c_vp, c_vp,
                                        c_vp, c_vp, c_vp]),
    "prism_greedy_select": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "prism_sync_target": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "prism_direct_reduce_scatter": (ctypes.c_int, [_P(DirectDesc), c_i32, c_vp]),
    "prism_direct_all_gather": (ctypes.c_int, [_P(DirectDesc), c_i32, c_vp]),
    "prism_direct_phase": (ctypes.c_int, [_P(DirectDesc), c_i32, c_i32, c_vp]),
    "prism_direct_flags_alloc": (ctypes.c_int, [_P(c_vp), c_vp]),
    "prism_direct_flags_free": (ctypes.c_int, [c_vp]),
    "prism_direct_flags_open": (ctypes.c_int, [c_vp, _P(c_vp)]),
    "prism_direct_flags_close": (ctypes.c_int, [c_vp]),
    "prism_direct_enable_peer": (ctypes.c_int, [c_i32]),
    "prism_direct_flags_read": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "prism_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "prism_profile_collect": (ctypes.c_int, [_P(ctypes.c_double), _P(c_i64)]),
    "prism_profile_kernel_name": (ctypes.c_char_p, [ctypes.c_int]),
}
N_KERNEL_IDS = 16

_lib = None


def lib():
    """Load (once) and return the native library; raises NativeLibraryError if unavailable."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C prism_amd/csrc` (hipcc --offload-arch=gfx950). prism_amd has no CPU fallback.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        if L.prism_abi_version() != 3:
            raise NativeLibraryError("ABI version mismatch between prism_amd and libprism_hip.so")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != PRISM_OK:
        msg = lib().prism_last_error()
        raise PrismError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) pointer of a torch tensor as c_void_p; None -> NULL."""
    if t is None:
        return c_vp(0)
    return c_vp(t.data_ptr())


def current_stream_handle():
    import torch
    return c_vp(torch.cuda.current_stream().cuda_stream)


def device_info(device=0):
    cu = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(64)
    check(lib().prism_device_info(device, ctypes.byref(cu), buf, 64), "prism_device_info")
    return cu.value, buf.value.decode()
