"""ctypes binding of libmuzero_amd.so (include/muzero_amd.h).

There is no CPU fallback: importing this module without the built HIP library raises, and every
wrapper raises on a non-zero return code with the library's error text.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch has loaded; loading it before torch
#                              leaves two runtimes in the process and every launch fails with "no ROCm-capable device"

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MUZERO_AMD_LIB") or os.path.join(HERE, "lib", "libmuzero_amd.so")  # override: experiment builds

MZ_F32, MZ_BF16, MZ_F16 = 0, 1, 2
MZ_RESTOWER_X3_PACKED_BYTES = (12 * 36 + 4) * 4096 * 2
MZ_TOWER_X3_ENDS_BYTES = (2 * 5 * 128 * 32 + 2 * 4 * 80 * 32) * 2   # include/muzero_amd.h
ST_OK, ST_ILLEGAL, ST_DONE = 0, 1, 2
VIS_I32, VIS_U16, VIS_U8 = 0, 1, 2


class MzError(RuntimeError):
    pass


class SearchCfg(C.Structure):
    """mz_search_cfg -- the config.py knobs the search reads (config.py:15,19,22,23,29)."""
    _fields_ = [("pb_c_base", C.c_double), ("pb_c_init", C.c_double), ("discount", C.c_double),
                ("noise_frac", C.c_double), ("bound_lo", C.c_double), ("bound_hi", C.c_double),
                ("bounds_none", C.c_int32), ("tie_lowest", C.c_int32), ("negamax", C.c_int32), ("device_noise", C.c_int32),
                ("noise_alpha", C.c_double), ("noise_seed", C.c_uint64)]


class PoolLayout(C.Structure):
    """mz_pool_layout"""
    _fields_ = [("total_bytes", C.c_int64), ("off_stats", C.c_int64), ("off_hidden", C.c_int64),
                ("off_meta", C.c_int64), ("off_path", C.c_int64), ("off_pbc", C.c_int64),
                ("n_trees", C.c_int32), ("max_nodes", C.c_int32), ("hidden_dtype", C.c_int32),
                ("hidden_elem_bytes", C.c_int32)]


class Replay(C.Structure):
    """mz_replay -- device arrays of the replay window (host struct of device pointers)"""
    _fields_ = [("white", C.c_void_p), ("black", C.c_void_p), ("actions", C.c_void_p), ("rewards", C.c_void_p),
                ("policy", C.c_void_p), ("root_values", C.c_void_p), ("length", C.c_void_p),
                ("window", C.c_int32), ("max_moves", C.c_int32), ("refused", C.c_void_p)]


class LeafRoute(C.Structure):
    """mz_leaf_route -- where a routed descent names the pool slots of its leaf"""
    _fields_ = [("src_slot", C.c_void_p), ("dst_slot", C.c_void_p), ("act_planes", C.c_void_p)]


class TowerWeights(C.Structure):
    """mz_tower_weights -- device pointers to the prepacked tower weights"""
    _fields_ = [("w_first", C.c_void_p), ("b_first", C.c_void_p), ("w_mid", C.c_void_p), ("b_mid", C.c_void_p),
                ("w_last", C.c_void_p), ("b_last", C.c_void_p), ("cin", C.c_int32), ("debug_stamps", C.c_void_p)]


class HeadsWeights(C.Structure):
    """mz_heads_weights -- device pointers of the prediction heads (f32; p1_w / p2_w bf16 [out][in])"""
    _fields_ = [(k, C.c_void_p) for k in ("pc_w", "pc_b", "p1_w", "p1_b", "p2_w", "p2_b", "vc_w", "vc_b", "v1_w", "v1_b")]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "muzero_amd: %s is missing. Build it with `make -C muzero_amd/csrc` (or "
        "`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_p, _i32, _i64, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
_LP, _CP, _RP = C.POINTER(PoolLayout), C.POINTER(SearchCfg), C.POINTER(Replay)

# name -> argtypes; every symbol include/muzero_amd.h declares (tests check the export list)
SIGNATURES = {
    "mz_env_reset": [_p] * 6 + [_i64, _p],
    "mz_env_legal_mask": [_p] * 4 + [_i64, _p],
    "mz_env_step": [_p] * 9 + [_i64, _p],
    "mz_env_step_packed": [_p, _p, _p, _p, _i64, _p],
    "mz_env_random_playout": [_u64, _u64, _i64, _i32, _i32] + [_p] * 9 + [_p],
    "mz_env_random_action": [_u64, _u64, _i32, _p, _p, _i64, _p],
    "mz_action_encode": [_p, _p, _p, _i32, _i64, _p],
    "mz_make_image": [_p, _p, _i64, _p, _i32, _p, _i32, _i64, _p],
    "mz_controller_image": [_p, _p, _i64, _p, _i32, _p, _i32, _i64, _p],
    "mz_softmax_sample": [_p, C.c_double, _p, _p, _i64, _p],
    "mz_pool_layout_compute": [_i32, _i32, _i32, _LP],
    "mz_pool_init": [_p, _LP, _CP, _p],
    "mz_mcts_root_init": [_p, _LP, _CP, _p, _p, _p, _p, _p, _p, _i32, _p],
    "mz_mcts_select": [_p, _LP, _CP, _p, _i32, _p],
    "mz_mcts_expand_backup": [_p, _LP, _CP, _p, _p, _p, _i32, _p],
    "mz_mcts_backup_select": [_p, _LP, _CP, _p, _p, _p, _p, _i32, _p],
    "mz_mcts_expand": [_p, _LP, _p, _p, _i32, _p],
    "mz_mcts_backup": [_p, _LP, _CP, _p, _i32, _p],
    "mz_mcts_node_scores": [_p, _LP, _CP, _p, _p, _p, _p, _i32, _p],
    "mz_mcts_root_stats": [_p, _LP, _p, _p, _p, _p, _i32, _p],
    "mz_select_action": [_p, _p, _i32, _p, _p, _i64, _p],
    "mz_bias_act_nhwc": [_p, _p, _p, _i64, _i32, _i32, _i32, _p],
    "mz_tower_forward": [C.POINTER(TowerWeights), _p, _p, _i32, _i64, _p],
    "mz_inference_forward": [C.POINTER(TowerWeights), C.POINTER(HeadsWeights), _p, _p, _i32, _p, _p, _i64, _p],
    "mz_inference_forward_routed": [C.POINTER(TowerWeights), C.POINTER(HeadsWeights), _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _i64, _p],
    "mz_mcts_select_routed": [_p, _LP, _CP, C.POINTER(LeafRoute), _i32, _p],
    "mz_mcts_backup_select_routed": [_p, _LP, _CP, _p, _p, C.POINTER(LeafRoute), _i32, _p],
    "mz_search": [C.POINTER(TowerWeights), C.POINTER(TowerWeights), C.POINTER(HeadsWeights), _CP, _p, _p, _p, _p, _i32, _i32, _p, _i64,
                  _p, _p, _p],
    "mz_search_noalloc": [C.POINTER(TowerWeights), C.POINTER(TowerWeights), C.POINTER(HeadsWeights), _CP, _p, _p, _p, _p, _i32, _i32,
                          _p, _p, _p],
    "mz_replay_save": [_RP, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i64, _i32, _p],
    "mz_replay_sample_batch": [_RP, _p, _p, _i32, _i32, _p, _p, _p, _p, _p],
    "mz_conv3x3_wgrad": [_p, _p, _p, _i32, _p, _p, _i32, _p, _i64, _p],
    "mz_conv3x3_pack": [_p, _p, _p, _p],
    "mz_conv3x3_forward": [_p, _p, _p, _p, _i64, _i32, _p],
    "mz_conv3x3_fused": [_p, _p, _p, _i32, _p, _p, _p, _i64, _i32, _p],
    "mz_conv3x3_pack_part": [_p, _p, _p, _i32, _p],
    "mz_conv3x3_pack_many": [_p, _p, _p, _i32, _i32, _p],
    "mz_conv3x3_wgrad_x3": [_p, _p, _p, _i32, _p, _p, _i32, _p, _i64, _p],
    "mz_restower_x3_pack": [_p, _p, _p],
    "mz_restower_x3_forward": [_p, _p, _p, _p, _i64, _i32, _p],
    "mz_restower_x3_stamps": [_p, _p, _p, _p, _i64, _p, _p],
    "mz_tower_x3_pack_ends": [_p, _i32, _p, _p, _p],
    "mz_tower_x3_forward": [_p, _p, _p, _p, _p, _i32, _p, _p, _i64, _p],
    "mz_inference_x3_forward": [_p, _p, _p, _p, _p, _i32, C.POINTER(HeadsWeights), _p, _p, _p, _p, _i64, _p],
}
for _name, _args in SIGNATURES.items():
    _f = getattr(lib, _name)
    _f.argtypes = _args
    _f.restype = C.c_int
lib.mz_search_workspace_bytes.restype = C.c_int64
lib.mz_search_workspace_bytes.argtypes = [_i32, _i32]
lib.mz_conv3x3_wgrad_workspace.restype = C.c_int64
lib.mz_conv3x3_wgrad_workspace.argtypes = [_i64]
lib.mz_last_error.restype = C.c_char_p
lib.mz_last_error.argtypes = []
lib.mz_version.restype = C.c_int
lib.mz_version.argtypes = []
EXPORTS = sorted(list(SIGNATURES) + ["mz_last_error", "mz_version", "mz_search_workspace_bytes", "mz_conv3x3_wgrad_workspace"])


def call(name, *args):
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MzError("%s failed (%d): %s" % (name, rc, lib.mz_last_error().decode()))


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """device pointer of a torch tensor (None -> NULL); refuses host tensors loudly."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MzError("muzero_amd kernels need device tensors; got a %s tensor (no CPU fallback)" % t.device)
    if not t.is_contiguous():
        raise MzError("muzero_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())
