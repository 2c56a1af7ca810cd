"""ctypes loader for the C-ABI in include/osg_abi.h (libosg_hip.so).

The shared library is the product: hand-written HIP for gfx950.  There is NO
fallback — if the library is missing or no MI355X is visible, loading or
context creation fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSG_VARIANT_LIB: an A/B build of the same library (tools/build_variant.sh) for measurements and their parity runs
LIB_PATH = os.path.abspath(os.environ["OSG_VARIANT_LIB"]) if os.environ.get("OSG_VARIANT_LIB") else os.path.join(_HERE, "libosg_hip.so")


class OsgError(RuntimeError):
    """Raised for any non-zero osg_status (mirrors pyspiel.SpielError)."""


class GameDesc(C.Structure):
    _fields_ = [
        ("game_kind", C.c_int32),
        ("num_players", C.c_int32),
        ("num_distinct_actions", C.c_int32),
        ("max_chance_outcomes", C.c_int32),
        ("max_game_length", C.c_int32),
        ("max_chance_nodes", C.c_int32),
        ("obs_size", C.c_int32),
        ("info_size", C.c_int32),
        ("obs_shape", C.c_int32 * 4),
        ("obs_rank", C.c_int32),
        ("info_shape", C.c_int32 * 4),
        ("info_rank", C.c_int32),
        ("mask_words", C.c_int32),
        ("compact_mask_bytes", C.c_int32),
        ("state_words", C.c_int32),
        ("state_word_bytes", C.c_int32),
        ("min_utility", C.c_double),
        ("max_utility", C.c_double),
        ("canonical", C.c_char * 128),
    ]


class MctsCfg(C.Structure):
    _fields_ = [
        ("uct_c", C.c_double),
        ("max_simulations", C.c_int32),
        ("n_rollouts", C.c_int32),
        ("solve", C.c_int32),
        ("max_nodes", C.c_int32),
        ("seed", C.c_uint64),
        ("index_offset", C.c_int64),
        ("layout", C.c_int32),
        ("child_selection_policy", C.c_int32),
    ]


class CfrCfg(C.Structure):
    _fields_ = [
        ("alternating_updates", C.c_int32),
        ("linear_averaging", C.c_int32),
        ("regret_matching_plus", C.c_int32),
        ("solver", C.c_int32),
        ("epsilon", C.c_double),
        ("kernel", C.c_int32),
        ("replicas", C.c_int32),
        ("random_initial_regrets", C.c_int32),
        ("seed", C.c_uint64),
        ("replica_offset", C.c_int64),
    ]


VP = C.c_void_p
I64 = C.c_int64
U64 = C.c_uint64
INT = C.c_int

# name -> (restype, argtypes).  Every symbol include/osg_abi.h declares.
SIGNATURES = {
    "osg_last_error": (C.c_char_p, []),
    "osg_ctx_create": (INT, [INT, VP, INT, C.POINTER(VP)]),
    "osg_ctx_destroy": (INT, [VP]),
    "osg_ctx_synchronize": (INT, [VP]),
    "osg_ctx_stream": (VP, [VP]),
    "osg_ctx_trim": (INT, [VP]),
    "osg_ctx_set_stream": (INT, [VP, VP]),
    "osg_game_describe": (INT, [C.c_char_p, C.POINTER(GameDesc)]),
    "osg_batch_create": (INT, [VP, C.c_char_p, I64, C.POINTER(VP)]),
    "osg_batch_destroy": (INT, [VP]),
    "osg_batch_size": (I64, [VP]),
    "osg_batch_describe": (INT, [VP, C.POINTER(GameDesc)]),
    "osg_batch_reset": (INT, [VP]),
    "osg_batch_copy": (INT, [VP, VP]),
    "osg_batch_gather": (INT, [VP, VP, VP, INT]),
    "osg_batch_download": (INT, [VP, VP]),
    "osg_batch_upload": (INT, [VP, VP]),
    "osg_batch_set_cells": (INT, [VP, C.c_int64, C.c_char_p, INT]),
    "osg_batch_device_ptr": (VP, [VP]),
    "osg_legal_mask": (INT, [VP, VP, INT]),
    "osg_apply": (INT, [VP, VP, INT, C.POINTER(I64)]),
    "osg_status_query": (INT, [VP, VP, VP, VP, INT]),
    "osg_chance_probs": (INT, [VP, VP, INT]),
    "osg_step": (INT, [VP, VP, VP, VP, VP]),
    "osg_observation": (INT, [VP, INT, INT, VP, INT]),
    "osg_information_state_string": (INT, [VP, I64, INT, C.c_char_p, INT]),
    "osg_observation_string": (INT, [VP, I64, INT, C.c_char_p, INT]),
    "osg_state_string": (INT, [VP, I64, C.c_char_p, INT]),
    "osg_action_string": (INT, [VP, I64, INT, C.c_int32, C.c_char_p, INT]),
    "osg_copy_bytes": (INT, [VP, VP, VP, I64]),
    "osg_env_step": (INT, [VP, VP, VP, U64, I64, I64, VP, VP, VP, VP]),
    "osg_env_step_compact": (INT, [VP, VP, VP, U64, I64, I64, VP, VP]),
    "osg_random_steps": (INT, [VP, U64, I64, INT, VP]),
    "osg_synth_batch": (INT, [VP, U64, I64, INT, VP, VP]),
    "osg_rollout": (INT, [VP, U64, I64, INT, VP, VP, INT]),
    "osg_mcts_search": (INT, [VP, C.POINTER(MctsCfg), VP, VP, VP, VP, VP, INT]),
    "osg_mcts_tree_create": (INT, [VP, C.POINTER(MctsCfg), INT, C.POINTER(VP)]),
    "osg_mcts_tree_destroy": (INT, [VP]),
    "osg_mcts_tree_advance": (INT, [VP, VP, VP, VP, VP, INT, C.POINTER(I64)]),
    "osg_mcts_tree_advance_host": (INT, [VP, VP, VP, VP, INT, VP, INT, C.POINTER(I64)]),
    "osg_mcts_tree_rollout_values": (INT, [VP, VP, VP]),
    "osg_mcts_tree_results": (INT, [VP, VP, VP, VP, VP, VP, VP]),
    "osg_mcts_tree_leaf_path": (INT, [VP, I64, VP, INT]),
    "osg_mcts_tree_nodes": (I64, [VP, I64]),
    "osg_mcts_tree_download": (INT, [VP, I64, I64, VP, VP, VP, VP, VP]),
    "osg_cfr_create": (INT, [VP, C.c_char_p, C.POINTER(CfrCfg), C.POINTER(VP)]),
    "osg_cfr_destroy": (INT, [VP]),
    "osg_cfr_sizes": (INT, [VP, C.POINTER(I64)]),
    "osg_cfr_iterate": (INT, [VP, INT]),
    "osg_cfr_reset": (INT, [VP]),
    "osg_cfr_iteration": (INT, [VP]),
    "osg_cfr_last_kernel": (C.c_char_p, [VP]),
    "osg_cfr_last_eval_kernel": (C.c_char_p, [VP]),
    "osg_cfr_set_iteration": (INT, [VP, INT]),
    "osg_cfr_replicas": (INT, [VP]),
    "osg_cfr_select_replica": (INT, [VP, INT]),
    "osg_mccfr_sample": (INT, [VP, U64, I64, I64]),
    "osg_cfr_upload_tables": (INT, [VP, VP, VP, VP]),
    "osg_mccfr_set_average_type": (INT, [VP, INT]),
    "osg_mccfr_sample_uniforms": (INT, [VP, INT, VP, INT, C.POINTER(C.c_int32)]),
    "osg_mccfr_full_average": (INT, [VP, C.c_double]),
    "osg_cfr_br_iterate": (INT, [VP, INT]),
    "osg_mccfr_iterate": (INT, [VP, U64, I64, I64]),
    "osg_cfr_table_ptrs": (INT, [VP, C.POINTER(VP), C.POINTER(VP), C.POINTER(VP)]),
    "osg_mccfr_delta_ptrs": (INT, [VP, C.POINTER(VP), C.POINTER(VP)]),
    "osg_mccfr_apply_deltas": (INT, [VP]),
    "osg_mccfr_sample_into": (INT, [VP, U64, I64, I64, VP]),
    "osg_mccfr_apply_deltas_from": (INT, [VP, VP]),
    "osg_mccfr_spare_delta_buffer": (INT, [VP, INT, C.POINTER(VP)]),
    "osg_cfr_tables": (INT, [VP, VP, VP, VP, VP, VP, VP]),
    "osg_cfr_evaluate_policy": (INT, [VP, INT, VP, VP, VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "osg_cfr_infostate_player": (INT, [VP, I64]),
    "osg_cfr_best_response": (INT, [VP, INT, VP, VP, VP]),
    "osg_cfr_infostate_key": (INT, [VP, I64, C.c_char_p, INT]),
    "osg_cfr_best_response_history_values": (INT, [VP, INT, VP, INT, VP]),
    "osg_cfr_tree_edges": (INT, [VP, VP, VP]),
    "osg_comm_unique_id": (INT, [VP]),
    "osg_comm_create": (INT, [VP, INT, INT, VP, C.POINTER(VP)]),
    "osg_comm_destroy": (INT, [VP]),
    "osg_comm_oneshot_create": (INT, [VP, INT, INT, I64, C.POINTER(VP)]),
    "osg_comm_oneshot_handle": (INT, [VP, VP]),
    "osg_comm_oneshot_connect": (INT, [VP, VP]),
    "osg_comm_check": (INT, [VP]),
    "osg_comm_rank": (INT, [VP]),
    "osg_comm_world": (INT, [VP]),
    "osg_allreduce_sum_f64": (INT, [VP, VP, I64]),
    "osg_allreduce_sum_i32": (INT, [VP, VP, I64]),
    "osg_allreduce_sum_f64_begin": (INT, [VP, VP, I64]),
    "osg_allreduce_end": (INT, [VP]),
}

_lib = None


def lib():
    """Load libosg_hip.so (once).  Fails loudly when the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OsgError(
                f"{LIB_PATH} is missing: the MI355X engine is hand-written HIP and has no "
                "CPU fallback.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C open_spiel_amd/csrc`.")
        from . import _load_torch_runtime_first
        _load_torch_runtime_first()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise OsgError(f"osg error {rc}: {lib().osg_last_error().decode()}")


def describe(game_string):
    d = GameDesc()
    check(lib().osg_game_describe(game_string.encode(), C.byref(d)))
    return d
