"""ctypes binding of libazmi.so (include/azmi.h).

The library is the product: if it is missing this module raises ImportError —
there is no Python/CPU fallback for any compute entry point.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZMI_LIB") or os.path.join(os.path.dirname(_HERE), "libazmi.so")      # AZMI_LIB: another build of the same ABI (A/B timing of a kernel change on ONE box)

AZMI_MAX_PLAYERS = 4
AZMI_MAX_PERMS = 8
AZMI_MAX_GROUPS = 4


class PlayParamsC(C.Structure):
    _fields_ = [
        ("games_to_play", C.c_uint32),
        ("concurrent_games", C.c_uint32),
        ("max_batch_size", C.c_uint32),
        ("max_cache_size", C.c_uint32),
        ("cache_shards", C.c_uint32),
        ("num_mcts_visits", C.c_uint32),
        ("mcts_visits", C.c_uint32 * AZMI_MAX_PLAYERS),
        ("cpuct", C.c_float),
        ("start_temp", C.c_float),
        ("final_temp", C.c_float),
        ("temp_decay_half_life", C.c_float),
        ("history_enabled", C.c_int32),
        ("self_play", C.c_int32),
        ("tree_reuse", C.c_int32),
        ("epsilon", C.c_float),
        ("mcts_root_temp", C.c_float),
        ("playout_cap_randomization", C.c_int32),
        ("playout_cap_depth", C.c_uint32),
        ("playout_cap_percent", C.c_float),
        ("fpu_reduction", C.c_float),
        ("root_fpu_zero", C.c_int32),
        ("shaped_dirichlet", C.c_int32),
        ("policy_target_pruning", C.c_int32),
        ("resign_percent", C.c_float),
        ("resign_playthrough_percent", C.c_float),
        ("num_eval_type", C.c_uint32),
        ("eval_type", C.c_int32 * AZMI_MAX_PLAYERS),
        ("gumbel_enabled", C.c_int32),
        ("gumbel_m", C.c_uint32),
        ("gumbel_c_visit", C.c_float),
        ("gumbel_c_scale", C.c_float),
        ("gumbel_full", C.c_int32),
        ("fast_search_uses_gumbel", C.c_int32),
        ("num_model_groups_given", C.c_uint32),
        ("model_groups", C.c_uint8 * AZMI_MAX_PLAYERS),
        ("num_seat_perms", C.c_uint32),
        ("seat_perms", (C.c_uint8 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("has_seat_visits", C.c_int32),
        ("has_seat_cap_visits", C.c_int32),
        ("has_seat_epsilon", C.c_int32),
        ("has_seat_mcts_root_temp", C.c_int32),
        ("has_seat_root_fpu_zero", C.c_int32),
        ("seat_visits", (C.c_uint32 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_cap_visits", (C.c_uint32 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_epsilon", (C.c_float * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_mcts_root_temp", (C.c_float * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_root_fpu_zero", (C.c_uint8 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("has_seat_gumbel_enabled", C.c_int32),
        ("has_seat_gumbel_m", C.c_int32),
        ("has_seat_gumbel_c_visit", C.c_int32),
        ("has_seat_gumbel_c_scale", C.c_int32),
        ("has_seat_gumbel_full", C.c_int32),
        ("has_seat_gumbel_use_improved_policy", C.c_int32),
        ("has_seat_resign_threshold", C.c_int32),
        ("has_seat_resign_consecutive", C.c_int32),
        ("seat_gumbel_enabled", (C.c_uint8 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_gumbel_full", (C.c_uint8 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_gumbel_use_improved_policy", (C.c_uint8 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_gumbel_m", (C.c_uint32 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_resign_consecutive", (C.c_uint32 * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_gumbel_c_visit", (C.c_float * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_gumbel_c_scale", (C.c_float * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("seat_resign_threshold", (C.c_float * AZMI_MAX_PLAYERS) * AZMI_MAX_PERMS),
        ("num_temp_decay_half_life_by_variant", C.c_uint32),
        ("temp_decay_half_life_by_variant", C.c_float * 4),
    ]


class MctsConfigC(C.Structure):
    _fields_ = [("cpuct", C.c_float), ("num_players", C.c_uint32), ("num_moves", C.c_uint32), ("epsilon", C.c_float),
                ("root_policy_temp", C.c_float), ("fpu_reduction", C.c_float), ("relative_values", C.c_int32),
                ("root_fpu_zero", C.c_int32), ("shaped_dirichlet", C.c_int32), ("gumbel_enabled", C.c_int32),
                ("gumbel_m", C.c_uint32), ("gumbel_c_visit", C.c_float), ("gumbel_c_scale", C.c_float),
                ("gumbel_full", C.c_int32), ("max_simulations", C.c_uint32)]


class EngineOptsC(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("device", C.c_int32),
        ("max_inline", C.c_uint32),
        ("history_capacity", C.c_uint32),
        ("log_moves", C.c_int32),
        ("move_log_capacity", C.c_uint32),
        ("sg_pinned_variant", C.c_int32),
        ("sg_variant_probs", C.c_float * 4),
    ]


# every symbol include/azmi.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_PP = C.POINTER
SYMBOLS = {
    "azmi_play_params_default": (None, [_PP(PlayParamsC)]),
    "azmi_engine_opts_default": (None, [_PP(EngineOptsC)]),
    "azmi_last_error": (C.c_char_p, []),
    "azmi_abi_version": (C.c_int, []),
    "azmi_device_count": (C.c_int, []),
    "azmi_game_info": (C.c_int, [C.c_int, _PP(C.c_uint32), _PP(C.c_uint32), _PP(C.c_uint32)]),
    "azmi_pm_create": (C.c_int, [C.c_int, _PP(PlayParamsC), _PP(EngineOptsC), _PP(_VP)]),
    "azmi_pm_create_with_caches": (C.c_int, [C.c_int, _PP(PlayParamsC), _PP(EngineOptsC), _PP(_VP), C.c_uint32, _PP(_VP)]),
    "azmi_pm_stop": (C.c_int, [_VP]),
    "azmi_pm_stopped": (C.c_int, [_VP, _PP(C.c_int)]),
    "azmi_pm_queue_counts": (C.c_int, [_VP, _PP(C.c_uint32), _PP(C.c_uint32)]),
    "azmi_pm_slot_state": (C.c_int, [_VP, C.c_uint32, _VP, C.c_uint32, _PP(C.c_uint32)]),
    "azmi_pm_slot_history": (C.c_int, [_VP, C.c_uint32, _VP, C.c_uint32, _PP(C.c_uint32)]),
    "azmi_pm_cache_stats": (C.c_int, [_VP, _VP]),
    "azmi_pm_stat_sums": (C.c_int, [_VP, _VP]),
    "azmi_pm_slot_canonical": (C.c_int, [_VP, C.c_uint32, _VP]),
    "azmi_pm_destroy": (None, [_VP]),
    "azmi_pm_round": (C.c_int, [_VP, _VP]),
    "azmi_pm_io_buffers": (C.c_int, [_VP, _PP(_VP), _PP(_VP), _PP(_VP)]),
    "azmi_pm_play": (C.c_int, [_VP, _VP]),
    "azmi_pm_poll": (C.c_int, [_VP, _VP, _PP(C.c_uint32), _PP(C.c_uint32)]),
    "azmi_pm_scores": (C.c_int, [_VP, _VP]),
    "azmi_pm_resign_scores": (C.c_int, [_VP, _VP]),
    "azmi_pm_stats": (C.c_int, [_VP, _VP]),
    "azmi_pm_counters": (C.c_int, [_VP, _VP]),
    "azmi_pm_pop_history": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint32, _PP(C.c_uint32)]),
    "azmi_pm_history_device": (C.c_int, [_VP, _PP(_VP), _PP(_VP), _PP(_VP), _PP(_VP), _PP(C.c_uint32)]),
    "azmi_pm_history_window": (C.c_int, [_VP, _PP(C.c_uint32), _PP(C.c_uint32), _PP(C.c_uint32)]),
    "azmi_pm_history_consume": (C.c_int, [_VP, C.c_uint32]),
    "azmi_pm_move_log": (C.c_int, [_VP, _VP, _VP, C.c_uint32, _PP(C.c_uint32)]),
    "azmi_pm_slot_games": (C.c_int, [_VP, _VP]),
    "azmi_pm_build_batch": (C.c_int, [_VP, _VP, C.c_uint32, _VP, _PP(C.c_uint32)]),
    "azmi_pm_update_inferences": (C.c_int, [_VP, _VP, C.c_uint32, _VP, _VP]),
    "azmi_net_blob_bytes": (C.c_size_t, None),
    "azmi_net_create": (C.c_int, None),
    "azmi_net_destroy": (None, None),
    "azmi_net_forward": (C.c_int, None),
    "azmi_net_last_error": (C.c_char_p, None),
    "azmi_cache_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _PP(_VP)]),
    "azmi_cache_destroy": (None, [_VP]),
    "azmi_cache_insert_many": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint32]),
    "azmi_cache_find_many": (C.c_int, [_VP, _VP, C.c_uint32, _VP, _VP, _VP]),
    "azmi_cache_stats": (C.c_int, [_VP, _VP]),
    "azmi_cache_last_error": (C.c_char_p, []),
    "azmi_run_rounds": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "azmi_run_rounds_groups": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, _VP]),
    "azmi_run_pipeline": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint64, _VP, _VP]),
    "azmi_pipeline_supported": (C.c_int, [_VP, _VP]),
    "azmi_pipeline_supported_groups": (C.c_int, [_VP, _VP, C.c_uint32]),
    "azmi_run_pipeline_groups": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_uint64, _VP, _VP]),
    "azmi_debug_pipe_log_dupes": (C.c_int, [_VP, _VP]),
    "azmi_debug_pipe_net_bench": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _VP]),
    "azmi_debug_pipe_net_answers": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, _VP, _VP]),
    "azmi_comm_available": (C.c_int, []),
    "azmi_comm_unique_id": (C.c_int, [_VP]),
    "azmi_comm_create": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, _PP(_VP)]),
    "azmi_comm_destroy": (None, [_VP]),
    "azmi_gather_counts": (C.c_int, [_VP, C.c_uint64, _VP, _VP]),
    "azmi_gather_rows": (C.c_int, [_VP, _VP, _VP, C.c_uint32, _VP, _VP, _VP]),
    "azmi_rng_probe": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_float, C.c_uint32, C.c_uint32, _VP]),
    "azmi_num_symmetries": (C.c_uint32, [C.c_int]),
    "azmi_symmetries": (C.c_int, [C.c_int, C.c_int, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, _VP]),
    "azmi_tafl_symmetries": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int, _VP]),
    "azmi_symmetries_last_error": (C.c_char_p, []),
    "azmi_pm_net_forward": (C.c_int, [_VP, _VP, _VP]),
    "azmi_pm_net_forward_group": (C.c_int, [_VP, C.c_uint32, _VP, _VP]),
    "azmi_pm_groups": (C.c_int, [_VP, _PP(C.c_uint32), _PP(C.c_uint32)]),
    "azmi_pm_perm_scores": (C.c_int, [_VP, C.c_uint32, _VP, _PP(C.c_uint32)]),
    "azmi_pm_build_batch_group": (C.c_int, [_VP, C.c_uint32, _VP, C.c_uint32, _VP, _PP(C.c_uint32)]),
    "azmi_net_forward_rows": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_uint32, _VP]),
    "azmi_net_eval_host": (None, [_VP, C.c_uint32, _VP, _VP, _VP]),
    "azmi_pm_round_net": (C.c_int, [_VP, _VP, _VP, C.c_uint32]),
    "azmi_mcts_create": (C.c_int, [C.c_int, _VP, C.c_uint64, C.c_int, _PP(C.c_void_p)]),
    "azmi_mcts_destroy": (None, [_VP]),
    "azmi_mcts_find_leaf": (C.c_int, [_VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP, C.c_uint32, _PP(C.c_uint32)]),
    "azmi_mcts_process_result": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP]),
    "azmi_mcts_find_leaf_batched": (C.c_int, [_VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP, C.c_uint32, _PP(C.c_uint32)]),
    "azmi_mcts_process_result_batched": (C.c_int, [_VP, C.c_uint32, _VP, _VP, C.c_int, _VP]),
    "azmi_mcts_in_flight_count": (C.c_int, [_VP, _PP(C.c_uint32)]),
    "azmi_mcts_reset_batch": (C.c_int, [_VP]),
    "azmi_mcts_update_root": (C.c_int, [_VP, _VP, C.c_uint32, _VP, C.c_uint32, C.c_uint32]),
    "azmi_mcts_query": (C.c_int, [_VP, C.c_uint32, C.c_float, C.c_uint32, _VP, _VP, _VP]),
    "azmi_game_replay_ex": (C.c_int, [C.c_int, C.c_int, _VP, C.c_uint32, _VP, C.c_uint32, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_uint32]),
    "azmi_game_replay_from": (C.c_int, [C.c_int, C.c_int, _VP, C.c_uint32, _VP, C.c_uint32, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "azmi_playout_eval": (C.c_int, [C.c_int, C.c_int, _VP, C.c_uint32, _VP, C.c_uint32, C.c_uint32, _VP, _VP, _VP]),
    "azmi_pm_num_variants": (C.c_uint32, [_VP]),
    "azmi_pm_variant_sums": (C.c_int, [_VP, C.c_uint32, _VP, _VP, _VP]),
    "azmi_sg_image": (C.c_int, [C.c_int, _VP, C.c_uint32, _VP, C.c_uint32, C.c_uint32, _VP, C.c_uint32, _VP, _VP, C.c_uint32]),
    "azmi_game_replay": (C.c_int, [C.c_int, C.c_int, _VP, C.c_uint32, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
}


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 and looks them up by file
    name, so if libazmi.so pulled in /opt/rocm's copies first, a later `import torch` would load a second runtime whose
    device enumeration fails ("No HIP GPUs are available").  Loading torch's copies first (without importing torch)
    makes libazmi.so bind to them by SONAME, whichever of the two is used first."""
    import importlib.util
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return          # keep the system runtime; torch must then be imported before this package


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        if args is not None:
            fn.argtypes = args
    return lib


lib = load()


def check(rc):
    if rc != 0:
        msg = lib.azmi_last_error().decode("utf-8", "replace")
        if rc == -6:   # AZMI_ERR_RANGE: the reference's std::out_of_range, which pybind11 maps to IndexError
            raise IndexError(msg or "index out of range")
        raise RuntimeError(msg or f"azmi error {rc}")
