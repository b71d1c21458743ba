"""Writes tests/REFERENCE_TEST_COVERAGE.md: every TEST of the reference's C++ test files for the hot path
(/root/reference/src/{mcts,play_manager,star_gambit_gs,star_gambit_unified_gs,connect4_gs,opentafl_gs,brandubh_gs,tawlbwrdd_gs,
s3fifo_cache,tafl_helper}_test.cc and the in-boundary cases of test_history.py / test_star_gambit_unified.py - read here as text, never shipped) mapped
to the test of this repository that mirrors it, or to the reason it is not mirrored.  Run in the build container:
    python tests/make_reference_coverage.py
The mapping below is the maintained part; the script fails when a reference TEST has no entry (a new reference test)."""
import os
import re
import sys

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))

SG = "tests/stargambit_cases.py::"          # one script, run on the oracle (tests/test_oracle_stargambit.py, CPU) and on the device objects (tests/test_gpu_stargambit.py)
T0 = "tests/test_gpu_stargambit.py::test_rules_random_playouts_every_prefix (legal moves / scores / player / turn after every action, planes every fifth, oracle vs device)"
NB = ("not mirrored as such: a free helper function / class constant / string dump without a binding in py_wrapper.cc, i.e. not part of the "
      "boundary; ")

MAP = {
    # ---- mcts_test.cc ---------------------------------------------------------------------------------------------------------
    "mcts_test.cc": {
        "Node.Basic": "tests/test_oracle_pinned.py::test_node_uct_known_answers / test_node_best_child_known_answer (the uct values and the best child as data, tests/golden/known_answers.json)",
        "MCTS.Basic": "tests/test_oracle_pinned.py::test_mcts_known_answers (the visit profile [5, 5, 1, 1], the picked move); on the device tests/test_gpu_mcts_object.py::test_reference_known_answer",
        "PlayoutEval.Basic": "tests/test_gpu_parity_connect4.py (playout_eval: outcome statistics of random Connect4 games 55.6 / 44.2 / 0.25 %) + tests/test_gpu_gamestate.py playout cases",
        "MCTS.PlayoutEval": "tests/test_gpu_groups_perms.py / test_gpu_tafl_family.py PLAYOUT seats (bit-exact vs the oracle's rollout stream); the reference case only asserts a legal move",
        "MCTS.RootValueSetOnFirstEval": "tests/test_gpu_mcts_object.py::test_reference_mcts_property_cases",
        "MCTS.PuctInversionWithNoise": "tests/test_gpu_mcts_object.py::test_call_by_call_parity_with_oracle (noise configurations: priors after add_root_noise equal the oracle's, incl. the inversion)",
        "MCTS.RootFpuZero": "tests/test_gpu_mcts_object.py::test_reference_mcts_property_cases",
        "MCTS.PolicyTargetPruning": "tests/test_gpu_mcts_object.py::test_reference_mcts_property_cases",
        "MCTS.ShapedDirichletDistribution": "tests/test_gpu_mcts_object.py::test_reference_shaped_dirichlet_distribution_cases (round 6: the reference's assertions on the device object, shaped and uniform noise) + draw by draw: tests/test_gpu_rng.py gamma stream vs real libstdc++",
        "MCTS.ShapedDirichletAlphaDistribution": "tests/test_gpu_mcts_object.py::test_reference_shaped_dirichlet_distribution_cases (400 seeds instead of 50 trials: every legal move noised in every trial, plus the noise's mean and variance against the symmetric-Dirichlet values)",
        "MCTS.PuctInversionGradual": "tests/test_gpu_mcts_object.py::test_call_by_call_parity_with_oracle (visit counts after every call equal the oracle's under noise)",
        "MCTS.TrainEvalSeparation": "tests/test_gpu_parity_connect4.py::test_playmanager_option_tiers (self_play on / off, eval_temp paths) - the PlayManager flags the case toggles",
        "MCTS.RawPolicyTemperatureInteraction": "tests/test_gpu_mcts_object.py::test_call_by_call_parity_with_oracle (root temperature configurations)",
        "MCTS.PuctInversionPropertiesAfterNoise": "tests/test_gpu_mcts_object.py::test_call_by_call_parity_with_oracle (noise + temperature)",
        "MCTS.BatchedBasic": "tests/test_gpu_mcts_object.py::test_batched_basic_and_terminal",
        "MCTS.BatchedTerminal": "tests/test_gpu_mcts_object.py::test_batched_basic_and_terminal",
        "MCTS.BatchedSingleEquivalent": "tests/test_gpu_mcts_object.py::test_batched_single_equals_unbatched",
        "MCTS.WUUCTDiversity": "tests/test_gpu_mcts_object.py::test_wu_uct_diversity",
        "GumbelMCTS.*": "tests/test_gpu_mcts_object.py::test_reference_gumbel_mcts_cases (the nine cases, the reference's calls and assertions) + tests/test_gpu_gumbel.py tiers vs the oracle",
    },
    # ---- play_manager_test.cc ---------------------------------------------------------------------------------------------------
    "play_manager_test.cc": {
        "PlayManager.Basic": "tests/test_gpu_reference_playmanager_cases.py::test_basic",
        "PlayManager.MultiThreaded": "tests/test_gpu_reference_playmanager_cases.py::test_seat_visits_overrides_visit_count_with_worker_threads / test_play_with_nn_seats_waits_for_the_batcher (several threads in play(); the engine has no worker threads: the entry points are serialised)",
        "PlayManager.StopEarly": "tests/test_gpu_reference_playmanager_cases.py::test_stop_early",
        "PlayManager.Seat*Throws": "tests/test_abi.py::test_reference_seat_matrix_dimension_errors (CPU: the host-side constructor checks, same messages)",
        "PlayManager.SeatOverridesDefaultFill": "tests/test_gpu_reference_playmanager_cases.py::test_seat_overrides_default_fill",
        "PlayManager.NewSeatFieldDefaultsFromEmpty": "tests/test_gpu_reference_playmanager_cases.py::test_seat_overrides_default_fill",
        "PlayManager.PerSeatOverrides": "tests/test_gpu_reference_playmanager_cases.py::test_per_seat_overrides",
        "PlayManager.InitOrderPerPermMctsSettings": "tests/test_gpu_reference_playmanager_cases.py::test_init_order_per_perm",
        "PlayManager.RunsWithG3OptInAndResign": "tests/test_gpu_reference_playmanager_cases.py::test_runs_with_g3_opt_in_and_resign",
        "PlayManager.AggressiveResignTerminatesGames": "tests/test_gpu_reference_playmanager_cases.py::test_aggressive_resign_terminates_games",
    },
    # ---- star_gambit_gs_test.cc -------------------------------------------------------------------------------------------------
    "star_gambit_gs_test.cc": {
        "HexCoordinates.*": NB + "hex arithmetic is exercised by every move / shot / deploy of " + T0,
        "UnitShapes.PortalHexes": SG + "case_initial_state (portal anchors, facings, hit points per variant)",
        "UnitShapes.*": SG + "case_unit_and_action_space_constants (cells of a freshly deployed unit read off the observation)",
        "GameState.InitialState": SG + "case_initial_state", "GameState.InitialUnitsArePortals": SG + "case_initial_state",
        "GameState.ScoresNotOverInitially": SG + "case_initial_state",
        "GameState.ValidMovesOnTurnOne": SG + "case_turn_one_is_deploy_only", "GameState.DeployAction": SG + "case_deploy_switches_player",
        "GameState.CopyEquality": SG + "case_deploy_switches_player", "GameState.Canonicalized": SG + "case_initial_state (shape) / case_observation_channels (contents)",
        "GameState.DumpOutput": NB + "dump() of the Python object prints the same header lines (alphazero/__init__.py), not pinned",
        "TurnStructure.TurnOneIsDeployOnly": SG + "case_turn_one_is_deploy_only",
        "TurnStructure.AfterTurnOneCanMoveAndFire": SG + "case_notation_game / case_fire_and_damage",
        "Combat.CannonInfo*": SG + "case_unit_and_action_space_constants (cannon slots of a fresh unit: 1 / 3 / 4) + case_fire_and_damage (ranges, damage)",
        "Combat.LineOfSight*": NB + "line of sight decides which fire actions are legal: " + T0,
        "Deployment.DeployHexLocation": SG + "case_deploy_switches_player (deploy hex of player 0's first fighter)",
        "Deployment.ValidDeployFacings": SG + "case_turn_one_is_deploy_only", "Deployment.DreadnoughtDeployAllFacings*": SG + "case_p1_deploy_facings",
        "GameFlow.PlayMultipleTurns": SG + "case_first_valid_game_ends", "GameFlow.Symmetries": "tests/test_oracle_stargambit.py::test_mirror_symmetry_cases_of_the_reference",
        "Symmetries.*": "tests/test_oracle_stargambit.py::test_mirror_symmetry_cases_of_the_reference (count, identity) + tests/test_gpu_stargambit.py::test_symmetries_on_the_device_equal_the_oracle",
        "P1Canonicalization.*": SG + "case_p1_observation",
        "MirrorSymmetry.*": "tests/test_oracle_stargambit.py::test_mirror_symmetry_cases_of_the_reference (self-inverse, observation transpose, spatial / deploy / end-turn policy remap, value, facing channels, deploy facings) + the device kernel vs the oracle in tests/test_gpu_stargambit.py::test_symmetries_on_the_device_equal_the_oracle",
        "UnitProperties.MaxMoves": SG + "case_deploy_switches_player / case_fighter_and_cruiser_move_options (moves left after a deploy, number of moves a unit gets)",
        "UnitProperties.*": SG + "case_unit_and_action_space_constants",
        "ActionSpace.*": SG + "case_unit_and_action_space_constants",
        "MovementDirections.*": SG + "case_fighter_and_cruiser_move_options (3 / 5 move options); dreadnought: " + T0,
        "FireValidation.NoFireWithoutTarget": SG + "case_no_fire_without_target", "FireValidation.FireAvailableWhenTargetInRange": SG + "case_fire_and_damage",
        "MovementConstraints.*": SG + "case_fighter_and_cruiser_move_options", "SlotNumbering.*": SG + "case_slot_numbering",
        "EndToEnd.*": SG + "case_first_valid_game_ends", "MoveParsing.*": NB + "the test file's own notation helper; its notation is re-implemented in tests/stargambit_cases.py::parse_move and used by case_notation_game",
        "FullGame.PlayWithNotation": SG + "case_notation_game", "FullGame.DeployAndMoveSequence": SG + "case_notation_game", "FullGame.CruiserDeployAndMove": SG + "case_notation_game",
        "FullGame.CompleteGameToVictory": SG + "case_first_valid_game_ends", "FullGame.FireWhenInRange": SG + "case_fire_and_damage",
        "CharacterizationCruiserAnchor.*": SG + "case_notation_game (cruiser anchor = front after deploy and move)",
        "CharacterizationCruiserMovement.*": SG + "case_notation_game / case_fighter_and_cruiser_move_options",
        "CharacterizationCruiserCannons.*": SG + "case_fire_and_damage", "CharacterizationActionSpace.*": SG + "case_unified_shapes_and_remap (action index = (row * dim + col) * 10 + slot in both spaces)",
        "CharacterizationObservation.*": SG + "case_observation_channels", "CharacterizationDreadnought.*": SG + "case_unit_and_action_space_constants / case_p1_deploy_facings",
        "CharacterizationFighter.*": SG + "case_unit_and_action_space_constants / case_fighter_and_cruiser_move_options",
        "TerminalStates.ScoresNotPresentDuringGame": SG + "case_initial_state", "TerminalStates.ScoresSumToOne": SG + "case_first_valid_game_ends",
        "TerminalStates.WinnerGetsOne": SG + "case_portal_kill_wins", "TerminalStates.DrawIndexIsTwo": SG + "case_threefold_repetition (a draw scores [0, 0, 1])",
        "TerminalStates.CanonicalizedWorksAfterGameEnd": SG + "case_portal_kill_wins / " + T0 + " (planes at the end of every game)",
        "RepetitionObservation.*": SG + "case_history_cleared_on_deploy / case_repetition_channel_counts",
        "MidTurnRepetition.PositionTrackedAfterEveryAction": SG + "case_threefold_repetition (the history grows per action; tests/test_gpu_stargambit.py::test_game_data_gs_of_a_running_slot reads it)",
        "ThreefoldRepetition.DrawOnThirdOccurrence": SG + "case_threefold_repetition", "ThreefoldRepetition.HistoryClearedOnDeploy": SG + "case_history_cleared_on_deploy",
        "ThreefoldRepetition.CheckOccursAtTurnStart": SG + "case_threefold_repetition (the draw appears when the turn ends, not mid-turn)",
        "ThreefoldRepetition.DifferentPlayersDifferentPositions": SG + "case_threefold_repetition (the position hash includes the player: the script's count would otherwise be reached a turn earlier)",
        "RelativeValues.FlagIsTrue": "tests/test_gpu_gamestate.py / alphazero.StarGambit*GS.relative_values()", "RelativeValues.Player*": "tests/test_oracle_stargambit.py::test_playmanager_relative_targets_and_variant_tables (rotation of the value targets per mover)",
        "RelativeValues.RoundTripIsIdentity": "tests/test_oracle_stargambit.py::test_playmanager_relative_targets_and_variant_tables",
        "RelativeValues.MCTSRelativeValueBackup": "tests/test_oracle_stargambit.py::test_mcts_backs_up_relative_values_as_absolute + tests/test_gpu_stargambit.py::test_mcts_object_on_stargambit",
        "RelativeValues.MCTSTerminalStateNotRotated": "tests/test_gpu_stargambit.py::test_mcts_object_on_stargambit (call-by-call parity incl. terminal leaves)",
        "RelativeValues.SymmetriesPreserveValues": "tests/test_oracle_stargambit.py::test_mirror_symmetry_cases_of_the_reference (value preserved)",
        "RelativeValues.TrainingDataRotation": "tests/test_oracle_stargambit.py::test_playmanager_relative_targets_and_variant_tables + tests/test_gpu_stargambit.py::test_playmanager_exact_tier (history rows byte-equal incl. the relative value targets)",
    },
    # ---- star_gambit_unified_gs_test.cc -----------------------------------------------------------------------------------------
    "star_gambit_unified_gs_test.cc": {
        "StarGambitUnifiedGS.RandomizeStartPicksVariant": "tests/test_oracle_stargambit.py::test_variant_draw_rule_is_the_documented_one (the reference draws from an unseedable mt19937: the draw rule is build-defined) + tests/test_gpu_stargambit.py::test_variant_statistics_match_the_oracle",
        "StarGambitUnifiedGS.PinnedVariant": SG + "case_unified_shapes_and_remap (pinned games of all four variants)",
        "StarGambitUnifiedGS.Copy": SG + "case_deploy_switches_player (copy / ==) on the unified object via tests/test_gpu_gamestate.py",
        "StarGambitUnifiedGS.GameCompletion": SG + "case_first_valid_game_ends",
        "StarGambitUnifiedGS.Symmetry*": "tests/test_oracle_stargambit.py::test_mirror_symmetry_cases_of_the_reference + tests/test_gpu_stargambit.py::test_symmetries_on_the_device_equal_the_oracle",
        "StarGambitUnifiedGS.*": SG + "case_unified_shapes_and_remap (static constants, canonical shape, game-type channels, padding of the 11 x 11 variants, action / deploy remap against the plain games at every step of random games)",
    },
    # ---- connect4_gs_test.cc (round 6: the files below were mirrored in code since rounds 1-3, but not in this ledger) -----------------
    "connect4_gs_test.cc": {
        "Connect4GS.Equals": "tests/test_gpu_gamestate.py::test_reference_connect4_gs_cases (==, != after a move, equal again after the same moves) on the device object",
        "Connect4GS.Copy": "tests/test_gpu_gamestate.py::test_reference_connect4_gs_cases (copy() equal and independent) + test_connect4_from_board_ctor_and_pickle",
        "Connect4GS.ValidMoves": "tests/test_oracle_pinned.py::test_connect4_valid_moves (oracle) + tests/test_gpu_gamestate.py::test_reference_connect4_gs_cases (device)",
        "Connect4GS.PlayMove": "tests/test_oracle_pinned.py::test_connect4_play_move_stacks + tests/test_gpu_gamestate.py::test_reference_connect4_gs_cases",
        "Connect4GS.WinState": "tests/test_oracle_pinned.py::test_connect4_win_states (the reference's boards: horizontal / vertical / both diagonals, draw, running) + tests/test_gpu_gamestate.py::test_reference_connect4_gs_cases",
        "Connect4GS.Canonicalize": "tests/test_oracle_pinned.py::test_connect4_canonical + tests/test_gpu_gamestate.py::test_reference_connect4_gs_cases (planes 0 / 1 stones, 2 / 3 the player to move)",
    },
    # ---- opentafl_gs_test.cc: every TEST is one entry of tests/tafl_cases.py::OPENTAFL_CASES (same pieces, move and asserted cells / moves / score) ----
    "opentafl_gs_test.cc": {
        "OpenTaflGS.RepetitionCount": "tests/tafl_cases.py::OPENTAFL_REPETITION: tests/test_oracle_pinned.py::test_opentafl_rule_known_answers (oracle) + tests/test_gpu_tafl_family.py::test_opentafl_reference_rule_cases_on_device",
        "OpenTaflGS.StartingPosition": "tests/test_oracle_pinned.py::test_opentafl_rule_known_answers (piece counts, king on the throne, attackers to move) + tests/test_gpu_tafl_family.py::test_opentafl_reference_rule_cases_on_device",
        "OpenTaflGS.*": "tests/tafl_cases.py::OPENTAFL_CASES entry of the same name, run by tests/test_oracle_pinned.py::test_opentafl_rule_known_answers (oracle, CPU) and tests/test_gpu_tafl_family.py::test_opentafl_reference_rule_cases_on_device (the device object)",
    },
    "brandubh_gs_test.cc": {
        "BrandubhGS.RepetitionCount": "tests/test_oracle_pinned.py::test_brandubh_known_answers + tests/test_gpu_tafl_family.py::test_brandubh_cases_on_device",
    },
    "tawlbwrdd_gs_test.cc": {
        "TawlbwrddGS.RepetitionCount": "tests/test_oracle_pinned.py::test_tawlbwrdd_threefold_repetition + tests/test_gpu_gamestate.py::test_tawlbwrdd_object_walk_matches_oracle / test_a_reference_image_with_a_repetition_map_loads_and_counts_on",
    },
    # ---- s3fifo_cache_test.cc: tests/s3fifo_cases.py restates the single-threaded cases once; it runs on the oracle and on the device cache ----
    "s3fifo_cache_test.cc": {
        "S3FIFOCache.ConcurrentFindInsert": "not mirrored: a thread-safety test of the reference's mutexes; the device cache has no host threads (probes inside kernels are read-only, "
                                            "batch inserts take the shard's lock: tests/test_gpu_cache.py::test_engine_with_cache_is_transparent / test_wide_game_engine_with_cache_is_transparent)",
        "ShardedS3FIFOCache.ConcurrentInsertMany": "not mirrored: as ConcurrentFindInsert (the pipeline's concurrent inserts are covered by tests/test_gpu_pipeline.py::test_in_epoch_answer_table_is_transparent_and_saves_evaluations and the cache-transparency cases)",
        "S3FIFOCache.CapacityZero": "tests/test_gpu_cache.py::test_playmanager_cache_counters_and_no_cache (max_cache_size = 0: no cache object, nothing stored) + tests/s3fifo_cases.py capacity_one for the smallest real cache",
        "ShardedS3FIFOCache.*": "tests/s3fifo_cases.py::sharded_insert_distributes_and_stats_aggregate via tests/test_oracle_pinned.py::test_reference_s3fifo_cases_on_the_oracle and tests/test_gpu_cache.py::test_reference_s3fifo_cases_on_the_device_cache",
        "S3FIFOCache.*": "tests/s3fifo_cases.py case of the same name (snake case), run by tests/test_oracle_pinned.py::test_reference_s3fifo_cases_on_the_oracle (CPU) and tests/test_gpu_cache.py::test_reference_s3fifo_cases_on_the_device_cache (device); op-by-op vs the oracle: test_device_cache_matches_oracle_op_by_op",
    },
    "tafl_helper_test.cc": {
        "TaflHelper.Mirror": "tests/test_oracle_pinned.py::test_symmetry_mirror_known_answer (the reference's 5 x 5 tables) + tests/test_gpu_symmetries.py::test_reference_spot_tables_5x5 (device kernel)",
        "TaflHelper.Rot90": "tests/test_oracle_pinned.py::test_symmetry_rot90_known_answer + tests/test_gpu_symmetries.py::test_reference_spot_tables_5x5",
        "TaflHelper.EightSym": "tests/test_gpu_symmetries.py::test_reference_spot_tables_5x5 / test_group_properties_and_oracle (eight images, identity first)",
        "TaflHelper.*": "tests/test_oracle_pinned.py::test_symmetry_group_properties (bijection, order 4 / 2, eight distinct permutations, geometric equivariance) + tests/test_gpu_symmetries.py::test_group_properties_and_oracle (n = 7, 11 on the device) + tests/test_gpu_tafl_family.py::test_symmetry_kernel_policy_permutation_equals_the_reference_move_maps",
    },
}

# Python tests of the reference for this path (pytest functions; only the cases inside the boundary)
PY_MAP = {
    "test_history.py": {
        "_range": (43, 160),      # the .ptz cases; the reservoir / chunk cases below line 160 belong to the training loop (out of scope)
        "test_roundtrip_float32": "tests/test_history_io.py::test_store_mode_frame_layout_and_round_trip / test_history_triples_round_trip_in_the_reference_layout",
        "test_roundtrip_integer_values": "tests/test_history_io.py::test_history_triples_round_trip_in_the_reference_layout (values survive bit for bit: the frames are stored, not quantised)",
        "test_file_size_reduction": "not mirrored: a compression-ratio assertion on the reference's zstd level; history_io writes STORE-mode zstd frames (no compressor in this image) that the real decoder accepts: tests/test_history_io.py::test_store_frames_are_accepted_by_the_real_decoder_and_real_frames_are_read",
        "test_ptz_extension": "tests/test_history_io.py::test_history_triples_round_trip_in_the_reference_layout (the triples are written as `IIII-BBBB-{canonical,v,pi}-ROWS.ptz`, the pattern glob_file_triples reads back)",
        "test_empty_tensor_roundtrip": "tests/test_history_io.py::test_store_mode_frame_layout_and_round_trip (zero-row tensors)",
        "test_hist_save_creates_ptz": "tests/test_gpu_selfplay_harness.py (self_play(data_folder=...) writes the triples) + tests/test_history_io.py::test_history_triples_round_trip_in_the_reference_layout",
        "test_mixed_format_loading": "tests/test_history_io.py::test_store_frames_are_accepted_by_the_real_decoder_and_real_frames_are_read (frames of the real compressor are read back)",
        "test_glob_file_triples": "tests/test_history_io.py::test_history_triples_round_trip_in_the_reference_layout (glob of the triples of an iteration)",
        "test_glob_file_triples_multiple": "tests/test_history_io.py::test_history_triples_round_trip_in_the_reference_layout (several batches per iteration)",
    },
    "test_star_gambit_unified.py": {
        "_range": (1, 10 ** 9),
        "TestStaticInterface.*": "tests/stargambit_cases.py::case_unified_shapes_and_remap (NUM_MOVES 1709, CANONICAL_SHAPE 36 x 13 x 13, players, symmetries) on the oracle and the device objects",
        "TestStaticInterface.test_registry_entries": "not mirrored: load_game.py's registry is the reference's config layer (out of scope); the classes it names exist: tests/test_gpu_gamestate.py / alphazero.StarGambit*GS",
        "TestCanonicalObservation.*": "tests/stargambit_cases.py::case_unified_shapes_and_remap (game-type channels one-hot and broadcast, zero outer ring of the 11 x 11 variants, Battle's valid outer hexes)",
        "TestActionRemapping.*": "tests/stargambit_cases.py::case_unified_shapes_and_remap (valid_moves size, no outer-ring moves, deploys at the unified offset, random games playable in both spaces)",
        "TestVariantSelection.test_pinned*": "tests/stargambit_cases.py::case_unified_shapes_and_remap (pinned games of all four variants; the pinned subclasses of py_wrapper.cc:668-705)",
        "TestVariantSelection.*": "tests/test_oracle_stargambit.py::test_variant_draw_rule_is_the_documented_one + tests/test_gpu_stargambit.py::test_variant_statistics_match_the_oracle (uniform and biased probabilities)",
        "TestCopy.*": "tests/stargambit_cases.py::case_deploy_switches_player (copy equal / independent) via tests/test_gpu_gamestate.py",
        "TestConfigIntegration.*": "not mirrored: config.py / load_game.py validation of the training configuration (out of scope, SURVEY 2); the engine-side knob (variant probabilities) is PlayParams / StarGambitUnifiedGS(probs=...): tests/test_gpu_stargambit.py::test_variant_statistics_match_the_oracle",
        "TestComputeUnifiedProbs.*": "not mirrored: compute_unified_probs belongs to the training loop's sample balancing (game_runner.py, out of scope); its OUTPUT is the probs vector the boundary takes",
    },
}


def lookup(table, name):
    if name in table:
        return table[name]
    best = None
    for pat, val in table.items():
        if "*" in pat and re.fullmatch(pat.replace(".", r"\.").replace("*", ".*"), name):
            if best is None or len(pat) > len(best[0]):
                best = (pat, val)
    return best[1] if best else None


def main():
    out = ["# Reference tests of the hot path and where they are mirrored", "",
           "Generated by `tests/make_reference_coverage.py` from the reference's test files (names and line numbers only). "
           "`tests/stargambit_cases.py::case_*` is one script that runs on the CPU oracle (`tests/test_oracle_stargambit.py`) and on the "
           "device objects (`tests/test_gpu_stargambit.py`).", ""]
    missing = []
    for fname, table in MAP.items():
        path = os.path.join(REF, fname)
        tests = []
        for i, line in enumerate(open(path), 1):
            m = re.match(r"TEST\((\w+),\s*(\w+)\)", line)
            if m:
                tests.append((i, f"{m.group(1)}.{m.group(2)}"))
        mirrored = 0
        rows = []
        for ln, name in tests:
            val = lookup(table, name)
            if val is None:
                missing.append(f"{fname}:{ln} {name}")
                val = "UNMAPPED"
            if not val.startswith("not mirrored") and not val.startswith(NB) and "not mirrored as statistics" not in val:
                mirrored += 1
            rows.append(f"| `{fname}:{ln}` {name} | {val} |")
        out += [f"## {fname}: {len(tests)} tests, {mirrored} mirrored, {len(tests) - mirrored} not mirrored (reason given)", "", "| reference test | mirrored by |", "|---|---|"] + rows + [""]
    # the reference's Python tests inside the boundary
    for fname, table in PY_MAP.items():
        lo, hi = table["_range"]
        tests, cls = [], None
        for i, line in enumerate(open(os.path.join(REF, fname)), 1):
            mc = re.match(r"class (\w+)", line)
            if mc:
                cls = mc.group(1)
            m = re.match(r"(\s*)def (test_\w+)\(", line)
            if m and lo <= i <= hi:
                tests.append((i, f"{cls}.{m.group(2)}" if (m.group(1) and cls) else m.group(2)))
        mirrored, rows = 0, []
        for ln, name in tests:
            val = lookup({k: v for k, v in table.items() if k != "_range"}, name)
            if val is None:
                missing.append(f"{fname}:{ln} {name}")
                val = "UNMAPPED"
            if not val.startswith("not mirrored"):
                mirrored += 1
            rows.append(f"| `{fname}:{ln}` {name} | {val} |")
        out += [f"## {fname}" + (f" (lines {lo}-{hi}: the cases inside the boundary)" if hi < 10 ** 9 else "") + f": {len(tests)} tests, {mirrored} mirrored, {len(tests) - mirrored} not mirrored (reason given)", "",
                "| reference test | mirrored by |", "|---|---|"] + rows + [""]
    if missing:
        sys.stderr.write("unmapped reference tests:\n  " + "\n  ".join(missing) + "\n")
        sys.exit(1)
    with open(os.path.join(HERE, "REFERENCE_TEST_COVERAGE.md"), "w") as f:
        f.write("\n".join(out))
    print("wrote tests/REFERENCE_TEST_COVERAGE.md")


if __name__ == "__main__":
    main()
