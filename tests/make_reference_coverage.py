"""Writes tests/REFERENCE_TEST_COVERAGE.md: every TEST of the reference's C++ test files for the hot path
(/root/reference/src/{mcts,play_manager,star_gambit_gs,star_gambit_unified_gs}_test.cc - read here as text, never shipped) mapped
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
        "MCTS.ShapedDirichletDistribution": "statistical test of the noise distribution: not mirrored as statistics; the shaped-Dirichlet draws are pinned draw by draw instead (tests/test_gpu_rng.py gamma stream vs real libstdc++, tests/test_gpu_parity_connect4.py option tiers with shaped noise: pcg32 position after every move)",
        "MCTS.ShapedDirichletAlphaDistribution": "as ShapedDirichletDistribution",
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
    if missing:
        sys.stderr.write("unmapped reference tests:\n  " + "\n  ".join(missing) + "\n")
        sys.exit(1)
    with open(os.path.join(HERE, "REFERENCE_TEST_COVERAGE.md"), "w") as f:
        f.write("\n".join(out))
    print("wrote tests/REFERENCE_TEST_COVERAGE.md")


if __name__ == "__main__":
    main()
