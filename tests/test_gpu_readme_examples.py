"""-m gpu: the two usage snippets of README.md run as written (smaller numbers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_readme_host_path_and_device_path(tmp_path):
    import alphazero
    params = alphazero.PlayParams()
    params.games_to_play, params.concurrent_games, params.max_batch_size = 16, 8, 8
    params.mcts_visits = [20, 20]; params.model_groups = [0, 0]; params.history_enabled = True
    pm = alphazero.PlayManager(alphazero.Connect4GS(), params)
    batch = np.zeros((8, 4, 6, 7), np.float32)
    my_model = lambda x: (np.full((len(x), 3), 1 / 3, np.float32), np.full((len(x), 7), 1 / 7, np.float32))
    while pm.remaining_games() > 0:
        idx = pm.build_batch(0, batch)
        if idx:
            v, pi = my_model(batch[:len(idx)])
            pm.update_inferences(0, idx, v, pi)
    canonical, v, pi = pm.history()
    assert pm.games_completed() == 16 and len(canonical) == len(v) == len(pi) > 16 * 7

    from alphazero import selfplay, torch_net
    net = alphazero.HipLeafNet(torch_net.random_init(torch_net.connect4_spec()), torch_net.connect4_spec())
    result, (canonical, v, pi) = selfplay.self_play(alphazero.Connect4GS, params, net, engines=4, data_folder=str(tmp_path / "history"))
    assert result.games == 16 and abs(sum(result.win_rates) - 1) < 1e-6 and result.game_length > 7
    other_net = alphazero.HipLeafNet(torch_net.random_init(torch_net.connect4_spec(), seed=5), torch_net.connect4_spec())
    match = selfplay.gating_match(alphazero.Connect4GS, params, net, other_net)
    assert match.n_games == 16

    gs = alphazero.Connect4GS()
    m = alphazero.MCTS(1.25, 2, 7)
    leaf = m.find_leaf(gs)
    vv, pp = my_model(np.zeros((1, 4, 6, 7)))
    m.process_result(gs, vv[0], pp[0])
    assert m.depth() == 1


def test_readme_stargambit_snippet():
    import alphazero
    from alphazero import selfplay, torch_net
    params = alphazero.PlayParams()
    params.games_to_play, params.concurrent_games, params.max_batch_size = 4, 4, 4
    params.mcts_visits = [12, 12]; params.model_groups = [0, 0]; params.history_enabled = True
    params.temp_decay_half_life_by_variant = [3, 4, 5, 8]
    sg_net = alphazero.HipLeafNet(torch_net.random_init(torch_net.stargambit_spec()), torch_net.stargambit_spec())
    result, samples = selfplay.self_play(alphazero.StarGambitUnifiedGS(-1, [0.25] * 4), params, sg_net, engines=2)
    assert result.games == 4 and sum(result.variant_game_counts.values()) == 4 and samples[0].shape[1:] == (36, 13, 13)
