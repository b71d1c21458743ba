"""Generates tests/golden/nn_connect4_6b64c.npz by importing the REFERENCE's own neural_net.py
(/root/reference/src/neural_net.py) in this container.  The reference cannot travel to the GPU box,
so only data is committed: the random-init state_dict (torch.manual_seed(0), BatchNorm statistics
randomised so folding is exercised), 64 canonical Connect4 inputs and the reference's outputs
(NNArch in eval mode, fp32, CPU; probabilities = exp(log_softmax) as NNWrapper.process returns).

`alphazero` and `zstandard` are absent from this image; neural_net.py imports them at module level
only for Tracy hooks / checkpoint compression, so they are stubbed in sys.modules for the import.
Run:  python tests/golden/make_nn_fixture.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"

stub = types.ModuleType("alphazero")
stub.tracy_is_enabled = lambda: False
stub._tracy_zone_begin = lambda *a: None
stub._tracy_zone_end = lambda: None
stub.tracy_frame_mark = lambda: None
stub._tracy_set_thread_name = lambda n: None
sys.modules["alphazero"] = stub
sys.modules["zstandard"] = types.ModuleType("zstandard")
sys.path.insert(0, REF)
import neural_net as ref_nn  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as orc  # noqa: E402  (only to produce realistic canonical inputs)


class C4:  # the static surface NNArch reads from a game class
    @staticmethod
    def CANONICAL_SHAPE(): return (4, 6, 7)
    @staticmethod
    def NUM_PLAYERS(): return 2
    @staticmethod
    def NUM_MOVES(): return 7


PEAKED = "--peaked" in sys.argv   # scale the last layer of both heads: a random-init net is nearly uniform (pi in [0.10, 0.19]),
                                   # which no tolerance test can tell from a constant; the scaled nets spread over the simplex


def _peak(net):
    """x8 on the output layers' weights (the reference NNArch stays the producer of the expected outputs)"""
    with torch.no_grad():
        net.v_fc2.weight.mul_(8.0)
        if hasattr(net, "pi_fc1"):
            net.pi_fc1.weight.mul_(8.0)
        else:
            net.pi_bn2.weight.mul_(8.0)


def main():
    args = ref_nn.NNArgs(num_channels=64, depth=6, kernel_size=3, dense_net=False, head_channels=32)
    torch.manual_seed(0)
    net = ref_nn.NNArch(C4, args)
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 1.0 + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.8 + 0.6)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    if PEAKED:
        _peak(net)
    net.eval()
    rng = np.random.default_rng(3)
    xs = []
    while len(xs) < 64:
        game = orc.Game(orc.GAME_CONNECT4)
        for _ in range(int(rng.integers(0, 30))):
            if game.scores() is not None:
                break
            game.play(int(rng.choice(np.flatnonzero(game.valid()))))
        xs.append(game.canonical())
    x = torch.from_numpy(np.stack(xs))
    with torch.no_grad():
        v, pi = net(x)
        v, pi = torch.exp(v), torch.exp(pi)
    out = {"input": x.numpy(), "v": v.numpy(), "pi": pi.numpy()}
    for k, t in net.state_dict().items():
        out["sd." + k] = t.numpy()
    path = os.path.join(HERE, "nn_connect4_6b64c_peaked.npz" if PEAKED else "nn_connect4_6b64c.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(p.numel() for p in net.parameters()), "params")
    # the package's own LeafNet must accept this state_dict and reproduce the reference bit for bit
    sys.modules.pop("alphazero")
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    from alphazero import torch_net
    mine = torch_net.LeafNet(torch_net.connect4_spec())
    mine.load_state_dict(net.state_dict())
    v2, pi2 = mine.process(x)
    print("LeafNet vs reference NNArch: max |dv| =", float((v2 - v).abs().max()), " max |dpi| =", float((pi2 - pi).abs().max()))


class Tw:  # TawlbwrddGS statics (py_wrapper.cc:549-560); POLICY_SHAPE switches the spatial head on
    @staticmethod
    def CANONICAL_SHAPE(): return (7, 11, 11)
    @staticmethod
    def NUM_PLAYERS(): return 2
    @staticmethod
    def NUM_MOVES(): return 2662
    @staticmethod
    def POLICY_SHAPE(): return (22, 11, 11)


def main_tawlbwrdd():
    """configs/tawlbwrdd.yaml:6-16: 4 blocks x 64 ch, k3, head_channels 64, one extra conv per head,
    v_fc_layers 2, spatial policy head.  16 positions from oracle random playouts."""
    args = ref_nn.NNArgs(num_channels=64, depth=4, kernel_size=3, dense_net=False, head_channels=64,
                         v_head_convs=1, pi_head_convs=1, v_fc_layers=2, spatial_policy="on")
    torch.manual_seed(0)
    net = ref_nn.NNArch(Tw, args)
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 1.0 + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.8 + 0.6)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    if PEAKED:
        _peak(net)
    net.eval()
    rng = np.random.default_rng(4)
    xs = []
    while len(xs) < 16:
        game = orc.Game(orc.GAME_TAWLBWRDD)
        for _ in range(int(rng.integers(0, 60))):
            if game.scores() is not None:
                break
            game.play(int(rng.choice(np.flatnonzero(game.valid()))))
        xs.append(game.canonical())
    x = torch.from_numpy(np.stack(xs))
    with torch.no_grad():
        v, pi = net(x)
        v, pi = torch.exp(v), torch.exp(pi)
    out = {"input": x.numpy(), "v": v.numpy(), "pi": pi.numpy()}
    for k, t in net.state_dict().items():
        out["sd." + k] = t.numpy()
    path = os.path.join(HERE, "nn_tawlbwrdd_4b64c_peaked.npz" if PEAKED else "nn_tawlbwrdd_4b64c.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(p.numel() for p in net.parameters()), "params")
    sys.modules.pop("alphazero", None)
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    from alphazero import torch_net
    mine = torch_net.LeafNet(torch_net.tawlbwrdd_spec())
    mine.load_state_dict(net.state_dict())
    v2, pi2 = mine.process(x)
    print("LeafNet vs reference NNArch (tawlbwrdd): max |dv| =", float((v2 - v).abs().max()), " max |dpi| =", float((pi2 - pi).abs().max()))



def _tafl_fixture(name, game_id, statics, args_kw, spec_fn, n_pos, out_name, make_game=None, peak=False):
    """One more reference NNArch fixture: `statics` = the game class surface NNArch reads, `args_kw` = the YAML's net keys."""
    cls = type(name, (), {k: staticmethod((lambda v: (lambda: v))(v)) for k, v in statics.items()})
    args = ref_nn.NNArgs(dense_net=False, kernel_size=3, spatial_policy="on", **args_kw)
    torch.manual_seed(0)
    net = ref_nn.NNArch(cls, args)
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 1.0 + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.8 + 0.6)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    if peak:
        _peak(net)
        if hasattr(net, "pi_global"):
            with torch.no_grad():
                net.pi_global[3].weight.mul_(2.0)     # the global logits are LayerNorm outputs: spread them too
    net.eval()
    rng = np.random.default_rng(6)
    xs = []
    while len(xs) < n_pos:
        game = make_game(len(xs)) if make_game else orc.Game(game_id)
        for _ in range(int(rng.integers(0, 50))):
            if game.scores() is not None:
                break
            game.play(int(rng.choice(np.flatnonzero(game.valid()))))
        xs.append(game.canonical())
    x = torch.from_numpy(np.stack(xs))
    with torch.no_grad():
        v, pi = net(x)
        v, pi = torch.exp(v), torch.exp(pi)
    out = {"input": x.numpy(), "v": v.numpy(), "pi": pi.numpy()}
    for k, t in net.state_dict().items():
        out["sd." + k] = t.numpy()
    path = os.path.join(HERE, out_name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(p.numel() for p in net.parameters()), "params")
    sys.modules.pop("alphazero", None)
    sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
    from alphazero import torch_net
    mine = torch_net.LeafNet(getattr(torch_net, spec_fn)())
    mine.load_state_dict(net.state_dict())
    v2, pi2 = mine.process(x)
    print(f"LeafNet vs reference NNArch ({name}): max |dv| =", float((v2 - v).abs().max()), " max |dpi| =", float((pi2 - pi).abs().max()))


def main_opentafl():
    """configs/open_tafl.yaml: 4 blocks x 64 ch, head 64, one extra conv per head, v_fc_layers 2, spatial head; 8 planes."""
    _tafl_fixture("OpenTafl", orc.GAME_OPENTAFL,
                  dict(CANONICAL_SHAPE=(8, 11, 11), NUM_PLAYERS=2, NUM_MOVES=2662, POLICY_SHAPE=(22, 11, 11)),
                  dict(num_channels=64, depth=4, head_channels=64, v_head_convs=1, pi_head_convs=1, v_fc_layers=2),
                  "opentafl_spec", 12, "nn_opentafl_4b64c.npz")


def main_brandubh():
    """configs/brandubh.yaml: 4 blocks x 32 ch, head 32, one extra conv per head, v_fc_layers 2, spatial head; 7x7."""
    _tafl_fixture("Brandubh", orc.GAME_BRANDUBH,
                  dict(CANONICAL_SHAPE=(7, 7, 7), NUM_PLAYERS=2, NUM_MOVES=686, POLICY_SHAPE=(14, 7, 7)),
                  dict(num_channels=32, depth=4, head_channels=32, v_head_convs=1, pi_head_convs=1, v_fc_layers=2),
                  "brandubh_spec", 16, "nn_brandubh_4b32c.npz")

def main_stargambit():
    """configs/star_gambit_unified.yaml:5-15: 4 blocks x 64 ch, head 64, one extra conv per head, v_fc_layers 2, spatial head over
    the 13 x 13 canvas (10 channels) + the global head for the 19 deploy / end-turn actions; 36 planes.  Positions of all four
    variants from oracle random playouts."""
    _tafl_fixture("StarGambitUnified", orc.GAME_STARGAMBIT,
                  dict(CANONICAL_SHAPE=(36, 13, 13), NUM_PLAYERS=2, NUM_MOVES=1709, POLICY_SHAPE=(10, 13, 13)),
                  dict(num_channels=64, depth=4, head_channels=64, v_head_convs=1, pi_head_convs=1, v_fc_layers=2),
                  "stargambit_spec", 12, "nn_stargambit_4b64c_peaked.npz" if PEAKED else "nn_stargambit_4b64c.npz",
                  make_game=lambda i: orc.Game.sg_unified(pinned=i % 4), peak=PEAKED)


if __name__ == "__main__":
    mode = next((a for a in sys.argv[1:] if not a.startswith("--")), "connect4")
    {"connect4": main, "tawlbwrdd": main_tawlbwrdd, "opentafl": main_opentafl, "brandubh": main_brandubh,
     "stargambit": main_stargambit}[mode]()
