"""Leaf-evaluation network: plumbing around the reference's `NNArch` weights.

* `NetSpec` / `LeafNet` restate the inference forward of the reference's ResNet-mode
  `NNArch` (/root/reference/src/neural_net.py:233-263 ResidualBlock, :266-510 NNArch) with
  IDENTICAL parameter names, so a reference checkpoint's `state_dict` loads unchanged and a
  random-init net of the same architecture can be built without the reference tree.
* `process()` mirrors `NNWrapper.process` (neural_net.py:800-823): eval mode, optional bf16
  autocast, returns `exp(log_softmax)` as float32 probabilities.
* `fold()` folds inference BatchNorms into the convolutions and lays the weights out for the
  hand-written MFMA kernels in csrc/ (see DESIGN.md §Leaf net).

Only the ResNet trunk (`dense_net=False`, `trunk_norm="batch"`, `trunk_act="relu"`) with the value
head and the flat / spatial policy heads is covered: that is what BASELINE.json's configs use.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn


@dataclass
class NetSpec:
    in_shape: tuple          # CANONICAL_SHAPE (C, H, W)
    num_moves: int
    num_players: int
    num_channels: int = 64   # NNArgs.num_channels
    depth: int = 6           # NNArgs.depth
    kernel_size: int = 3
    head_channels: int = 32
    head_pool: bool = True
    v_fc_hidden: int = -1    # -1 -> head_channels * 8 (neural_net.py:92-95)
    v_head_convs: int = 0
    pi_head_convs: int = 0
    v_fc_layers: int = 1
    policy_shape: tuple = None  # POLICY_SHAPE (C, H, W) -> spatial head (neural_net.py:390-427)
    pi_fc_hidden: int = -1   # -1 -> head_channels * 8; hidden width of pi_global (spatial head with global actions)

    def __post_init__(self):
        if self.v_fc_hidden == -1:
            self.v_fc_hidden = self.head_channels * 8
        if self.pi_fc_hidden == -1:
            self.pi_fc_hidden = self.head_channels * 8


def _conv(cin, cout, k):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding="same", bias=False)  # neural_net.py:147-155


class ResidualBlock(nn.Module):  # neural_net.py:233-263 (no downsample)
    def __init__(self, ch, k):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(ch)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = _conv(ch, ch, k)
        self.bn2 = nn.BatchNorm2d(ch)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = _conv(ch, ch, k)

    def forward(self, x):
        out = self.conv1(self.relu1(self.bn1(x)))
        out = self.conv2(self.relu2(self.bn2(out)))
        return out + x


class LeafNet(nn.Module):
    def __init__(self, spec: NetSpec):
        super().__init__()
        self.spec = spec
        C, H, W = spec.in_shape
        ch, HC, k = spec.num_channels, spec.head_channels, spec.kernel_size
        self.conv1 = _conv(C, ch, k)
        self.bn1 = nn.BatchNorm2d(ch)
        self.conv_layers = nn.Sequential(*[ResidualBlock(ch, k) for _ in range(spec.depth)])
        self.v_conv = _conv(ch, HC, 1)
        self.pi_conv = _conv(ch, HC, 1)
        self.v_bn = nn.BatchNorm2d(HC)
        self.v_relu = nn.ReLU(inplace=True)
        if spec.v_head_convs > 0:
            layers = []
            for _ in range(spec.v_head_convs):
                layers += [_conv(HC, HC, k), nn.BatchNorm2d(HC), nn.ReLU(inplace=True)]
            self.v_extra_convs = nn.Sequential(*layers)
        self.v_pool = nn.AdaptiveAvgPool2d(1) if spec.head_pool else None
        self.v_flatten = nn.Flatten()
        self.v_fc1 = nn.Linear(HC if spec.head_pool else HC * H * W, spec.v_fc_hidden)
        self.v_fc1_relu = nn.ReLU(inplace=True)
        if spec.v_fc_layers > 1:
            extra = []
            for _ in range(spec.v_fc_layers - 1):
                extra += [nn.Linear(spec.v_fc_hidden, spec.v_fc_hidden), nn.ReLU(inplace=True)]
            self.v_fc_extra = nn.Sequential(*extra)
        self.v_fc2 = nn.Linear(spec.v_fc_hidden, spec.num_players + 1)
        self.pi_bn = nn.BatchNorm2d(HC)
        self.pi_relu = nn.ReLU(inplace=True)
        if spec.pi_head_convs > 0:
            layers = []
            for _ in range(spec.pi_head_convs):
                layers += [_conv(HC, HC, k), nn.BatchNorm2d(HC), nn.ReLU(inplace=True)]
            self.pi_extra_convs = nn.Sequential(*layers)
        if spec.policy_shape is not None:
            pc, ph, pw = spec.policy_shape
            assert (ph, pw) == (H, W) and pc * ph * pw <= spec.num_moves
            self.pi_conv2 = _conv(HC, pc, 1)
            self.pi_bn2 = nn.BatchNorm2d(pc)
            self.num_global_actions = spec.num_moves - pc * ph * pw
            if self.num_global_actions > 0:   # neural_net.py:413-426: deploys + end turn of StarGambit
                self.pi_flatten = nn.Flatten()
                self.pi_pool = nn.AdaptiveAvgPool2d(1) if spec.head_pool else None
                self.pi_global = nn.Sequential(nn.Linear(HC if spec.head_pool else HC * H * W, spec.pi_fc_hidden), nn.ReLU(inplace=True),
                                               nn.Linear(spec.pi_fc_hidden, self.num_global_actions), nn.LayerNorm(self.num_global_actions))
        else:
            self.pi_flatten = nn.Flatten()
            self.pi_fc1 = nn.Linear(H * W * HC, spec.num_moves)

    def forward(self, s):  # neural_net.py:448-510 — returns LOG-probabilities like the reference
        s = self.bn1(self.conv1(s))
        s = self.conv_layers(s)
        v = self.v_relu(self.v_bn(self.v_conv(s)))
        if hasattr(self, "v_extra_convs"):
            v = self.v_extra_convs(v)
        if self.v_pool is not None:
            v = self.v_pool(v)
        v = self.v_fc1_relu(self.v_fc1(self.v_flatten(v)))
        if hasattr(self, "v_fc_extra"):
            v = self.v_fc_extra(v)
        v = torch.log_softmax(self.v_fc2(v), dim=1)
        pi = self.pi_relu(self.pi_bn(self.pi_conv(s)))
        if hasattr(self, "pi_extra_convs"):
            pi = self.pi_extra_convs(pi)
        if self.spec.policy_shape is not None:
            sp = self.pi_bn2(self.pi_conv2(pi)).permute(0, 2, 3, 1).reshape(pi.shape[0], -1)
            if self.num_global_actions > 0:   # neural_net.py:486-493
                flat = self.pi_flatten(self.pi_pool(pi) if self.pi_pool is not None else pi)
                sp = torch.cat([sp, self.pi_global(flat)], dim=1)
            pi = sp
        else:
            pi = self.pi_fc1(self.pi_flatten(pi))
        return v, torch.log_softmax(pi, dim=1)

    @torch.no_grad()
    def process(self, batch, amp_dtype=None):
        """NNWrapper.process (neural_net.py:800-823): probabilities as float32."""
        self.eval()
        if amp_dtype is not None:
            with torch.amp.autocast(batch.device.type, dtype=amp_dtype):
                v, pi = self(batch)
        else:
            v, pi = self(batch)
        return torch.exp(v).float(), torch.exp(pi).float()


def connect4_spec(depth=6, channels=64, kernel_size=3, head_channels=32):
    """BASELINE config 2: Connect4, 6 blocks x 64 channels, k=3, flat policy head."""
    return NetSpec(in_shape=(4, 6, 7), num_moves=7, num_players=2, num_channels=channels, depth=depth,
                   kernel_size=kernel_size, head_channels=head_channels)


def tawlbwrdd_spec(depth=4):
    """BASELINE config 3 / configs/tawlbwrdd.yaml:6-16: 4 blocks x 64 channels, k=3, head_channels 64,
    one extra conv per head, two value FC layers, spatial policy head (POLICY_SHAPE 22 x 11 x 11)."""
    return NetSpec(in_shape=(7, 11, 11), num_moves=2662, num_players=2, num_channels=64, depth=depth, kernel_size=3,
                   head_channels=64, v_head_convs=1, pi_head_convs=1, v_fc_layers=2, policy_shape=(22, 11, 11))


def brandubh_spec():
    """configs/brandubh.yaml: 4 blocks x 32 channels, head_channels 32, extra head convs, two value FC layers, spatial head."""
    return NetSpec(in_shape=(7, 7, 7), num_moves=686, num_players=2, num_channels=32, depth=4, kernel_size=3, head_channels=32,
                   v_head_convs=1, pi_head_convs=1, v_fc_layers=2, policy_shape=(14, 7, 7))


def opentafl_spec(depth=4, channels=64, head_channels=64):
    """OpenTafl: 8 canonical planes (plane 7 = turn / max_turns), 11x11, spatial head (the Tawlbwrdd YAML shape)."""
    return NetSpec(in_shape=(8, 11, 11), num_moves=2662, num_players=2, num_channels=channels, depth=depth, kernel_size=3,
                   head_channels=head_channels, v_head_convs=1, pi_head_convs=1, v_fc_layers=2, policy_shape=(22, 11, 11))


def stargambit_spec(depth=4, channels=64, head_channels=64):
    """BASELINE config 5 / configs/star_gambit_unified.yaml:5-15: 4 blocks x 64 channels, k=3, head_channels 64, one extra conv
    per head, two value FC layers, spatial policy head (POLICY_SHAPE 10 x 13 x 13) + 19 global actions (deploys, end turn)."""
    return NetSpec(in_shape=(36, 13, 13), num_moves=1709, num_players=2, num_channels=channels, depth=depth, kernel_size=3,
                   head_channels=head_channels, v_head_convs=1, pi_head_convs=1, v_fc_layers=2, policy_shape=(10, 13, 13))


def random_init(spec, seed=0, randomize_bn=True):
    """Random-init net of the named architecture (no checkpoint / dataset is reachable here).
    BatchNorm statistics and affine parameters are randomised too so that BN folding is exercised."""
    torch.manual_seed(seed)
    net = LeafNet(spec)
    if randomize_bn:
        g = torch.Generator().manual_seed(seed + 1)
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) * 1.0 + 0.5)
                m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.8 + 0.6)
                m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return net.eval()


def bn_affine(bn):
    """Inference BatchNorm as y = a * x + b."""
    a = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    b = bn.bias.detach().double() - bn.running_mean.detach().double() * a
    return a, b
