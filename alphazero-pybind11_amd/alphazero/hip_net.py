"""Host side of the fused MFMA leaf net (csrc/leafnet.hip): BatchNorm folding, bf16 conversion and
the MFMA-fragment weight layout, plus the ctypes wrapper `HipLeafNet`.

Weight blob layout (little-endian, in this order; `frag` = [k-step][m-tile][lane 0..63][8 bf16] where
element j of lane l is  W[co = 16*mt + (l & 15)][k = 32*ks + 8*(l >> 4) + j]):
  stem    frag[2][4]   W[co][k = tap*C_in + ci] (k >= 9*C_in zero)      + bias[64] f32   (bn1 folded)
  block i a1[64] b1[64] c1[64] f32 | conv1 frag[18][4] (bn2 folded, k = tap*64 + ci) | conv2 frag[18][4]
  heads   frag[2][4]   rows 0-31 = v_conv * v_bn, rows 32-63 = pi_conv * pi_bn + bias[64] f32
  v_fc1 W^T[32][hidden] b[hidden] | v_fc2 W[P+1][hidden] b[P+1]  (f32) | pi_fc1: per pixel p frag hi(W[m][c*HW+p]) frag lo | b[M] f32
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import lib
from .torch_net import bn_affine


class NetDescC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_channels", "height", "width", "channels", "depth", "kernel_size",
                                         "head_channels", "v_hidden", "num_moves", "num_players", "v_head_convs",
                                         "pi_head_convs", "v_fc_layers", "policy_channels", "precision", "pi_hidden")]


lib.azmi_net_blob_bytes.restype = C.c_size_t
lib.azmi_net_blob_bytes.argtypes = [C.POINTER(NetDescC)]
lib.azmi_net_create.restype = C.c_int
lib.azmi_net_create.argtypes = [C.POINTER(NetDescC), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
lib.azmi_net_destroy.restype = None
lib.azmi_net_destroy.argtypes = [C.c_void_p]
lib.azmi_net_forward.restype = C.c_int
lib.azmi_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
lib.azmi_net_last_error.restype = C.c_char_p


def _bf16_bits(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(torch.bfloat16).view(torch.int16).numpy()


def _frags(wmat):
    """wmat [16*MT][K] float -> bf16 fragments [K/32][MT][64][8] as bytes."""
    co, K = wmat.shape
    assert co % 16 == 0 and K % 32 == 0
    nmt = co // 16
    lanes = np.arange(64)
    out = np.zeros((K // 32, nmt, 64, 8), np.int16)
    bits = _bf16_bits(wmat)
    for ks in range(K // 32):
        for mt in range(nmt):
            rows = 16 * mt + (lanes & 15)
            cols = 32 * ks + 8 * (lanes >> 4)
            out[ks, mt] = np.stack([bits[rows, cols + j] for j in range(8)], axis=1)
    return out.tobytes()


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32)).tobytes()


def _f32_frags(w):
    """w [N][K] (rows = outputs, N % 16 == 0, K % 16 == 0) -> f32 MFMA A-fragments [N/16][K/16][64 lanes][4]:
    element j of lane l = w[16*tile + (l & 15)][16*group + 4*j + (l >> 4)] (v_mfma_f32_16x16x4_f32, k-step j)."""
    w = np.asarray(w, dtype=np.float64)
    N, K = w.shape
    assert N % 16 == 0 and K % 16 == 0
    lanes = np.arange(64)
    out = np.zeros((N // 16, K // 16, 64, 4), np.float32)
    for t in range(N // 16):
        rows = 16 * t + (lanes & 15)
        for g in range(K // 16):
            for j in range(4):
                out[t, g, :, j] = w[rows, 16 * g + 4 * j + (lanes >> 4)]
    return out.tobytes()


def _pad(t, *shape):
    """t zero-padded (at the high end of every axis) to `shape`, float64."""
    t = torch.as_tensor(t, dtype=torch.float64)
    out = torch.zeros(shape, dtype=torch.float64)
    out[tuple(slice(0, n) for n in t.shape)] = t
    return out


def _conv_frags(w):
    """conv weight [co][ci][3][3] (BatchNorm scale already folded) -> 9 chunks of 8 KB: k = tap*64 + ci, channels zero-padded to 64."""
    return _frags(_pad(w.permute(0, 2, 3, 1), 64, 3, 3, 64).reshape(64, 576).numpy())


def fold_spatial(net, x3=False):
    """Spatial-policy-head nets (Tafl family, StarGambit: one extra conv per head, v_fc_layers >= 1; trunk and head widths up
    to 64 channels - narrower ones (configs/brandubh.yaml: 32) are zero-padded: the padded channels have zero weights, scales
    and biases, so they stay exactly 0 through every affine / ReLU / conv).  Image read by csrc/leafnet_sp.h:
      stream of 8 KB chunks frag[2 k-steps][4 m-tiles]: stem (2: k = tap*8 + ci for <= 8 input planes; else 9: one more 64-channel
        conv) | per block conv1 (9) conv2 (9) | value-head 1x1 (1) |
        policy-head 1x1 (1) | value extra conv (9) | policy extra conv (9) | policy 1x1 (1, rows >= policy channels zero)
      fp32: stem_b[64] | per block a1 b1 c1 | head_b[128] | vx_b[64] | px_b[64] | pol_b[32]
      value FC: fc1 f32-frag[Hd/16][4] b | extra FC (f32-frag[Hd/16][Hd/16])* then (b[Hd])* | fc2 f32-frag[1][Hd/16] b[16]
      pi_global (StarGambit): W1 f32-frag[Hp/16][4] b1 | W2 f32-frag[2][Hp/16] b2[32] | LayerNorm g[32] b[32]."""
    spec = net.spec
    Cin, H, W = spec.in_shape
    if not (spec.num_channels <= 64 and spec.head_channels <= 64 and spec.kernel_size == 3 and spec.head_pool and Cin <= 64
            and spec.v_head_convs == 1 and spec.pi_head_convs == 1 and (H, W) in ((11, 11), (7, 7), (13, 13))):
        raise RuntimeError("the bf16 MFMA spatial-head kernel covers the configs/tawlbwrdd.yaml, configs/open_tafl.yaml, "
                           "configs/brandubh.yaml and configs/star_gambit_unified.yaml nets (11x11, 7x7 or 13x13, <= 64 trunk / head "
                           "channels, one extra conv per head); use precision='fp32' for other shapes")
    pc = spec.policy_shape[0]
    Hd, L, P1 = spec.v_fc_hidden, spec.v_fc_layers, spec.num_players + 1
    sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    stream, prm = bytearray(), bytearray()
    # x3 (precision "bf16x3", csrc/leafnet_sp.h Geo<.., SPLIT>): every 8 KB chunk three times - high parts, high parts, low parts
    _frags = _split_frags if x3 else globals()["_frags"]
    _conv_frags = (lambda w_: _split_frags(_pad(w_.permute(0, 2, 3, 1), 64, 3, 3, 64).reshape(64, 576).numpy())) if x3 else globals()["_conv_frags"]
    a, b = (t.cpu() for t in bn_affine(net.bn1))
    w = sd["conv1.weight"] * a[:, None, None, None]
    if Cin <= 8:    # k = tap * 8 + ci, K = 72 padded to 128 (csrc/leafnet_sp.h stem_chunks)
        stream += _frags(_pad(_pad(w.permute(0, 2, 3, 1), 64, 3, 3, 8).reshape(64, 72), 64, 128).numpy())
    else:
        stream += _conv_frags(w)
    prm += _f32(_pad(b, 64))
    for i, blk in enumerate(net.conv_layers):
        a1, b1 = (t.cpu() for t in bn_affine(blk.bn1))
        a2, b2 = (t.cpu() for t in bn_affine(blk.bn2))
        stream += _conv_frags(sd[f"conv_layers.{i}.conv1.weight"] * a2[:, None, None, None])
        stream += _conv_frags(sd[f"conv_layers.{i}.conv2.weight"])
        prm += _f32(_pad(a1, 64)) + _f32(_pad(b1, 64)) + _f32(_pad(b2, 64))
    av, bv = (t.cpu() for t in bn_affine(net.v_bn))
    ap, bp = (t.cpu() for t in bn_affine(net.pi_bn))
    stream += _frags(_pad(sd["v_conv.weight"][:, :, 0, 0] * av[:, None], 64, 64).numpy())
    stream += _frags(_pad(sd["pi_conv.weight"][:, :, 0, 0] * ap[:, None], 64, 64).numpy())
    prm += _f32(torch.cat([_pad(bv, 64), _pad(bp, 64)]))
    for seq, name in ((net.v_extra_convs, "v_extra_convs"), (net.pi_extra_convs, "pi_extra_convs")):
        a, b = (t.cpu() for t in bn_affine(seq[1]))
        stream += _conv_frags(sd[f"{name}.0.weight"] * a[:, None, None, None])
        prm += _f32(_pad(b, 64))
    a2, b2 = (t.cpu() for t in bn_affine(net.pi_bn2))
    stream += _frags(_pad(sd["pi_conv2.weight"][:, :, 0, 0] * a2[:, None], 64, 64).numpy())
    prm += _f32(_pad(b2, 32))
    assert len(stream) == (3 if x3 else 1) * ((2 if Cin <= 8 else 9) + 18 * spec.depth + 2 + 18 + 1) * 8192
    blob = stream + prm
    blob += _f32_frags(_pad(sd["v_fc1.weight"], Hd, 64).numpy()) + _f32(sd["v_fc1.bias"])
    for l in range(L - 1):
        blob += _f32_frags(sd[f"v_fc_extra.{2 * l}.weight"].numpy())
    for l in range(L - 1):
        blob += _f32(sd[f"v_fc_extra.{2 * l}.bias"])
    blob += _f32_frags(_pad(sd["v_fc2.weight"], 16, Hd).numpy()) + _f32(_pad(sd["v_fc2.bias"], 16))
    G = spec.num_moves - pc * H * W
    Hp = spec.pi_fc_hidden if G > 0 else 0
    if G > 0:   # pi_global (neural_net.py:421-426): head channels zero-padded to 64, outputs to 32
        blob += _f32_frags(_pad(sd["pi_global.0.weight"], Hp, 64).numpy()) + _f32(sd["pi_global.0.bias"])
        blob += _f32_frags(_pad(sd["pi_global.2.weight"], 32, Hp).numpy()) + _f32(_pad(sd["pi_global.2.bias"], 32))
        blob += _f32(_pad(sd["pi_global.3.weight"], 32)) + _f32(_pad(sd["pi_global.3.bias"], 32))
    desc = NetDescC(Cin, H, W, 64, spec.depth, 3, 64, Hd, spec.num_moves, spec.num_players, 1, 1, L, pc, 2 if x3 else 0, Hp)
    assert len(blob) == lib.azmi_net_blob_bytes(C.byref(desc)), (len(blob), lib.azmi_net_blob_bytes(C.byref(desc)))
    return desc, bytes(blob)


def fold_fp32(net):
    """fp32 path (csrc/leafnet_f32.hip): torch layouts, BatchNorms folded in double, no bf16 anywhere."""
    spec = net.spec
    Cin, H, W = spec.in_shape
    assert spec.kernel_size == 3 and spec.head_pool
    sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    blob = bytearray()

    def conv_bn(wname, bn):
        a, b = (t.cpu() for t in bn_affine(bn))
        return _f32(sd[wname] * a[:, None, None, None]) + _f32(b)

    blob += conv_bn("conv1.weight", net.bn1)
    for i, blk in enumerate(net.conv_layers):
        a1, b1 = (t.cpu() for t in bn_affine(blk.bn1))
        blob += _f32(a1) + _f32(b1) + conv_bn(f"conv_layers.{i}.conv1.weight", blk.bn2) + _f32(sd[f"conv_layers.{i}.conv2.weight"])
    blob += conv_bn("v_conv.weight", net.v_bn)
    for i in range(spec.v_head_convs):
        blob += conv_bn(f"v_extra_convs.{3 * i}.weight", net.v_extra_convs[3 * i + 1])
    blob += _f32(sd["v_fc1.weight"]) + _f32(sd["v_fc1.bias"])
    for l in range(spec.v_fc_layers - 1):
        blob += _f32(sd[f"v_fc_extra.{2 * l}.weight"]) + _f32(sd[f"v_fc_extra.{2 * l}.bias"])
    blob += _f32(sd["v_fc2.weight"]) + _f32(sd["v_fc2.bias"])
    blob += conv_bn("pi_conv.weight", net.pi_bn)
    for i in range(spec.pi_head_convs):
        blob += conv_bn(f"pi_extra_convs.{3 * i}.weight", net.pi_extra_convs[3 * i + 1])
    pc = 0
    if spec.policy_shape is not None:
        pc = spec.policy_shape[0]
        blob += conv_bn("pi_conv2.weight", net.pi_bn2)
        if spec.num_moves > pc * H * W:   # pi_global: torch layouts
            for k in ("pi_global.0.weight", "pi_global.0.bias", "pi_global.2.weight", "pi_global.2.bias", "pi_global.3.weight", "pi_global.3.bias"):
                blob += _f32(sd[k])
    else:
        blob += _f32(sd["pi_fc1.weight"]) + _f32(sd["pi_fc1.bias"])
    desc = NetDescC(Cin, H, W, spec.num_channels, spec.depth, 3, spec.head_channels, spec.v_fc_hidden, spec.num_moves, spec.num_players,
                    spec.v_head_convs, spec.pi_head_convs, spec.v_fc_layers, pc, 1,
                    spec.pi_fc_hidden if (pc and spec.num_moves > pc * H * W) else 0)
    assert len(blob) == lib.azmi_net_blob_bytes(C.byref(desc)), (len(blob), lib.azmi_net_blob_bytes(C.byref(desc)))
    return desc, bytes(blob)


def _split_frags(wmat, passes=("hi", "hi", "lo")):
    """wmat [64][K] (K = 64 * taps) -> the bf16x3 weight stream of one convolution: per tap (two k-steps = one 8 KB chunk) the
    high parts' fragments, the high parts' again, the low parts' (w = hi + lo; csrc/leafnet_c4.h SPLIT: the three chunks meet
    the activations' high, low and high planes)."""
    w = torch.as_tensor(np.ascontiguousarray(wmat), dtype=torch.float64)
    hi = w.float().to(torch.bfloat16).double()
    lo = w - hi
    fh = np.frombuffer(_frags(hi.numpy()), np.int16).reshape(-1, 2, 4, 64, 8)     # [tap][2 k-steps][4 m-tiles][64 lanes][8]
    fl = np.frombuffer(_frags(lo.numpy()), np.int16).reshape(-1, 2, 4, 64, 8)
    out = bytearray()
    for t in range(fh.shape[0]):
        for which in passes:
            out += (fh if which == "hi" else fl)[t].tobytes()
    return bytes(out)


def fold(net, precision="bf16"):
    """LeafNet (reference NNArch parameter names) -> (NetDescC, blob bytes).  precision: "bf16" (bf16 MFMA operands), "bf16x3"
    (weights and activations as bf16 high + low parts, three MFMAs per product - the 1e-5 tier on the matrix cores; the Connect4
    family and, since round 4, the spatial-head nets) or "fp32" (plain fp32 kernels, any shape)."""
    if precision == "fp32":
        return fold_fp32(net)
    spec = net.spec
    x3 = precision == "bf16x3"
    if precision not in ("bf16", "bf16x3"):
        raise ValueError("precision must be 'bf16', 'bf16x3' or 'fp32'")
    if spec.policy_shape is not None:
        return fold_spatial(net, x3)
    Cin, H, W = spec.in_shape
    if not (spec.num_channels == 64 and spec.head_channels == 32 and spec.kernel_size == 3 and spec.policy_shape is None
            and spec.head_pool and spec.v_head_convs == 0 and spec.pi_head_convs == 0 and spec.v_fc_layers == 1
            and 9 * Cin <= 64 and (H, W) == (6, 7)):
        raise RuntimeError("the bf16 MFMA flat-head kernel covers the Connect4 net family (6x7, 64 trunk / 32 head channels, "
                           "flat policy head); use precision='fp32' for other shapes")
    sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    blob = bytearray()
    # stem: conv1 * bn1
    a, b = bn_affine(net.bn1)
    a, b = a.cpu(), b.cpu()
    w = sd["conv1.weight"] * a[:, None, None, None]                  # [64][Cin][3][3]
    wm = np.zeros((64, 64))
    wm[:, : 9 * Cin] = w.permute(0, 2, 3, 1).reshape(64, 9 * Cin).numpy()  # k = tap*Cin + ci
    blob += (_split_frags(wm, ("hi", "lo")) if x3 else _frags(wm)) + _f32(b)       # (inputs are 0 / 1: only the weights split)
    conv = (lambda m: _split_frags(m)) if x3 else _frags
    for i, blk in enumerate(net.conv_layers):
        a1, b1 = (t.cpu() for t in bn_affine(blk.bn1))
        a2, b2 = (t.cpu() for t in bn_affine(blk.bn2))
        w1 = sd[f"conv_layers.{i}.conv1.weight"] * a2[:, None, None, None]
        w2 = sd[f"conv_layers.{i}.conv2.weight"]
        blob += _f32(a1) + _f32(b1) + _f32(b2)
        blob += conv(w1.permute(0, 2, 3, 1).reshape(64, 576).numpy())   # k = tap*64 + ci
        blob += conv(w2.permute(0, 2, 3, 1).reshape(64, 576).numpy())
    av, bv = (t.cpu() for t in bn_affine(net.v_bn))
    ap, bp = (t.cpu() for t in bn_affine(net.pi_bn))
    wh = torch.cat([sd["v_conv.weight"][:, :, 0, 0] * av[:, None], sd["pi_conv.weight"][:, :, 0, 0] * ap[:, None]], 0)
    blob += conv(wh.numpy()) + _f32(torch.cat([bv, bp]))
    # value fc1 for v_mfma_f32_16x16x4_f32 (csrc/leafnet_c4.h): A-fragments [hidden / 16][2][64 lanes][4]
    if spec.v_fc_hidden % 16:
        raise RuntimeError("the Connect4-family tile wants v_fc_hidden to be a multiple of 16")
    blob += _f32_frags(sd["v_fc1.weight"].numpy()) + _f32(sd["v_fc1.bias"])
    blob += _f32(sd["v_fc2.weight"]) + _f32(sd["v_fc2.bias"])
    # flat policy head: the reference's feature order is (c, h, w) -> c*HW + p.  The kernel contracts one pixel position p at
    # a time on the bf16 matrix pipe, logits[m] += W_p[m][c] h[c][p], with W and h split into bf16 high + low parts:
    # per p one A-fragment of the high parts of W_p (16 rows, rows >= num_moves zero), then one of the low parts
    wp = torch.zeros((16, 32, H * W), dtype=torch.float64)
    wp[: spec.num_moves] = sd["pi_fc1.weight"].reshape(spec.num_moves, 32, H * W)
    hi = wp.float().to(torch.bfloat16).double()
    lo = wp - hi
    for p_ in range(H * W):
        blob += _frags(hi[:, :, p_].numpy()) + _frags(lo[:, :, p_].numpy())
    blob += _f32(sd["pi_fc1.bias"])
    desc = NetDescC(Cin, H, W, 64, spec.depth, 3, 32, spec.v_fc_hidden, spec.num_moves, spec.num_players, 0, 0, 1, 0, 2 if x3 else 0, 0)
    assert len(blob) == lib.azmi_net_blob_bytes(C.byref(desc)), (len(blob), lib.azmi_net_blob_bytes(C.byref(desc)))
    return desc, bytes(blob)


class HipLeafNet:
    """The fused MFMA kernel as an evaluator: forward(canonical, v_out, pi_out) on device tensors."""

    def __init__(self, net, spec=None, max_batch=None, device=0, precision="bf16"):
        """precision: "bf16" = MFMA fast path (bf16 operands, fp32 accumulate); "bf16x3" = the same tile with split operands
        (hi + lo bf16 parts, three MFMAs per product; Connect4 family): the 1e-5 tier on the matrix cores; "fp32" = plain fp32
        kernels for any shape."""
        self.desc, blob = fold(net, precision)
        self._blob = blob
        h = C.c_void_p()
        rc = lib.azmi_net_create(C.byref(self.desc), blob, len(blob), int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError(lib.azmi_net_last_error().decode())
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.azmi_net_destroy(self._h)
            self._h = None

    def forward(self, canonical, v_out, pi_out, stream=None):
        n = canonical.shape[0]
        assert canonical.dtype == torch.float32 and canonical.is_contiguous() and v_out.is_contiguous() and pi_out.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        rc = lib.azmi_net_forward(self._h, canonical.data_ptr(), v_out.data_ptr(), pi_out.data_ptr(), n, C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(lib.azmi_net_last_error().decode())

    def process(self, canonical):
        """NNWrapper.process signature: returns (v, pi) probabilities as new float32 tensors."""
        n = canonical.shape[0]
        v = torch.empty((n, self.desc.num_players + 1), dtype=torch.float32, device=canonical.device)
        pi = torch.empty((n, self.desc.num_moves), dtype=torch.float32, device=canonical.device)
        self.forward(canonical.contiguous().float(), v, pi)
        return v, pi
