"""CPU: the reference's sample file format (`.ptz` triples, game_runner.py:200-217, 736-747, 280-289) written without the
`zstandard` package: the frame is a valid Zstandard frame in store mode."""
import os
import struct
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))


def test_store_mode_frame_layout_and_round_trip():
    from alphazero import history_io as h
    for n in (0, 1, 1000, 128 * 1024, 128 * 1024 + 1, 300_000):
        data = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8).tobytes()
        frame = h.zstd_store(data)
        assert struct.unpack_from("<I", frame, 0)[0] == 0xFD2FB528            # RFC 8878 magic number
        assert frame[4] == 0xC0 and frame[5] == 0x38                          # 8-byte content size, 128 KiB window
        assert struct.unpack_from("<Q", frame, 6)[0] == n                     # Frame_Content_Size
        blocks = max(1, -(-n // (128 * 1024)))
        assert len(frame) == 14 + 3 * blocks + n                              # raw blocks: 3-byte header + payload
        first = frame[14] | frame[15] << 8 | frame[16] << 16
        assert (first >> 1) & 3 == 0 and first >> 3 == min(n, 128 * 1024)     # Block_Type raw, Block_Size
        assert h.zstd_unstore(frame) == data
    two = h.zstd_store(b"abc") + h.zstd_store(b"def")                         # concatenated frames decode in sequence
    assert h.zstd_unstore(two) == b"abcdef"
    rle = struct.pack("<I", 0xFD2FB528) + bytes([0x20, 5]) + struct.pack("<I", (5 << 3) | (1 << 1) | 1)[:3] + b"z"
    assert h.zstd_unstore(rle) == b"zzzzz"                                    # a single-segment frame with one RLE block


def test_history_triples_round_trip_in_the_reference_layout(tmp_path):
    from alphazero import history_io as h
    g = torch.Generator().manual_seed(0)
    canon = (torch.rand((37, 4, 6, 7), generator=g) < 0.3).float()
    v = torch.nn.functional.one_hot(torch.randint(0, 3, (37,), generator=g), 3).float()
    pi = torch.softmax(torch.randn((37, 7), generator=g), 1)
    paths = h.write_history_batch(str(tmp_path), 12, 3, canon, v, pi)
    assert [os.path.basename(p) for p in paths] == ["0012-0003-canonical-37.ptz", "0012-0003-v-37.ptz", "0012-0003-pi-37.ptz"]
    (c_path, v_path, pi_path, size), = h.glob_file_triples(str(tmp_path))
    assert size == 37
    c2, v2, p2 = h.load_compressed(c_path), h.load_compressed(v_path), h.load_compressed(pi_path)
    assert c2.dtype == v2.dtype == p2.dtype == torch.float16                  # half storage (neural_net.py:13-36)
    assert torch.equal(c2.float(), canon) and torch.equal(v2.float(), v)
    assert (p2.float() - pi).abs().max() < 1e-3
    big = torch.tensor([1.0, 70000.0])                                        # would overflow float16: stays float32
    h.save_compressed(big, str(tmp_path / "big.ptz"))
    assert h.load_compressed(str(tmp_path / "big.ptz")).dtype == torch.float32


def test_store_frames_are_accepted_by_the_real_decoder_and_real_frames_are_read():
    """when the system has libzstd: its decoder reads the store-mode frames, and frames IT compresses (what the reference
    writes) are read back by load path"""
    import pytest
    from alphazero import history_io as h
    lz = h._libzstd()
    if lz is None:
        pytest.skip("no libzstd on this system")
    import ctypes
    for n in (0, 7, 200_000, 400_001):
        data = np.random.default_rng(n).integers(0, 4, n, dtype=np.uint8).tobytes()     # compressible
        frame = h.zstd_store(data)
        assert lz.ZSTD_getFrameContentSize(frame, len(frame)) == n
        out = ctypes.create_string_buffer(max(n, 1))
        r = lz.ZSTD_decompress(out, max(n, 1), frame, len(frame))
        assert not lz.ZSTD_isError(r) and out.raw[:n] == data
        real = h.zstd_compress(data, 1)
        assert h.zstd_decompress(real) == data and (n < 1000 or len(real) < n // 2)     # really compressed
