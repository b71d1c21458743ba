"""The reference's on-disk sample format (game_runner.py:200-217, 736-747, 280-289): each batch of finished
samples is three files `{iteration:04d}-{batch:04d}-{canonical|v|pi}-{size}.ptz`, a `.ptz` being a Zstandard frame around
`torch.save(tensor)` with the tensor stored as float16 (float32 if a value would overflow).

`zstandard` (the Python package) is not installed in every image this runs in.  Order of preference: the package, used exactly
like the reference (level 1); the system's libzstd through ctypes (ZSTD_compress / ZSTD_decompress, the same frames); and,
with neither, STORE mode — a valid Zstandard frame made of raw blocks (RFC 8878 §3.1.1.2.2), which any zstd decoder, the
reference's `load_compressed` included, reads back — read by a small parser that understands raw and RLE blocks (a compressed
block then needs a real decoder and says so)."""
import glob
import io
import os
import struct

import torch

_MAGIC = 0xFD2FB528
_BLOCK_MAX = 128 * 1024


def _zstd():
    try:
        import zstandard
        return zstandard
    except ImportError:
        return None


_LIBZSTD = None


def _libzstd():
    """The system's libzstd through ctypes, or None."""
    global _LIBZSTD
    if _LIBZSTD is None:
        import ctypes
        import ctypes.util
        _LIBZSTD = False
        name = ctypes.util.find_library("zstd")
        if name:
            try:
                z = ctypes.CDLL(name)
                z.ZSTD_compressBound.restype = ctypes.c_size_t; z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
                z.ZSTD_compress.restype = ctypes.c_size_t
                z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
                z.ZSTD_decompress.restype = ctypes.c_size_t
                z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
                z.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
                z.ZSTD_getFrameContentSize.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
                z.ZSTD_isError.restype = ctypes.c_uint; z.ZSTD_isError.argtypes = [ctypes.c_size_t]
                _LIBZSTD = z
            except (OSError, AttributeError):
                _LIBZSTD = False
    return _LIBZSTD or None


def zstd_compress(data: bytes, level=1) -> bytes:
    z = _zstd()
    if z:
        return z.ZstdCompressor(level=level, threads=-1).compress(data)
    lz = _libzstd()
    if lz:
        import ctypes
        cap = lz.ZSTD_compressBound(len(data))
        dst = ctypes.create_string_buffer(cap)
        n = lz.ZSTD_compress(dst, cap, data, len(data), int(level))
        if not lz.ZSTD_isError(n):
            return dst.raw[:n]
    return zstd_store(data)


def zstd_decompress(blob: bytes) -> bytes:
    z = _zstd()
    if z:
        return z.ZstdDecompressor().decompress(blob)
    lz = _libzstd()
    if lz:
        import ctypes
        size = lz.ZSTD_getFrameContentSize(blob, len(blob))
        if size < (1 << 62):                                   # not ZSTD_CONTENTSIZE_UNKNOWN / _ERROR
            dst = ctypes.create_string_buffer(max(int(size), 1))
            n = lz.ZSTD_decompress(dst, max(int(size), 1), blob, len(blob))
            if not lz.ZSTD_isError(n):
                return dst.raw[:n]
    return zstd_unstore(blob)


def zstd_store(data: bytes) -> bytes:
    """One Zstandard frame holding `data` uncompressed: magic, header (window 128 KiB, 8-byte content size), raw blocks."""
    out = [struct.pack("<I", _MAGIC), bytes([0xC0, 0x38]), struct.pack("<Q", len(data))]
    n = len(data)
    if n == 0:
        out.append(struct.pack("<I", 1)[:3])                       # one empty raw block, last
    pos = 0
    while pos < n:
        size = min(_BLOCK_MAX, n - pos)
        last = 1 if pos + size == n else 0
        out.append(struct.pack("<I", (size << 3) | last)[:3])      # Block_Header: last, type 0 (raw), size
        out.append(data[pos:pos + size])
        pos += size
    return b"".join(out)


def zstd_unstore(blob: bytes) -> bytes:
    """Reads frames made of raw / RLE blocks (what zstd_store writes, and what zstd emits for incompressible input)."""
    pos, out = 0, []
    while pos < len(blob):
        (magic,) = struct.unpack_from("<I", blob, pos)
        if (magic & 0xFFFFFFF0) == 0x184D2A50:                    # skippable frame
            (sz,) = struct.unpack_from("<I", blob, pos + 4)
            pos += 8 + sz
            continue
        if magic != _MAGIC:
            raise ValueError("not a Zstandard frame")
        pos += 4
        fhd = blob[pos]; pos += 1
        fcs_flag, single, checksum, did = fhd >> 6, (fhd >> 5) & 1, (fhd >> 2) & 1, fhd & 3
        if not single:
            pos += 1                                               # Window_Descriptor
        pos += (0, 1, 2, 4)[did]
        pos += (1 if single else 0, 2, 4, 8)[fcs_flag]
        while True:
            h = blob[pos] | (blob[pos + 1] << 8) | (blob[pos + 2] << 16); pos += 3
            last, btype, size = h & 1, (h >> 1) & 3, h >> 3
            if btype == 0:
                out.append(blob[pos:pos + size]); pos += size
            elif btype == 1:
                out.append(blob[pos:pos + 1] * size); pos += 1
            else:
                raise RuntimeError("this .ptz holds compressed Zstandard blocks: install `zstandard` to read it")
            if last:
                break
        if checksum:
            pos += 4
    return b"".join(out)


def to_half_safe(tensor, dtype=torch.float16):                     # neural_net.py:18-36
    if not tensor.is_floating_point():
        return tensor
    if tensor.numel() == 0:
        return tensor.to(dtype)
    if tensor.abs().max().item() > torch.finfo(dtype).max:
        return tensor
    return tensor.to(dtype)


def save_compressed(tensor, path, half_storage=True, zstd_level=1):   # game_runner.py:200-210
    if half_storage:
        tensor = to_half_safe(tensor)
    tensor = tensor.detach().cpu().contiguous().clone()
    buf = io.BytesIO()
    torch.save(tensor, buf)
    frame = zstd_compress(buf.getvalue(), zstd_level)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(frame)
    os.replace(tmp, path)


def load_compressed(path):                                         # game_runner.py:213-217
    with open(path, "rb") as f:
        blob = f.read()
    data = zstd_decompress(blob)
    return torch.load(io.BytesIO(data), map_location="cpu", weights_only=True)


def write_history_batch(folder, iteration, batch, canonical, v, pi):
    """One batch of samples in the reference's layout (game_runner.py:736-747); returns the three paths."""
    os.makedirs(folder, exist_ok=True)
    size = int(canonical.shape[0])
    prefix = os.path.join(folder, f"{int(iteration):04d}-{int(batch):04d}")
    paths = []
    for name, t in (("canonical", canonical), ("v", v), ("pi", pi)):
        path = f"{prefix}-{name}-{size}.ptz"
        save_compressed(torch.as_tensor(t), path)
        paths.append(path)
    return paths


def glob_file_triples(directory, pattern="*-canonical-*.ptz"):     # game_runner.py:280-289
    triples = []
    for c_path in sorted(glob.glob(os.path.join(directory, pattern))):
        size = int(c_path.rsplit("-", 1)[-1].split(".")[0])
        triples.append((c_path, c_path.replace("-canonical-", "-v-"), c_path.replace("-canonical-", "-pi-"), size))
    return triples
