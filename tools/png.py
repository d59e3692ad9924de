"""Minimal PNG writer (8-bit RGB) for eyeballing rendered frames; no third-party imaging library in the image."""
import struct
import zlib

import numpy as np


def write_png(path: str, rgb: np.ndarray) -> None:
    a = (np.clip(rgb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8) if rgb.dtype != np.uint8 else rgb
    h, w, _ = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
