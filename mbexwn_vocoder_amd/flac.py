"""Minimal FLAC writer (mono / multi-channel 16-bit PCM, VERBATIM sub-frames) for the CLI's default output format.

The reference CLI writes ``--format flac`` through its ``sndio`` module on top of libsndfile (reference
bin/resynth_mel.py:104-105), which is not part of this image.  A FLAC stream does not have to be compressed: a
VERBATIM sub-frame stores the samples as they are, and with 16-bit samples every field behind the frame header is byte
aligned, so a valid stream (magic, STREAMINFO with the MD5 of the samples, frames with CRC-8 / CRC-16) can be assembled
with numpy alone.  Any FLAC decoder reads the result; the files are the size of a wav.

Format: https://xiph.org/flac/format.html (STREAMINFO, FRAME_HEADER, SUBFRAME_VERBATIM, FRAME_FOOTER).
"""
import hashlib
import struct

import numpy as np

BLOCK = 4096                                # samples per channel of every frame but the last
_RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}


def _crc_table(poly, bits):
    table = []
    top = 1 << (bits - 1)
    mask = (1 << bits) - 1
    for byte in range(256):
        crc = byte << (bits - 8)
        for _ in range(8):
            crc = ((crc << 1) ^ poly) & mask if crc & top else (crc << 1) & mask
        table.append(crc)
    return table


_CRC8 = _crc_table(0x07, 8)
_CRC16 = _crc_table(0x8005, 16)


def crc8(data):
    crc = 0
    for byte in data:
        crc = _CRC8[crc ^ byte]
    return crc


def crc16(data):
    crc = 0
    for byte in data:
        crc = ((crc << 8) & 0xFFFF) ^ _CRC16[(crc >> 8) ^ byte]
    return crc


def _utf8_number(value):
    """The "UTF-8" coding of a frame number (up to 31 bits)."""
    if value < 0x80:
        return bytes([value])
    out = []
    lead_bits = 6
    while value >= (1 << lead_bits):
        out.append(0x80 | (value & 0x3F))
        value >>= 6
        lead_bits -= 1
    lead = (0xFF << (lead_bits + 1)) & 0xFF
    out.append(lead | value)
    return bytes(reversed(out))


def to_pcm16(data):
    """float audio in [-1, 1] -> int16, scaled by 0x7FFF as libsndfile scales normalised floats for a 16-bit file (the
    reference's sndio path, bin/resynth_mel.py:104-105), rounded and clipped; int16 passes through; any other integer type
    is refused (its scale is ambiguous: convert it yourself).  (frames,) or (frames, channels)."""
    data = np.asarray(data)
    if data.dtype == np.int16:
        return data
    if not np.issubdtype(data.dtype, np.floating):
        raise TypeError(f"FLAC writer: float or int16 samples expected, got {data.dtype}")
    return np.clip(np.rint(data.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)


def encode(data, rate):
    """bytes of a FLAC stream holding ``data`` (float or int16; (frames,) or (frames, channels <= 8)) at ``rate`` Hz."""
    pcm = to_pcm16(data)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    n, channels = pcm.shape
    if not 1 <= channels <= 8 or not 0 < rate < (1 << 20):
        raise ValueError("FLAC: 1..8 channels and a sample rate below 2^20 Hz")
    rate = int(rate)
    frames = []
    min_frame = max_frame = 0
    for index, start in enumerate(range(0, n, BLOCK)):
        block = pcm[start:start + BLOCK]
        size = block.shape[0]
        size_code = 12 if size == BLOCK else 7                       # 1100: 4096; 0111: 16-bit (blocksize - 1) follows
        rate_code = _RATE_CODES.get(rate, 0)                         # 0000: take the rate from STREAMINFO
        head = bytes([0xFF, 0xF8, (size_code << 4) | rate_code, ((channels - 1) << 4) | (4 << 1)])     # 100: 16 bits
        head += _utf8_number(index)
        if size_code == 7:
            head += struct.pack(">H", size - 1)
        head += bytes([crc8(head)])
        # one VERBATIM sub-frame per channel: 0 | 000001 | 0, then the samples big-endian
        body = b"".join(b"\x02" + block[:, ch].astype(">i2").tobytes() for ch in range(channels))
        frame = head + body
        frame += struct.pack(">H", crc16(frame))
        frames.append(frame)
        min_frame = len(frame) if min_frame == 0 else min(min_frame, len(frame))
        max_frame = max(max_frame, len(frame))
    md5 = hashlib.md5(pcm.astype("<i2").tobytes()).digest()
    info = struct.pack(">HH", BLOCK, BLOCK) + min_frame.to_bytes(3, "big") + max_frame.to_bytes(3, "big")
    packed = (rate << 44) | ((channels - 1) << 41) | ((16 - 1) << 36) | n          # 20 + 3 + 5 + 36 bits
    info += packed.to_bytes(8, "big") + md5
    assert len(info) == 34
    header = b"fLaC" + bytes([0x80]) + len(info).to_bytes(3, "big") + info         # last-block flag | STREAMINFO
    return header + b"".join(frames)


def write(path, data, rate):
    with open(path, "wb") as fo:
        fo.write(encode(data, rate))
    return path
