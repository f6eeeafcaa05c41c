"""Built-in FLAC writer of the CLI (mbexwn_vocoder_amd/flac.py): the stream is parsed back field by field as the format
document lays it out -- magic, STREAMINFO (block sizes, frame sizes, rate / channels / bits, sample count, MD5), every
frame's sync code, CRC-8 and CRC-16, and the verbatim samples."""
import hashlib
import struct

import numpy as np
import pytest

from mbexwn_vocoder_amd import flac


def _parse(stream):
    assert stream[:4] == b"fLaC"
    assert stream[4] == 0x80 and int.from_bytes(stream[5:8], "big") == 34          # last metadata block, STREAMINFO
    info = stream[8:42]
    min_block, max_block = struct.unpack(">HH", info[:4])
    min_frame, max_frame = int.from_bytes(info[4:7], "big"), int.from_bytes(info[7:10], "big")
    packed = int.from_bytes(info[10:18], "big")
    rate, channels, bits, total = packed >> 44, ((packed >> 41) & 7) + 1, ((packed >> 36) & 31) + 1, packed & ((1 << 36) - 1)
    md5 = info[18:34]
    pos, frames, index = 42, [], 0
    sizes = []
    while pos < len(stream):
        start = pos
        assert stream[pos] == 0xFF and stream[pos + 1] == 0xF8                      # sync, fixed block size
        size_code, rate_code = stream[pos + 2] >> 4, stream[pos + 2] & 15
        assert stream[pos + 3] >> 4 == channels - 1 and (stream[pos + 3] >> 1) & 7 == 4 and stream[pos + 3] & 1 == 0
        pos += 4
        first = stream[pos]                                                         # "UTF-8" coded frame number
        extra = 0 if first < 0x80 else (1 if first < 0xE0 else (2 if first < 0xF0 else 3))
        number = first if extra == 0 else first & (0x3F >> extra)
        for kk in range(extra):
            number = (number << 6) | (stream[pos + 1 + kk] & 0x3F)
        assert number == index
        pos += 1 + extra
        if size_code == 7:
            size = struct.unpack(">H", stream[pos:pos + 2])[0] + 1
            pos += 2
        else:
            assert size_code == 12
            size = 4096
        assert flac.crc8(stream[start:pos]) == stream[pos]
        pos += 1
        block = np.zeros((size, channels), dtype=np.int16)
        for ch in range(channels):
            assert stream[pos] == 0x02                                              # verbatim sub-frame, no wasted bits
            block[:, ch] = np.frombuffer(stream[pos + 1:pos + 1 + 2 * size], dtype=">i2")
            pos += 1 + 2 * size
        assert flac.crc16(stream[start:pos]) == struct.unpack(">H", stream[pos:pos + 2])[0]
        pos += 2
        sizes.append(pos - start)
        frames.append(block)
        index += 1
    pcm = np.concatenate(frames)
    assert (min_block, max_block) == (4096, 4096) and (min_frame, max_frame) == (min(sizes), max(sizes))
    assert total == pcm.shape[0] and bits == 16
    assert hashlib.md5(pcm.astype("<i2").tobytes()).digest() == md5
    return rate, pcm, rate_code


def test_crc_known_answers():
    assert flac.crc8(b"123456789") == 0xF4                    # CRC-8 (poly 0x07)
    assert flac.crc16(b"123456789") == 0xFEE8                 # CRC-16/BUYPASS (poly 0x8005, init 0)


def test_mono_stream_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    audio = np.clip(0.4 * rng.normal(size=24000 * 2 + 123), -1.2, 1.2).astype(np.float32)     # 12 frames, a short last one
    path = flac.write(str(tmp_path / "a.flac"), audio, 24000)
    rate, pcm, rate_code = _parse(open(path, "rb").read())
    assert rate == 24000 and rate_code == 7 and pcm.shape == (audio.size, 1)
    want = np.clip(np.rint(audio.astype(np.float64) * 32767), -32768, 32767).astype(np.int16)   # libsndfile's scale
    assert np.array_equal(pcm[:, 0], want)


def test_stereo_unlisted_rate_and_many_frames():
    rng = np.random.default_rng(1)
    pcm_in = rng.integers(-32768, 32768, size=(4096 * 130 + 1, 2)).astype(np.int16)          # frame numbers above 127
    rate, pcm, rate_code = _parse(flac.encode(pcm_in, 12345))
    assert rate == 12345 and rate_code == 0 and np.array_equal(pcm, pcm_in)


def test_other_integer_types_are_refused_and_soundfile_decodes_when_installed(tmp_path):
    """int32 (or any integer but int16) has no agreed scale: refused instead of being clipped to full scale.  Where a real
    decoder is installed (soundfile = libsndfile), it must read back what was written."""
    from mbexwn_vocoder_amd import flac
    with pytest.raises(TypeError):
        flac.to_pcm16(np.arange(10, dtype=np.int32))
    sf = pytest.importorskip("soundfile")
    rng = np.random.default_rng(3)
    audio = rng.uniform(-0.9, 0.9, size=5000).astype(np.float32)
    path = flac.write(str(tmp_path / "x.flac"), audio, 24000)
    data, rate = sf.read(path, dtype="int16")
    assert rate == 24000 and np.array_equal(data, flac.to_pcm16(audio))
