"""RIFF / WAVE reader for the facade (`LibreASR.transcribe(path)`): integer PCM of 8 / 16 / 24 / 32 bits and IEEE float32 / float64,
plain `fmt ` chunks and WAVE_FORMAT_EXTENSIBLE.  Returns the FIRST channel as float32 in [-1, 1) -- the reference keeps one channel
(`ChannelCut`, transforms.py:128-132) of what torchaudio.load hands out (integer PCM scaled by 2^(bits-1)) -- the sample rate and the
bit depth.  The container has no audio I/O library (no torchaudio / soundfile), like the FLAC reader beside it."""
import struct

import numpy as np


def decode(path):
    data = open(path, "rb").read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if size < 16:
                raise ValueError("short fmt chunk")
            tag, nch, sr, _, align, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and size >= 40:                       # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the tag
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, nch, sr, align, bits)
        elif cid == b"data":
            # written to a pipe (ffmpeg / sox): the sizes could not be patched -- data size 0xFFFFFFFF, or 0 together with a RIFF
            # size of 0 / 0xFFFFFFFF: "to the end of the file".  A size of 0 in a file whose RIFF size is real is an EMPTY data chunk
            # (followed by LIST / id3 chunks that are not samples): honoured as declared
            riff = struct.unpack("<I", data[4:8])[0]
            if size == 0xFFFFFFFF or (size == 0 and riff in (0, 0xFFFFFFFF)):
                body, size = data[pos + 8:], len(data) - pos - 8
            pcm = body
        pos += 8 + size + (size & 1)                                # chunks are word-aligned
    if fmt is None or pcm is None:
        raise ValueError("WAVE file without fmt / data chunk")
    tag, nch, sr, align, bits = fmt
    if bits not in (8, 16, 24, 32, 64):
        raise ValueError(f"unsupported PCM width {bits}")
    if nch < 1 or align == 0 or align != nch * bits // 8:
        raise ValueError("inconsistent WAVE header")
    n = len(pcm) // align
    if n == 0:
        raise ValueError("WAVE file without samples")
    raw = np.frombuffer(pcm, dtype=np.uint8, count=n * align).reshape(n, nch, bits // 8)[:, 0, :]      # first channel
    if tag == 1:                                                    # integer PCM, little-endian; 8-bit is unsigned
        if bits == 8:
            x = (raw[:, 0].astype(np.float32) - 128.0) / 128.0
        elif bits in (16, 24, 32):
            v = np.zeros(n, np.int64)
            for b in range(bits // 8):
                v |= raw[:, b].astype(np.int64) << (8 * b)
            v -= (v >> (bits - 1)) << bits                          # sign
            x = (v.astype(np.float64) / float(1 << (bits - 1))).astype(np.float32)
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif tag == 3 and bits in (32, 64):                             # IEEE float
        x = np.ascontiguousarray(raw).view("<f4" if bits == 32 else "<f8").reshape(n).astype(np.float32)
    else:
        raise ValueError(f"unsupported WAVE format tag {tag} / {bits} bits")
    return np.ascontiguousarray(x), int(sr), int(bits)
