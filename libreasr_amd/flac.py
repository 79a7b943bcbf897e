"""
Minimal FLAC decoder (pure Python + numpy) for BASELINE config 1 (demo/3729-6852-0035.flac):
no torchaudio / soundfile / sox / ffmpeg / flac binary exists in this image.

Supports what a 16 kHz 16-bit mono/stereo speech file uses: STREAMINFO, fixed-blocksize frames,
CONSTANT / VERBATIM / FIXED / LPC subframes, Rice partitions (both coding methods), mid/side /
left-side / right-side decorrelation.  decode() returns float32 PCM of channel 0 scaled to [-1, 1)
(torchaudio.load semantics: int / 2^(bps-1)), the sample rate and whether the MD5 of the decoded
PCM matches the STREAMINFO signature (the file's self-check: 93b7bac1... for the demo).
"""
import hashlib
import struct

import numpy as np


class _Bits:
    __slots__ = ("d", "p", "n")

    def __init__(self, data, pos=0):
        self.d, self.p, self.n = data, pos * 8, len(data) * 8

    def read(self, k):
        if k == 0:
            return 0
        p = self.p
        b0, b1 = p >> 3, (p + k + 7) >> 3
        v = int.from_bytes(self.d[b0:b1], "big")
        v >>= (b1 << 3) - (p + k)
        self.p = p + k
        return v & ((1 << k) - 1)

    def read_signed(self, k):
        v = self.read(k)
        return v - (1 << k) if v >> (k - 1) else v

    def unary(self):
        # number of 0 bits before the next 1 bit
        d, p = self.d, self.p
        cnt = 0
        b = p >> 3
        cur = d[b] & (0xFF >> (p & 7))
        if cur:
            lead = 8 - cur.bit_length() - (p & 7)
            self.p = p + lead + 1
            return lead
        cnt = 8 - (p & 7)
        b += 1
        while d[b] == 0:
            cnt += 8
            b += 1
        lead = 8 - d[b].bit_length()
        cnt += lead
        self.p = (b << 3) + lead + 1
        return cnt

    def align(self):
        self.p = (self.p + 7) & ~7


def _utf8(br):
    x = br.read(8)
    n = 0
    while x & (0x80 >> n):
        n += 1
    if n == 0:
        return x
    v = x & (0x7F >> n)
    for _ in range(n - 1):
        v = (v << 6) | (br.read(8) & 0x3F)
    return v


def _residual(br, order, blocksize, out):
    method = br.read(2)
    pbits = 4 if method == 0 else 5
    esc = (1 << pbits) - 1
    porder = br.read(4)
    nparts = 1 << porder
    i = order
    for part in range(nparts):
        cnt = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.read(pbits)
        if k == esc:
            nb = br.read(5)
            for _ in range(cnt):
                out[i] = br.read_signed(nb) if nb else 0
                i += 1
        else:
            rd, un = br.read, br.unary
            for _ in range(cnt):
                q = un()
                v = (q << k) | rd(k) if k else q
                out[i] = (v >> 1) ^ -(v & 1)
                i += 1


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(br, blocksize, bps):
    if br.read(1):
        raise ValueError("bad subframe padding")
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
    out = [0] * blocksize
    if typ == 0:
        out = [br.read_signed(bps)] * blocksize
    elif typ == 1:
        out = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= typ <= 12:
        order = typ - 8
        for i in range(order):
            out[i] = br.read_signed(bps)
        _residual(br, order, blocksize, out)
        co = _FIXED[order]
        for i in range(order, blocksize):
            s = out[i]
            for j, c in enumerate(co):
                s += c * out[i - 1 - j]
            out[i] = s
    elif typ >= 32:
        order = (typ & 31) + 1
        for i in range(order):
            out[i] = br.read_signed(bps)
        prec = br.read(4) + 1
        shift = br.read_signed(5)
        co = [br.read_signed(prec) for _ in range(order)]
        _residual(br, order, blocksize, out)
        for i in range(order, blocksize):
            s = 0
            for j in range(order):
                s += co[j] * out[i - 1 - j]
            out[i] += s >> shift
    else:
        raise ValueError(f"reserved subframe type {typ}")
    if wasted:
        out = [v << wasted for v in out]
    return out


def decode(path):
    data = open(path, "rb").read()
    if data[:4] != b"fLaC":
        raise ValueError("not a FLAC file")
    pos = 4
    info = None
    while True:
        hdr = data[pos]
        ln = int.from_bytes(data[pos + 1:pos + 4], "big")
        if hdr & 0x7F == 0:
            info = data[pos + 4:pos + 4 + ln]
        pos += 4 + ln
        if hdr & 0x80:
            break
    x = int.from_bytes(info[10:18], "big")
    sr, nch, bps, total = x >> 44, ((x >> 41) & 7) + 1, ((x >> 36) & 31) + 1, x & ((1 << 36) - 1)
    min_bs = struct.unpack(">H", info[0:2])[0]
    md5_ref = info[18:34]
    chans = [[] for _ in range(nch)]
    br = _Bits(data, pos)
    done = 0
    BS = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608}
    while done < total:
        if br.read(14) != 0x3FFE:
            raise ValueError("lost frame sync")
        br.read(1)
        br.read(1)
        bs_code, sr_code = br.read(4), br.read(4)
        ch_code, bps_code = br.read(4), br.read(3)
        br.read(1)
        _utf8(br)
        if bs_code == 6:
            bs = br.read(8) + 1
        elif bs_code == 7:
            bs = br.read(16) + 1
        elif bs_code >= 8:
            bs = 256 << (bs_code - 8)
        else:
            bs = BS.get(bs_code, min_bs)
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        br.read(8)  # header CRC-8
        fbps = {0: bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24}[bps_code]
        if ch_code < 8:
            subs = [_subframe(br, bs, fbps) for _ in range(ch_code + 1)]
        else:
            a = _subframe(br, bs, fbps + (1 if ch_code == 9 else 0))
            b = _subframe(br, bs, fbps + (1 if ch_code in (8, 10) else 0))
            a, b = np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)
            if ch_code == 8:        # left / side
                subs = [a, a - b]
            elif ch_code == 9:      # side / right
                subs = [a + b, b]
            else:                   # mid / side
                mid = (a << 1) | (b & 1)
                subs = [(mid + b) >> 1, (mid - b) >> 1]
        br.align()
        br.read(16)  # frame CRC-16
        for c in range(nch):
            chans[c].append(np.asarray(subs[c], dtype=np.int64))
        done += bs
    pcm = np.stack([np.concatenate(c)[:total] for c in chans], axis=1)     # [N, C]
    byts = (bps + 7) // 8
    if byts == 2:
        raw = pcm.astype("<i2").tobytes()
    else:
        raw = b"".join(int(v).to_bytes(byts, "little", signed=True) for v in pcm.reshape(-1))
    md5_ok = hashlib.md5(raw).digest() == md5_ref
    return (pcm[:, 0].astype(np.float32) / np.float32(1 << (bps - 1))), int(sr), bool(md5_ok)
