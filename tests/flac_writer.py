"""A small FLAC *encoder* written from the format specification (RFC 9639), test infrastructure only:
no FLAC codec exists in this image, so the native decoder (csrc/audio_io.cpp) is exercised against streams
produced here -- every subframe type, both Rice codings, escape partitions, wasted bits, the three stereo
decorrelations, odd final blocks -- plus the STREAMINFO MD5 (hashlib) as the end-to-end check."""
import hashlib

import numpy as np


class BitWriter(object):
    def __init__(self):
        self.acc = 0
        self.nbits = 0
        self.out = bytearray()

    def put(self, value, bits):
        if bits == 0:
            return
        value &= (1 << bits) - 1
        self.acc = (self.acc << bits) | value
        self.nbits += bits
        while self.nbits >= 8:
            self.nbits -= 8
            self.out.append((self.acc >> self.nbits) & 0xFF)
        self.acc &= (1 << self.nbits) - 1

    def unary(self, q):
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.nbits:
            self.put(0, 8 - self.nbits)

    def bytes(self):
        assert self.nbits == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def _utf8_number(n):
    if n < 0x80:
        return bytes([n])
    tail = []
    lead_bits = 6
    while n >= (1 << lead_bits):
        tail.append(0x80 | (n & 0x3F))
        n >>= 6
        lead_bits -= 1
    lead = ((0xFF << (lead_bits + 1)) & 0xFF) | n
    return bytes([lead] + tail[::-1])


def _write_residual(bw, res, order, blocksize, porder, force_escape=False):
    res = [int(v) for v in res]
    # zig-zag fold
    folded = [(v << 1) if v >= 0 else ((-v) << 1) - 1 for v in res]
    parts = 1 << porder
    assert (blocksize >> porder) << porder == blocksize or porder == 0
    ks = []
    pos = 0
    for p in range(parts):
        count = (blocksize >> porder) - (order if p == 0 else 0)
        seg = folded[pos:pos + count]
        pos += count
        best_k, best_bits = 0, None
        for k in range(0, 31):
            bits = sum((u >> k) + 1 + k for u in seg)
            if best_bits is None or bits < best_bits:
                best_k, best_bits = k, bits
        ks.append(best_k)
    method = 1 if max(ks + [0]) > 14 else 0
    bw.put(method, 2)
    bw.put(porder, 4)
    pos = 0
    for p in range(parts):
        count = (blocksize >> porder) - (order if p == 0 else 0)
        seg_f = folded[pos:pos + count]
        seg_r = res[pos:pos + count]
        pos += count
        if force_escape and p == parts - 1:
            bw.put(15 if method == 0 else 31, 4 if method == 0 else 5)
            width = max([1] + [int(abs(v)).bit_length() + 1 for v in seg_r])
            bw.put(width, 5)
            for v in seg_r:
                bw.put(v, width)
            continue
        k = ks[p]
        bw.put(k, 4 if method == 0 else 5)
        for u in seg_f:
            bw.unary(u >> k)
            bw.put(u & ((1 << k) - 1), k)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _write_subframe(bw, x, bps, kind, porder=0, force_escape=False, lpc_order=8, lpc_precision=12):
    """x: python ints of one channel of one block; kind in constant / verbatim / fixedN / lpc / auto."""
    x = [int(v) for v in x]
    n = len(x)
    wasted = 0
    if any(x):
        while all((v >> wasted) & 1 == 0 for v in x):
            wasted += 1
    if wasted:
        x = [v >> wasted for v in x]
        bps -= wasted
    if kind == "auto":
        kind = "constant" if len(set(x)) == 1 else "fixed2"
    if kind == "constant" and len(set(x)) != 1:
        kind = "verbatim"
    bw.put(0, 1)
    if kind == "constant":
        code = 0
    elif kind == "verbatim":
        code = 1
    elif kind.startswith("fixed"):
        order = int(kind[5:])
        code = 8 + order
    else:
        order = lpc_order
        code = 31 + order
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        bw.put(x[0], bps)
        return
    if kind == "verbatim":
        for v in x:
            bw.put(v, bps)
        return
    if kind.startswith("fixed"):
        coef, shift = FIXED[order], 0
    else:
        # least-squares predictor on the float signal, quantised to lpc_precision bits
        a = np.asarray(x, np.float64)
        rows = np.stack([a[order - 1 - j:n - 1 - j] for j in range(order)], axis=1)
        sol = np.linalg.lstsq(rows, a[order:], rcond=None)[0] if n > 2 * order else np.zeros(order)
        peak = max(np.abs(sol).max(), 1e-9)
        shift = int(min(15, max(0, lpc_precision - 1 - int(np.ceil(np.log2(peak + 1e-12))) - 1)))
        coef = [int(np.clip(round(c * (1 << shift)), -(1 << (lpc_precision - 1)), (1 << (lpc_precision - 1)) - 1))
                for c in sol]
    for v in x[:order]:
        bw.put(v, bps)
    if not kind.startswith("fixed"):
        bw.put(lpc_precision - 1, 4)
        bw.put(shift, 5)
        for c in coef:
            bw.put(c, lpc_precision)
    res = []
    for i in range(order, n):
        acc = sum(coef[j] * x[i - 1 - j] for j in range(order))
        res.append(x[i] - (acc >> shift))
    _write_residual(bw, res, order, n, porder, force_escape)


def write_flac(path, samples, rate, bps, blocksize=1024, plan=None, with_md5=True):
    """samples: int array [n] or [n, channels].  plan: list (cycled over blocks) of dicts with keys
    kind ('auto', 'constant', 'verbatim', 'fixed0'..'fixed4', 'lpc'), stereo ('indep', 'left_side',
    'side_right', 'mid_side'), porder, escape, lpc_order."""
    samples = np.asarray(samples)
    if samples.ndim == 1:
        samples = samples[:, None]
    n, nch = samples.shape
    plan = plan or [{}]
    md5 = hashlib.md5()
    nbytes = (bps + 7) // 8
    raw = bytearray()
    for row in samples:
        for v in row:
            raw += int(v).to_bytes(nbytes, "little", signed=True)
    md5.update(bytes(raw))
    frames = bytearray()
    min_f, max_f = 1 << 24, 0
    fno = 0
    for start in range(0, n, blocksize):
        blk = samples[start:start + blocksize]
        bs = len(blk)
        opt = plan[fno % len(plan)]
        kind = opt.get("kind", "auto")
        stereo = opt.get("stereo", "indep") if nch == 2 else "indep"
        porder = opt.get("porder", 0)
        while porder and ((bs >> porder) << porder != bs or (bs >> porder) <= 32):
            porder -= 1
        bw = BitWriter()
        bw.put(0b11111111111110, 14)
        bw.put(0, 1)
        bw.put(0, 1)                                   # fixed block size stream: frame number coded
        std = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12,
               8192: 13, 16384: 14, 32768: 15}
        bs_code = std.get(bs, 6 if bs <= 256 else 7)
        bw.put(bs_code, 4)
        sr_codes = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9,
                    48000: 10, 96000: 11}
        sr_code = sr_codes.get(rate, 0) if opt.get("rate_in_frame", True) else 0
        bw.put(sr_code, 4)
        ch_code = {"indep": nch - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[stereo]
        bw.put(ch_code, 4)
        ss_code = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bps, 0) if opt.get("bits_in_frame", True) else 0
        bw.put(ss_code, 3)
        bw.put(0, 1)
        for byte in _utf8_number(fno):
            bw.put(byte, 8)
        if bs_code == 6:
            bw.put(bs - 1, 8)
        elif bs_code == 7:
            bw.put(bs - 1, 16)
        header = bw.bytes()
        bw.put(crc8(header), 8)
        chans = [[int(v) for v in blk[:, c]] for c in range(nch)]
        widths = [bps] * nch
        if stereo == "left_side":
            chans = [chans[0], [l - r for l, r in zip(chans[0], chans[1])]]
            widths = [bps, bps + 1]
        elif stereo == "side_right":
            chans = [[l - r for l, r in zip(chans[0], chans[1])], chans[1]]
            widths = [bps + 1, bps]
        elif stereo == "mid_side":
            chans = [[(l + r) >> 1 for l, r in zip(chans[0], chans[1])], [l - r for l, r in zip(chans[0], chans[1])]]
            widths = [bps, bps + 1]
        for c in range(nch):
            _write_subframe(bw, chans[c], widths[c], kind, porder, opt.get("escape", False),
                            opt.get("lpc_order", 8))
        bw.align()
        body = bw.bytes()
        frame = body + crc16(body).to_bytes(2, "big")
        min_f, max_f = min(min_f, len(frame)), max(max_f, len(frame))
        frames += frame
        fno += 1
    info = bytearray(34)
    info[0:2] = blocksize.to_bytes(2, "big")
    info[2:4] = blocksize.to_bytes(2, "big")
    info[4:7] = (min_f if fno else 0).to_bytes(3, "big")
    info[7:10] = max_f.to_bytes(3, "big")
    info[10:18] = ((rate << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | n).to_bytes(8, "big")
    if with_md5:
        info[18:34] = md5.digest()
    with open(path, "wb") as fh:
        # a PADDING block first exercises metadata skipping; STREAMINFO must still be first per the spec,
        # so: STREAMINFO, then PADDING (last)
        fh.write(b"fLaC" + bytes([0x00]) + (34).to_bytes(3, "big") + bytes(info))
        fh.write(bytes([0x81]) + (16).to_bytes(3, "big") + bytes(16))
        fh.write(bytes(frames))
