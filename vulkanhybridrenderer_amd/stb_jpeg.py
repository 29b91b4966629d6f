"""JPEG -> RGBA8 with the results of stb_image 2.26, the decoder the reference's scene loader uses
(/root/reference/src/scene/scene_loader.cpp:277-290: stbi_load / stbi_load_from_memory with STBI_rgb_alpha;
the decoder itself is the vendored third-party dependencies/stb/stb_image.h, restated here from its published
algorithm, not copied).

Entropy decoding (baseline and progressive Huffman, ITU T.81) is the same in every conforming decoder.  What differs between
decoders, and is therefore followed step by step:
  * dequantised coefficients are kept as 16-bit integers (products wrap);
  * the inverse DCT is the "slow integer" LL&M factorisation with 12-bit constants: a column pass that keeps two extra bits
    (rounding constant 512, shift 10), then a row pass with rounding 65536 + (128 << 17), shift 17 and a clamp to 0..255;
  * chroma upsampling: 1x1 pass-through; 2x1 horizontal and 1x2 vertical (3 * near + far + 2) >> 2; 2x2 the separable
    form with the vertical sums 3 * near + far kept unrounded and one rounding at the end ((3 * a + b + 8) >> 4), image
    borders replicated; anything else nearest-neighbour.  Which chroma row is "near" follows the decoder's row stepping:
    output row j takes chroma row j >> 1 as near and row (j >> 1) + (j & 1 ? 1 : -1), clamped to the component's rows, as far;
  * YCbCr -> RGB in 20-bit fixed point with constants rounded to 12 bits and shifted left by 8, the Cb contribution to green
    masked to its upper 16 bits (the scalar path mirrors what the SIMD path can compute), + 0.5 rounding, clamp;
  * a three-component file without a JFIF marker whose Adobe marker says "transform 0", or whose component ids are 'R', 'G',
    'B', is taken as RGB.
tests/test_reference_pins.py compares the output with stb_image itself on committed files (texel for texel).
"""
import numpy as np


class JpegError(ValueError):
    pass


_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                    54, 47, 55, 62, 63], np.int64)


def _f2f(x):
    return int(x * 4096 + 0.5)


_C = {k: _f2f(v) for k, v in dict(a=0.5411961, b=-1.847759065, c=0.765366865, d=1.175875602, e=0.298631336, f=2.053119869,
                                  g=3.072711026, h=1.501321110, i=-0.899976223, j=-2.562915447, k=-1.961570560, l=-0.390180644).items()}


def _idct_1d(s):
    """One pass of the LL&M inverse DCT on s[0..7] (int64 arrays of equal shape): -> (x0..x3, t0..t3); the caller adds its
    rounding constant to the x terms and forms x +- t."""
    p2, p3 = s[2], s[6]
    p1 = (p2 + p3) * _C["a"]
    t2 = p1 + p3 * _C["b"]
    t3 = p1 + p2 * _C["c"]
    p2, p3 = s[0], s[4]
    t0 = (p2 + p3) * 4096
    t1 = (p2 - p3) * 4096
    x0, x3, x1, x2 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = s[7], s[5], s[3], s[1]
    p3, p4, p1, p2 = t0 + t2, t1 + t3, t0 + t3, t1 + t2
    p5 = (p3 + p4) * _C["d"]
    t0, t1, t2, t3 = t0 * _C["e"], t1 * _C["f"], t2 * _C["g"], t3 * _C["h"]
    p1 = p5 + p1 * _C["i"]
    p2 = p5 + p2 * _C["j"]
    p3 = p3 * _C["k"]
    p4 = p4 * _C["l"]
    t3 = t3 + p1 + p4
    t2 = t2 + p2 + p3
    t1 = t1 + p2 + p4
    t0 = t0 + p1 + p3
    return (x0, x1, x2, x3), (t0, t1, t2, t3)


def _wrap32(a):
    return ((a + (1 << 31)) % (1 << 32)) - (1 << 31)          # C int arithmetic (two's complement wrap)


def idct_blocks(coef):
    """coef [n, 8, 8] int16 (row-major natural order) -> [n, 8, 8] uint8."""
    d = coef.astype(np.int64)
    (x0, x1, x2, x3), (t0, t1, t2, t3) = _idct_1d([d[:, r, :] for r in range(8)])            # columns: s_k = row k
    x0, x1, x2, x3 = x0 + 512, x1 + 512, x2 + 512, x3 + 512
    v = np.stack([_wrap32(x0 + t3) >> 10, _wrap32(x1 + t2) >> 10, _wrap32(x2 + t1) >> 10, _wrap32(x3 + t0) >> 10,
                  _wrap32(x3 - t0) >> 10, _wrap32(x2 - t1) >> 10, _wrap32(x1 - t2) >> 10, _wrap32(x0 - t3) >> 10], 1)
    (x0, x1, x2, x3), (t0, t1, t2, t3) = _idct_1d([v[:, :, c] for c in range(8)])            # rows: s_k = column k
    bias = 65536 + (128 << 17)
    x0, x1, x2, x3 = x0 + bias, x1 + bias, x2 + bias, x3 + bias
    o = np.stack([_wrap32(x0 + t3) >> 17, _wrap32(x1 + t2) >> 17, _wrap32(x2 + t1) >> 17, _wrap32(x3 + t0) >> 17,
                  _wrap32(x3 - t0) >> 17, _wrap32(x2 - t1) >> 17, _wrap32(x1 - t2) >> 17, _wrap32(x0 - t3) >> 17], 2)
    return np.clip(o, 0, 255).astype(np.uint8)


class _Huffman:
    """Canonical Huffman table: code lengths 1..16 -> symbols; decode by length with the max-code table."""

    def __init__(self, counts, symbols):
        self.symbols = symbols
        self.maxcode, self.delta = [0] * 18, [0] * 17
        code = k = 0
        for length in range(1, 17):
            self.delta[length] = k - code
            code += counts[length - 1]
            k += counts[length - 1]
            self.maxcode[length] = code          # first code NOT of this length
            code <<= 1
        self.maxcode[17] = 1 << 30
        # 9-bit acceleration table: (symbol, length) for codes of up to 9 bits
        self.fast = [None] * 512
        code = k = 0
        for length in range(1, 10):
            for _ in range(counts[length - 1]):
                base = code << (9 - length)
                for j in range(1 << (9 - length)):
                    self.fast[base + j] = (symbols[k], length)
                code += 1
                k += 1
            code <<= 1


class _Bits:
    """MSB-first bit reader over entropy-coded data: 0xFF00 is a stuffed 0xFF, any other marker ends the data (zeros are
    fed from then on, like the decoder does once it has seen a marker)."""

    def __init__(self, data, pos):
        self.data, self.pos = data, pos
        self.acc = self.n = 0
        self.marker = None

    def fill(self, need):
        d = self.data
        while self.n < need:
            b = 0
            if self.marker is None and self.pos < len(d):
                b = d[self.pos]
                self.pos += 1
                if b == 0xFF:
                    c = d[self.pos] if self.pos < len(d) else 0xD9
                    while c == 0xFF and self.pos + 1 < len(d):       # fill bytes
                        self.pos += 1
                        c = d[self.pos]
                    self.pos += 1
                    if c != 0:
                        self.marker = c
                        b = 0
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFFFF
            self.n += 8

    def get(self, n):
        if n == 0:
            return 0
        if self.n < n:
            self.fill(n)
        self.n -= n
        return (self.acc >> self.n) & ((1 << n) - 1)

    def bit(self):
        return self.get(1)

    def decode(self, h):
        if self.n < 16:
            self.fill(16)
        look = (self.acc >> (self.n - 9)) & 511
        f = h.fast[look]
        if f is not None:
            self.n -= f[1]
            return f[0]
        code = (self.acc >> (self.n - 16)) & 0xFFFF
        for length in range(10, 17):
            if (code >> (16 - length)) < h.maxcode[length]:
                self.n -= length
                idx = (code >> (16 - length)) + h.delta[length]
                if not 0 <= idx < len(h.symbols):
                    raise JpegError("bad huffman code")
                return h.symbols[idx]
        raise JpegError("bad huffman code")

    def extend_receive(self, n):
        v = self.get(n)
        return v if v >= (1 << (n - 1)) else v - (1 << n) + 1

    def reset(self):
        self.acc = self.n = 0
        self.marker = None


def _i16(v):
    return ((v + 32768) & 0xFFFF) - 32768


class _Component:
    pass


def _decode(data):
    if data[:2] != b"\xff\xd8":
        raise JpegError("not a JPEG")
    pos = 2
    qt = {}
    hdc, hac = {}, {}
    comps, progressive = None, False
    width = height = 0
    restart_interval = 0
    jfif, adobe_transform = False, -1
    eob_run = 0
    n = len(data)

    def u16(p):
        return (data[p] << 8) | data[p + 1]

    while True:
        while pos < n and data[pos] != 0xFF:
            pos += 1
        while pos < n and data[pos] == 0xFF:
            pos += 1
        if pos >= n:
            raise JpegError("no EOI")
        m = data[pos]
        pos += 1
        if m == 0xD9:
            break
        if m == 0xD8 or 0xD0 <= m <= 0xD7 or m == 0x01:
            continue
        length = u16(pos)
        seg, end = pos + 2, pos + length
        if m == 0xDB:                                                   # DQT
            p = seg
            while p < end:
                pq, tq = data[p] >> 4, data[p] & 15
                p += 1
                tab = np.zeros(64, np.int64)
                for i in range(64):
                    if pq:
                        tab[_ZIGZAG[i]] = u16(p); p += 2
                    else:
                        tab[_ZIGZAG[i]] = data[p]; p += 1
                qt[tq] = tab
        elif m == 0xC4:                                                 # DHT
            p = seg
            while p < end:
                tc, th = data[p] >> 4, data[p] & 15
                counts = list(data[p + 1: p + 17])
                total = sum(counts)
                symbols = list(data[p + 17: p + 17 + total])
                p += 17 + total
                (hac if tc else hdc)[th] = _Huffman(counts, symbols)
        elif m in (0xC0, 0xC1, 0xC2):                                   # SOF0/1/2
            progressive = m == 0xC2
            if data[seg] != 8:
                raise JpegError("only 8-bit JPEG")
            height, width = u16(seg + 1), u16(seg + 3)
            nc = data[seg + 5]
            if nc not in (1, 3):
                raise JpegError(f"{nc}-component JPEG (CMYK / YCCK) is not supported")
            comps = []
            for i in range(nc):
                c = _Component()
                c.id, hv, c.tq = data[seg + 6 + 3 * i], data[seg + 7 + 3 * i], data[seg + 8 + 3 * i]
                c.h, c.v = hv >> 4, hv & 15
                c.pred = 0
                comps.append(c)
            hmax, vmax = max(c.h for c in comps), max(c.v for c in comps)
            mcu_w, mcu_h = 8 * hmax, 8 * vmax
            mcux, mcuy = (width + mcu_w - 1) // mcu_w, (height + mcu_h - 1) // mcu_h
            for c in comps:
                c.x = (width * c.h + hmax - 1) // hmax
                c.y = (height * c.v + vmax - 1) // vmax
                c.bw, c.bh = mcux * c.h, mcuy * c.v                     # blocks, padded to whole MCUs
                c.coef = np.zeros((c.bh, c.bw, 64), np.int64)
        elif 0xC3 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise JpegError("arithmetic-coded / lossless / hierarchical JPEG is not supported")
        elif m == 0xDD:
            restart_interval = u16(seg)
        elif m == 0xE0:
            if length >= 7 and data[seg:seg + 5] == b"JFIF\0":
                jfif = True
        elif m == 0xEE:
            if length >= 14 and data[seg:seg + 6] == b"Adobe\0":
                adobe_transform = data[seg + 11]
        elif m == 0xDA:                                                 # SOS + entropy-coded data
            ns = data[seg]
            scan = []
            for i in range(ns):
                cid, tabs = data[seg + 1 + 2 * i], data[seg + 2 + 2 * i]
                c = next((c for c in comps if c.id == cid), None)
                if c is None:
                    raise JpegError("bad SOS component")
                c.td, c.ta = tabs >> 4, tabs & 15
                scan.append(c)
            ss, se, ah_al = data[seg + 1 + 2 * ns], data[seg + 2 + 2 * ns], data[seg + 3 + 2 * ns]
            ah, al = ah_al >> 4, ah_al & 15
            if not progressive:
                ss, se, ah, al = 0, 63, 0, 0
            bits = _Bits(data, end)
            for c in comps:
                c.pred = 0
            eob_run = 0

            def block_baseline(c, blk):
                t = bits.decode(hdc[c.td])
                diff = bits.extend_receive(t) if t else 0
                c.pred += diff
                q = qt[c.tq]
                blk[0] = _i16(c.pred * int(q[0]))
                k = 1
                h = hac[c.ta]
                while k < 64:
                    rs = bits.decode(h)
                    s, r = rs & 15, rs >> 4
                    if s == 0:
                        if rs != 0xF0:
                            break
                        k += 16
                    else:
                        k += r
                        z = int(_ZIGZAG[k])
                        blk[z] = _i16(bits.extend_receive(s) * int(q[z]))
                        k += 1

            def block_dc(c, blk):
                if ah == 0:
                    t = bits.decode(hdc[c.td])
                    diff = bits.extend_receive(t) if t else 0
                    c.pred += diff
                    blk[0] = _i16(c.pred * (1 << al))
                elif bits.bit():
                    blk[0] = _i16(int(blk[0]) + (1 << al))

            def block_ac(c, blk):
                nonlocal eob_run
                h = hac[c.ta]
                if ah == 0:
                    if eob_run:
                        eob_run -= 1
                        return
                    k = ss
                    while k <= se:
                        rs = bits.decode(h)
                        s, r = rs & 15, rs >> 4
                        if s == 0:
                            if r < 15:
                                eob_run = (1 << r)
                                if r:
                                    eob_run += bits.get(r)
                                eob_run -= 1
                                break
                            k += 16
                        else:
                            k += r
                            blk[int(_ZIGZAG[k])] = _i16(bits.extend_receive(s) * (1 << al))
                            k += 1
                    return
                bit = 1 << al                                            # refinement scan
                if eob_run:
                    eob_run -= 1
                    for k in range(ss, se + 1):
                        z = int(_ZIGZAG[k])
                        if blk[z] != 0 and bits.bit() and (int(blk[z]) & bit) == 0:
                            blk[z] = _i16(int(blk[z]) + (bit if blk[z] > 0 else -bit))
                    return
                k = ss
                while k <= se:
                    rs = bits.decode(h)
                    s, r = rs & 15, rs >> 4
                    if s == 0:
                        if r < 15:
                            eob_run = (1 << r) - 1
                            if r:
                                eob_run += bits.get(r)
                            r = 64                                       # run to the end of the band, refining as we go
                    else:
                        if s != 1:
                            raise JpegError("bad huffman code")
                        s = bit if bits.bit() else -bit
                    while k <= se:
                        z = int(_ZIGZAG[k])
                        k += 1
                        if blk[z] != 0:
                            if bits.bit() and (int(blk[z]) & bit) == 0:
                                blk[z] = _i16(int(blk[z]) + (bit if blk[z] > 0 else -bit))
                        else:
                            if r == 0:
                                blk[z] = _i16(s)
                                break
                            r -= 1

            def one(c, by, bx):
                blk = c.coef[by, bx]
                if not progressive:
                    block_baseline(c, blk)
                elif ss == 0:
                    block_dc(c, blk)
                else:
                    block_ac(c, blk)

            todo = restart_interval if restart_interval else 0x7FFFFFFF

            def restart():
                nonlocal eob_run, todo
                if bits.n < 24:
                    bits.fill(24)
                if bits.marker is not None and 0xD0 <= bits.marker <= 0xD7:
                    bits.reset()
                    for cc in comps:
                        cc.pred = 0
                    eob_run = 0
                    todo = restart_interval if restart_interval else 0x7FFFFFFF
                    return True
                return False

            done = False
            if ns == 1:
                c = scan[0]
                w, h = (c.x + 7) >> 3, (c.y + 7) >> 3                     # the component's own blocks, not the padded MCUs
                for by in range(h):
                    for bx in range(w):
                        one(c, by, bx)
                        todo -= 1
                        if todo <= 0 and not restart():
                            done = True
                            break
                    if done:
                        break
            else:
                for my in range(mcuy):
                    for mx in range(mcux):
                        for c in scan:
                            for y in range(c.v):
                                for x in range(c.h):
                                    one(c, my * c.v + y, mx * c.h + x)
                        todo -= 1
                        if todo <= 0 and not restart():
                            done = True
                            break
                    if done:
                        break
            # continue at the marker that ended the scan
            pos = bits.pos
            if bits.marker is not None:
                pos -= 2
            continue
        pos = end
    if comps is None:
        raise JpegError("no frame header")
    planes = []
    for c in comps:
        coef = c.coef
        if progressive:                                                 # dequantise at the end (16-bit products)
            coef = _i16(coef * qt[c.tq][None, None, :])
        px = idct_blocks(coef.reshape(-1, 8, 8).astype(np.int16)).reshape(c.bh, c.bw, 8, 8)
        planes.append(px.transpose(0, 2, 1, 3).reshape(c.bh * 8, c.bw * 8))
    rgb_ids = sum(1 for c, ch in zip(comps, b"RGB") if c.id == ch) if len(comps) == 3 else 0
    is_rgb = len(comps) == 3 and (rgb_ids == 3 or (adobe_transform == 0 and not jfif))
    return width, height, comps, planes, is_rgb


def _upsample(plane, comp, hs, vs, width, height):
    """One component to full resolution, row by row like the decoder's line stepping."""
    w_lo = (width + hs - 1) // hs
    src = plane[:, :w_lo].astype(np.int32)
    out = np.zeros((height, max(width, w_lo * hs)), np.uint8)
    ystep, ypos, l0, l1 = vs >> 1, 0, 0, 0
    for j in range(height):
        y_bot = ystep >= (vs >> 1)
        near, far = (src[l1], src[l0]) if y_bot else (src[l0], src[l1])
        if hs == 1 and vs == 1:
            row = near
        elif hs == 1 and vs == 2:
            row = (3 * near + far + 2) >> 2
        elif hs == 2 and vs == 1:
            row = np.empty(2 * w_lo, np.int32)
            if w_lo == 1:
                row[:] = near[0]
            else:
                row[0] = near[0]
                row[1] = (near[0] * 3 + near[1] + 2) >> 2
                mid = 3 * near[1:-1] + 2
                row[2:-2:2] = (mid + near[:-2]) >> 2
                row[3:-2:2] = (mid + near[2:]) >> 2
                row[-2] = (near[-2] * 3 + near[-1] + 2) >> 2
                row[-1] = near[-1]
        elif hs == 2 and vs == 2:
            t = 3 * near + far
            row = np.empty(2 * w_lo, np.int32)
            if w_lo == 1:
                row[:] = (t[0] + 2) >> 2
            else:
                row[0] = (t[0] + 2) >> 2
                row[1:-1:2] = (3 * t[:-1] + t[1:] + 8) >> 4
                row[2:-1:2] = (3 * t[1:] + t[:-1] + 8) >> 4
                row[-1] = (t[-1] + 2) >> 2
        else:
            row = np.repeat(near, hs)
        out[j, :len(row)] = row
        ystep += 1
        if ystep >= vs:
            ystep = 0
            l0 = l1
            ypos += 1
            if ypos < comp.y:
                l1 += 1
    return out[:, :width]


def _fx(x):
    return int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8


def decode_rgba(data):
    """Encoded JPEG bytes -> [h, w, 4] uint8, alpha 255 (stbi_load_from_memory(..., STBI_rgb_alpha))."""
    width, height, comps, planes, is_rgb = _decode(bytes(data))
    hmax, vmax = max(c.h for c in comps), max(c.v for c in comps)
    full = [_upsample(p, c, hmax // c.h, vmax // c.v, width, height) for p, c in zip(planes, comps)]
    out = np.empty((height, width, 4), np.uint8)
    out[..., 3] = 255
    if len(full) == 1:
        out[..., 0] = out[..., 1] = out[..., 2] = full[0]
    elif is_rgb:
        out[..., 0], out[..., 1], out[..., 2] = full
    else:
        y = (full[0].astype(np.int64) << 20) + (1 << 19)
        cb, cr = full[1].astype(np.int64) - 128, full[2].astype(np.int64) - 128
        r = y + cr * _fx(1.40200)
        g = y + cr * -_fx(0.71414) + (_wrap32(cb * -_fx(0.34414)) & ~0xFFFF)
        b = y + cb * _fx(1.77200)
        for i, ch in enumerate((r, g, b)):
            out[..., i] = np.clip(_wrap32(ch) >> 20, 0, 255)
    return out
