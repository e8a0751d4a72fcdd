"""Independent reader for the JPEG XL codestreams this encoder emits (TEST INFRASTRUCTURE).

There is no djxl in this image, so "every output is decodable" is checked by a decoder written
from the codestream format itself (ISO/IEC 18181-1: headers, TOC, Brotli-style prefix codes,
clustered context maps, hybrid-uint tokens, the modular sub-bitstream with its meta-adaptive
context tree for the DC image and the AC metadata, VarDCT coefficient tokens with the
zero-density context model, default dequantisation, chroma from luma, inverse DCTs of the three
transforms cjxl_tiny uses, inverse XYB).  It follows the *format*, not the encoder: contexts come
from the transmitted tree / block-context map / context maps, never from the encoder's tables;
the only constants shared with the encoder are the format's own defaults (dequantisation
weights, coefficient orders, the two zero-density LUTs), read from csrc/jxlt_tables.h.

Restrictions (= what cjxl_tiny can emit): one frame, VarDCT, XYB, one pass, prefix codes (no ANS,
no LZ77), no coefficient-order permutations, default quant matrices, transforms DCT8 / DCT16X8 /
DCT8X16, no patches/splines/noise, no gaborish; the edge-preserving filter is signalled but not
applied here (it only smooths), so the reconstruction is judged by PSNR, not bit-exactly.

Every section must be consumed to within its last byte of the TOC size, every token must fall
inside its alphabet, every block must be covered exactly once: a malformed stream raises
DecodeError.  Pure Python + numpy: meant for images up to about a megapixel.
"""
import math
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


class DecodeError(ValueError):
    pass


def _need(cond, what):
    if not cond:
        raise DecodeError(what)


# --------------------------------------------------------------------------- constants of the format
def _tables():
    text = (ROOT / "libjxl-tiny_amd" / "csrc" / "jxlt_tables.h").read_text()
    out = {}
    for name in ("kQuantWeightBits", "kCoeffOrder", "kCoeffFreqContext", "kCoeffNumNonzeroContext"):
        m = re.search(r"JXLT_%s\[\d+\]\s*=\s*\{(.*?)\};" % name, text, re.S)
        out[name] = [int(v.rstrip("u"), 0) for v in re.findall(r"0x[0-9a-fA-F]+u?|\d+", m.group(1))]
    w = np.array(out["kQuantWeightBits"], dtype=np.uint32).view(np.float32)
    return w, out["kCoeffOrder"], out["kCoeffFreqContext"], out["kCoeffNumNonzeroContext"]


_W, _ORDER, _FREQ_CTX, _NNZ_CTX = _tables()
_QUANT_BIAS = (1.0 - 0.05465007330715401, 1.0 - 0.07005449891748593, 1.0 - 0.049935103337343655, 0.145)
_DC_STEP = (1.0 / 4096, 1.0 / 512, 1.0 / 256)  # default DC dequantisation
# strategy codes of the format that this subset knows: code -> (name, covered x, covered y, order id)
_STRATEGIES = {0: ("DCT8", 1, 1, 0), 6: ("DCT16X8", 1, 2, 4), 7: ("DCT8X16", 2, 1, 4)}


# --------------------------------------------------------------------------- bit reader
class BitReader:
    def __init__(self, data, bit_pos=0):
        self.data = bytes(data)
        self.pos = bit_pos  # absolute bit position
        self.limit = 8 * len(self.data)
        self._big = int.from_bytes(self.data, "little") if len(self.data) <= (1 << 16) else None

    def peek(self, n):
        if self._big is not None:
            return (self._big >> self.pos) & ((1 << n) - 1)
        byte = self.pos >> 3
        chunk = int.from_bytes(self.data[byte:byte + 8], "little")
        return (chunk >> (self.pos & 7)) & ((1 << n) - 1)

    def skip(self, n):
        self.pos += n
        _need(self.pos <= self.limit, "read past the end of a section")

    def read(self, n):
        if n == 0:
            return 0
        v = self.peek(n)
        self.skip(n)
        return v

    def align(self):
        pad = (-self.pos) & 7
        _need(self.read(pad) == 0, "non-zero padding bits")

    def u32(self, *dists):
        """U32 of the format: 2 selector bits, then dists[sel] = int (Val) or (bits, offset)."""
        d = dists[self.read(2)]
        if isinstance(d, int):
            return d
        return self.read(d[0]) + d[1]


# --------------------------------------------------------------------------- prefix codes
class PrefixCode:
    def __init__(self, lengths):
        self.lengths = lengths
        nz = [(l, s) for s, l in enumerate(lengths) if l]
        if len(nz) <= 1:
            self.single = nz[0][1] if nz else 0
            self.table = None
            return
        self.single = None
        _need(sum(1 << (15 - l) for l, _ in nz) == 1 << 15, "prefix code is not complete")
        table_sym = np.zeros(1 << 15, dtype=np.int32)
        table_len = np.zeros(1 << 15, dtype=np.int32)
        code = 0
        prev_len = 0
        for l, s in sorted(nz):
            code <<= (l - prev_len)
            prev_len = l
            rev = int(format(code, "0%db" % l)[::-1], 2)  # stream carries the code MSB first
            table_sym[rev::1 << l] = s
            table_len[rev::1 << l] = l
            code += 1
        self.table = (table_sym.tolist(), table_len.tolist())

    def read(self, br):
        if self.table is None:
            return self.single
        idx = br.peek(15)
        br.skip(self.table[1][idx])
        return self.table[0][idx]


_CL_ORDER = (1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15)


def _read_prefix_code(br, alphabet_size):
    """RFC 7932 section 3.4 / 3.5 as used by 18181-1."""
    if alphabet_size == 1:
        return PrefixCode([0])
    hskip = br.read(2)
    if hskip == 1:  # simple code
        nsym = br.read(2) + 1
        bits = max(0, (alphabet_size - 1).bit_length())
        syms = [br.read(bits) for _ in range(nsym)]
        _need(all(s < alphabet_size for s in syms) and len(set(syms)) == nsym, "bad simple prefix code")
        lengths = [0] * alphabet_size
        if nsym == 1:
            return _single(syms[0], alphabet_size)
        if nsym == 2:
            ls = (1, 1)
        elif nsym == 3:
            ls = (1, 2, 2)
        else:
            ls = (1, 2, 3, 3) if br.read(1) else (2, 2, 2, 2)
        for s, l in zip(syms, ls):
            lengths[s] = l
        return PrefixCode(lengths)
    # complex code: lengths of the 18 code-length symbols, fixed variable-length code
    cl = [0] * 18
    space = 32
    nonzero = 0
    for i in range(hskip, 18):
        v = br.peek(4)
        if v & 3 == 0:
            l, n = 0, 2
        elif v & 3 == 2:
            l, n = 3, 2
        elif v & 3 == 1:
            l, n = 4, 2
        elif v & 7 == 3:
            l, n = 2, 3
        elif v == 7:
            l, n = 1, 4
        else:
            l, n = 5, 4
        br.skip(n)
        cl[_CL_ORDER[i]] = l
        if l:
            space -= 32 >> l
            nonzero += 1
            if space <= 0:
                break
    _need(nonzero == 1 or space == 0, "code-length code is not complete")
    if nonzero == 1:
        only = max(range(18), key=lambda s: cl[s])
        cl_code = _single(only, 18)
    else:
        cl_code = PrefixCode(cl)
    lengths = [0] * alphabet_size
    sym, prev, repeat, repeat_len, space = 0, 8, 0, 0, 32768
    while sym < alphabet_size and space > 0:
        c = cl_code.read(br)
        if c < 16:
            repeat = 0
            lengths[sym] = c
            sym += 1
            if c:
                prev = c
                space -= 32768 >> c
        else:
            extra = 2 if c == 16 else 3
            new_len = prev if c == 16 else 0
            if repeat_len != new_len:
                repeat, repeat_len = 0, new_len
            old = repeat
            if repeat > 0:
                repeat = (repeat - 2) << extra
            repeat += br.read(extra) + 3
            delta = repeat - old
            _need(sym + delta <= alphabet_size, "code-length run overflows the alphabet")
            for i in range(delta):
                lengths[sym + i] = repeat_len
            sym += delta
            if repeat_len:
                space -= delta << (15 - repeat_len)
    _need(space == 0, "prefix code lengths do not fill the code space")
    return PrefixCode(lengths)


def _single(symbol, alphabet_size):
    pc = PrefixCode([0] * alphabet_size)
    pc.single = symbol
    return pc


class HybridUint:
    def __init__(self, br, log_alpha=15):
        self.split_exp = br.read((log_alpha + 1 - 1).bit_length() if log_alpha else 0)
        self.msb = self.lsb = 0
        if self.split_exp != log_alpha:
            self.msb = br.read(self.split_exp.bit_length())
            _need(self.msb <= self.split_exp, "msb_in_token > split_exponent")
            self.lsb = br.read((self.split_exp - self.msb).bit_length())
            _need(self.lsb + self.msb <= self.split_exp, "lsb_in_token + msb_in_token > split_exponent")
        self.split = 1 << self.split_exp

    def value(self, token, br):
        if token < self.split:
            return token
        m, l = self.msb, self.lsb
        nbits = self.split_exp - (m + l) + ((token - self.split) >> (m + l))
        _need(nbits <= 29, "hybrid-uint token too large")
        low = token & ((1 << l) - 1)
        hi = ((token >> l) & ((1 << m) - 1)) | (1 << m)
        return (((hi << nbits) | br.read(nbits)) << l) | low


class EntropyStream:
    """A clustered set of prefix codes with hybrid-uint configurations (18181-1 C.2)."""

    def __init__(self, br, num_contexts):
        _need(br.read(1) == 0, "LZ77 is not part of the cjxl_tiny subset")
        self.context_map = _read_context_map(br, num_contexts) if num_contexts > 1 else [0]
        n = max(self.context_map) + 1
        _need(br.read(1) == 1, "ANS is not part of the cjxl_tiny subset (use_prefix_code must be 1)")
        self.uint = [HybridUint(br) for _ in range(n)]
        counts = []
        for _ in range(n):
            if br.read(1) == 0:
                counts.append(1)
            else:
                nb = br.read(4)
                counts.append((1 << nb) + br.read(nb) + 1)
        self.codes = [_read_prefix_code(br, c) for c in counts]
        self.counts = counts

    def read(self, br, ctx):
        k = self.context_map[ctx]
        return self.uint[k].value(self.codes[k].read(br), br)


def _read_context_map(br, num_contexts):
    if br.read(1):  # simple
        bits = br.read(2)
        cmap = [br.read(bits) for _ in range(num_contexts)]
    else:
        use_mtf = br.read(1)
        nested = EntropyStream(br, 1)
        cmap = [nested.read(br, 0) for _ in range(num_contexts)]
        if use_mtf:
            mtf = list(range(256))
            for i, v in enumerate(cmap):
                cmap[i] = mtf.pop(v)
                mtf.insert(0, cmap[i])
    _need(max(cmap) < 256, "context map entry out of range")
    return cmap


def _unpack_signed(u):
    return (u >> 1) ^ (-(u & 1))


# --------------------------------------------------------------------------- modular sub-bitstream
class MATree:
    """Meta-adaptive context tree (18181-1 H.4.2); nodes in the transmitted (breadth-first) order."""

    def __init__(self, br):
        es = EntropyStream(br, 6)
        self.nodes = []
        pending = 1
        leaves = 0
        while pending:
            pending -= 1
            prop = es.read(br, 1) - 1
            if prop >= 0:
                _need(prop < 16, "property %d is outside the subset" % prop)
                split = _unpack_signed(es.read(br, 0))
                self.nodes.append([prop, split, 0, 0])
                pending += 2
            else:
                predictor = es.read(br, 2)
                offset = _unpack_signed(es.read(br, 3))
                mul_log = es.read(br, 4)
                mul_bits = es.read(br, 5)
                self.nodes.append([-1, leaves, predictor, offset, (mul_bits + 1) << mul_log])
                leaves += 1
            _need(len(self.nodes) < (1 << 20), "tree too large")
        # children of the i-th decision node, in order of appearance
        nxt = 1
        for node in self.nodes:
            if node[0] >= 0:
                node[2], node[3] = nxt, nxt + 1  # left = property > splitval, right = otherwise
                nxt += 2
        self.num_leaves = leaves
        self.used = sorted({n[0] for n in self.nodes if n[0] >= 0})

    def leaf(self, props):
        node = self.nodes[0]
        while node[0] >= 0:
            node = self.nodes[node[2] if props[node[0]] > node[1] else node[3]]
        return node


def _decode_modular_channels(br, tree, es, stream_id, shapes):
    """Channels of one modular sub-bitstream that uses the global tree; no transforms."""
    _need(br.read(1) == 1, "modular group must use the global tree")
    _need(br.read(1) == 1, "non-default weighted-predictor parameters are outside the subset")
    _need(br.read(2) == 0, "modular transforms are outside the subset")
    _need(all(p in (0, 1, 2, 3, 4, 5, 6, 7, 9) for p in tree.used), "tree uses properties outside the subset: %s" % tree.used)
    out = []
    for ch, (h, w) in enumerate(shapes):
        img = [[0] * w for _ in range(h)]
        props = [0] * 16
        props[0], props[1] = ch, stream_id
        for y in range(h):
            row = img[y]
            top = img[y - 1] if y else None
            props[2] = y
            for x in range(w):
                west = row[x - 1] if x else (top[x] if y else 0)
                north = top[x] if y else west
                nw = top[x - 1] if (x and y) else west
                props[3] = x
                props[4], props[5], props[6], props[7] = abs(north), abs(west), north, west
                props[9] = west + north - nw
                leaf = tree.leaf(props)
                pred = leaf[2]
                if pred == 0:
                    guess = 0
                elif pred == 1:
                    guess = west
                elif pred == 2:
                    guess = north
                elif pred == 5:
                    lo, hi = (north, west) if north < west else (west, north)
                    guess = min(max(west + north - nw, lo), hi)
                else:
                    raise DecodeError("predictor %d is outside the subset" % pred)
                row[x] = _unpack_signed(es.read(br, leaf[1])) * leaf[4] + leaf[3] + guess
        out.append(img)
    return out


# --------------------------------------------------------------------------- inverse transforms
def _idct_matrix(n):
    k = np.arange(n)[:, None]
    x = np.arange(n)[None, :]
    m = np.cos((x + 0.5) * k * np.pi / n)
    m[1:] *= math.sqrt(2.0)
    return m  # pixels = coefficients(k) @ m


_I8, _I16 = _idct_matrix(8), _idct_matrix(16)
_LLF_SCALE = 0.901764195028874394  # DCT resample scale between the 2-point and the 16-point basis


def _block_pixels(name, coef):
    if name == "DCT8":
        c = coef.reshape(8, 8)          # [h][v]
        return _I8.T @ c.T @ _I8        # rows = y, columns = x
    if name == "DCT16X8":
        c = coef.reshape(8, 16)         # [h][v], 16 rows x 8 columns of pixels
        return _I16.T @ c.T @ _I8
    c = coef.reshape(8, 16)             # [v][h], 8 rows x 16 columns
    return _I8.T @ c @ _I16


# --------------------------------------------------------------------------- the decoder
class Decoded:
    pass


def decode(codestream):
    br = BitReader(codestream)
    out = Decoded()
    # ---- codestream + image headers
    _need(br.read(16) == 0x0AFF, "not a JPEG XL codestream")
    size_u32 = lambda: br.u32((9, 1), (13, 1), (18, 1), (30, 1))  # noqa: E731
    if br.read(1):
        ys = 8 * (br.read(5) + 1)
        ratio = br.read(3)
        _need(ratio == 0, "aspect-ratio shortcuts unsupported")
        xs = 8 * (br.read(5) + 1)
    else:
        ys = size_u32()
        ratio = br.read(3)
        _need(ratio == 0, "aspect-ratio shortcuts unsupported")
        xs = size_u32()
    out.xsize, out.ysize = xs, ys
    _need(br.read(1) == 0, "all-default image metadata means 8-bit sRGB, not this encoder")
    _need(br.read(1) == 0, "extra metadata fields unsupported")
    _need(br.read(1) == 1, "float samples expected")
    _need(br.u32(32, 16, 24, (6, 1)) == 32 and br.read(4) + 1 == 8, "binary32 samples expected")
    br.read(1)  # modular_16bit_buffers
    _need(br.u32(0, 1, (4, 2), (12, 1)) == 0, "extra channels unsupported")
    _need(br.read(1) == 1, "xyb_encoded expected")
    _need(br.read(1) == 0 and br.read(1) == 0, "colour encoding: explicit, no ICC expected")
    enum = lambda: br.u32(0, 1, (4, 2), (6, 18))  # noqa: E731
    _need(enum() == 0 and enum() == 1 and enum() == 1, "RGB / D65 / sRGB primaries expected")
    _need(br.read(1) == 0 and enum() == 8, "linear transfer function expected")
    enum()  # rendering intent
    _need(_u64(br) == 0, "extensions unsupported")
    _need(br.read(1) == 1, "default transform data expected")
    br.align()
    # ---- frame header
    _need(br.read(1) == 0, "all-default frame header is not VarDCT-with-flags")
    _need(br.read(2) == 0 and br.read(1) == 0, "regular VarDCT frame expected")
    flags = _u64(br)
    out.flags = flags
    _need(flags & ~0x80 == 0, "patches / splines / noise are outside the subset")
    _need(br.read(2) == 0, "upsampling unsupported")
    out.x_qm_scale, out.b_qm_scale = br.read(3), br.read(3)
    _need(br.u32(1, 2, 3, (3, 4)) == 1, "one pass expected")
    _need(br.read(1) == 0, "custom frame size unsupported")
    _need(br.u32(0, 1, 2, (2, 3)) == 0, "replace blend mode expected")
    _need(br.read(1) == 1, "is_last expected")
    _need(br.u32(0, (4, 0), (5, 16), (10, 48)) == 0, "frame name unsupported")
    if br.read(1):  # all-default loop filter: gaborish on, 2 EPF iterations
        out.gaborish, out.epf_iters = True, 2
    else:
        out.gaborish = bool(br.read(1))
        _need(not out.gaborish or br.read(1) == 0, "custom gaborish weights unsupported")
        out.epf_iters = br.read(2)
        if out.epf_iters:
            _need(br.read(1) == 0 and br.read(1) == 0 and br.read(1) == 0, "custom EPF parameters unsupported")
        _need(_u64(br) == 0, "loop-filter extensions unsupported")
    _need(_u64(br) == 0, "frame-header extensions unsupported")
    # ---- TOC
    xg, yg = -(-xs // 256), -(-ys // 256)
    xdg, ydg = -(-xs // 2048), -(-ys // 2048)
    num_groups, num_dc_groups = xg * yg, xdg * ydg
    num_sections = 1 if num_groups == 1 else 2 + num_dc_groups + num_groups
    _need(br.read(1) == 0, "permuted TOC unsupported")
    br.align()
    sizes = [br.u32((10, 0), (14, 1024), (22, 17408), (30, 4211712)) for _ in range(num_sections)]
    br.align()
    start = br.pos // 8
    _need(start + sum(sizes) == len(codestream), "TOC sizes (%d) do not add up to the codestream length (%d)"
          % (start + sum(sizes), len(codestream)))
    out.section_sizes = sizes
    sections = []
    for s in sizes:
        sections.append(codestream[start:start + s])
        start += s
    single = num_sections == 1
    rd = BitReader(sections[0])

    def next_section(index):
        """Sections are byte-aligned chunks, except in single-section frames (bit-concatenated)."""
        nonlocal rd
        if single:
            return rd
        _finish(rd)
        rd = BitReader(sections[index])
        return rd

    def _finish(r):
        _need(r.limit - r.pos < 8, "section has %d unread bits" % (r.limit - r.pos))
        _need(r.read(r.limit - r.pos) == 0, "non-zero padding at the end of a section")

    # ---- DC global: quantiser, block context map, global tree, DC code
    _need(rd.read(1) == 1, "default DC dequantisation expected")
    global_scale = rd.u32((11, 1), (11, 2049), (12, 4097), (16, 8193))
    quant_dc = rd.u32(16, (5, 1), (8, 1), (16, 1))
    out.global_scale, out.quant_dc = global_scale, quant_dc
    _need(rd.read(1) == 0, "default block context map is not what cjxl_tiny emits")
    _need(rd.read(16) == 0, "DC / quant-field thresholds of the block context map unsupported")
    bcm = _read_context_map(rd, 39)
    num_block_ctx = max(bcm) + 1
    _need(rd.read(1) == 1, "default chroma-from-luma DC parameters expected")
    _need(rd.read(1) == 1, "global tree expected")
    tree = MATree(rd)
    dc_code = EntropyStream(rd, tree.num_leaves)
    out.tree_leaves = tree.num_leaves

    xb, yb = -(-xs // 8), -(-ys // 8)
    xt, yt = -(-xs // 64), -(-ys // 64)
    qdc = np.zeros((3, yb, xb), dtype=np.int32)
    strat = np.full((yb, xb), -1, dtype=np.int32)     # strategy code at first blocks, -2 = covered by a neighbour
    rawq = np.zeros((yb, xb), dtype=np.int32)
    cfl = np.zeros((2, yt, xt), dtype=np.int32)
    # ---- DC groups
    for g in range(num_dc_groups):
        rd = next_section(1 + g)
        gx, gy = g % xdg, g // xdg
        bx0, by0 = gx * 256, gy * 256
        nbx, nby = min(256, xb - bx0), min(256, yb - by0)
        _need(rd.read(2) == 0, "extra DC precision unsupported")
        chans = _decode_modular_channels(rd, tree, dc_code, 1 + g, [(nby, nbx)] * 3)
        for i, c in enumerate((1, 0, 2)):
            qdc[c, by0:by0 + nby, bx0:bx0 + nbx] = np.array(chans[i], dtype=np.int32)
        # AC metadata
        nbits = (nbx * nby - 1).bit_length()
        nblocks = rd.read(nbits) + 1
        _need(nblocks <= nbx * nby, "more varblocks than blocks")
        ntx, nty = -(-nbx // 8), -(-nby // 8)
        meta = _decode_modular_channels(rd, tree, dc_code, 1 + 2 * num_dc_groups + g,
                                        [(nty, ntx), (nty, ntx), (2, nblocks), (nby, nbx)])
        cfl[0, gy * 32:gy * 32 + nty, gx * 32:gx * 32 + ntx] = np.array(meta[0], dtype=np.int32)
        cfl[1, gy * 32:gy * 32 + nty, gx * 32:gx * 32 + ntx] = np.array(meta[1], dtype=np.int32)
        _need(np.abs(cfl).max() <= 128, "chroma-from-luma factor out of range")
        i = 0
        for y in range(nby):
            for x in range(nbx):
                if strat[by0 + y, bx0 + x] != -1:
                    continue
                _need(i < nblocks, "block info runs out before the group is covered")
                code, q = meta[2][0][i], meta[2][1][i] + 1
                i += 1
                _need(code in _STRATEGIES, "transform %d is outside the subset" % code)
                _need(1 <= q <= 256, "quant field value out of range")
                _, cx, cy, _ = _STRATEGIES[code]
                _need(x + cx <= nbx and y + cy <= nby, "varblock crosses the group edge")
                _need((strat[by0 + y:by0 + y + cy, bx0 + x:bx0 + x + cx] == -1).all(), "varblocks overlap")
                strat[by0 + y:by0 + y + cy, bx0 + x:bx0 + x + cx] = -2
                strat[by0 + y, bx0 + x] = code
                rawq[by0 + y:by0 + y + cy, bx0 + x:bx0 + x + cx] = q
        _need(i == nblocks, "%d block-info entries unused" % (nblocks - i))
        _need(all(0 <= v < 8 for row in meta[3] for v in row), "EPF sharpness out of range")
    # ---- AC global
    rd = next_section(1 + num_dc_groups)
    _need(rd.read(1) == 1, "default dequantisation matrices expected")
    nh_bits = (num_groups - 1).bit_length()
    _need(rd.read(nh_bits) == 0, "one histogram set expected")
    _need(rd.u32(0x5F, 0x13, 0, (13, 0)) == 0, "coefficient-order permutations unsupported")
    ac_code = EntropyStream(rd, 495 * num_block_ctx)
    out.ac_clusters = max(ac_code.context_map) + 1

    # ---- AC groups -> coefficients -> pixels
    scale = global_scale / 65536.0
    x_mul = 1.25 ** (out.x_qm_scale - 2)
    b_mul = 1.25 ** (out.b_qm_scale - 2)
    dc_step = [s / (scale * quant_dc) for s in _DC_STEP]
    xyb = np.zeros((3, yb * 8, xb * 8), dtype=np.float64)
    nzgrid = np.zeros((3, yb, xb), dtype=np.int32)
    out.num_tokens = 0
    for g in range(num_groups):
        rd = next_section(2 + num_dc_groups + g)
        ggx, ggy = g % xg, g // xg
        gbx0, gby0 = ggx * 32, ggy * 32
        nbx, nby = min(32, xb - gbx0), min(32, yb - gby0)
        for by in range(gby0, gby0 + nby):
            for bx in range(gbx0, gbx0 + nbx):
                code = strat[by, bx]
                if code < 0:
                    _need(code == -2, "block not covered by any varblock")
                    continue
                name, cx, cy, order_id = _STRATEGIES[code]
                covered = cx * cy
                log2c = covered.bit_length() - 1
                size = 64 * covered
                order = _ORDER[0:64] if covered == 1 else _ORDER[64:192]
                wofs = (0, 64, 128) if covered == 1 else (192, 320, 448)
                q = int(rawq[by, bx])
                coefs = np.zeros((3, size))
                for c in (1, 0, 2):
                    bctx = bcm[(c ^ 1 if c < 2 else 2) * 13 + order_id]
                    # number of non-zeros, predicted from the blocks above / to the left in the group
                    if bx == gbx0:
                        pred = 32 if by == gby0 else int(nzgrid[c, by - 1, bx])
                    elif by == gby0:
                        pred = int(nzgrid[c, by, bx - 1])
                    else:
                        pred = (int(nzgrid[c, by - 1, bx]) + int(nzgrid[c, by, bx - 1]) + 1) // 2
                    bucket = pred if pred < 8 else (36 if pred >= 64 else 4 + pred // 2)
                    nz = ac_code.read(rd, bucket * num_block_ctx + bctx)
                    out.num_tokens += 1
                    _need(nz <= size - covered, "more non-zeros than coefficients")
                    nzgrid[c, by:by + cy, bx:bx + cx] = (nz + covered - 1) >> log2c
                    base = 37 * num_block_ctx + 458 * bctx
                    prev = 0 if nz > size // 16 else 1
                    k = covered
                    qc = [0] * size
                    while nz:
                        _need(k < size, "non-zeros left after the last coefficient")
                        ctx = base + (_NNZ_CTX[(nz + covered - 1) >> log2c] + _FREQ_CTX[k >> log2c]) * 2 + prev
                        v = _unpack_signed(ac_code.read(rd, ctx))
                        out.num_tokens += 1
                        qc[order[k]] = v
                        prev = 1 if v else 0
                        nz -= prev
                        k += 1
                    # dequantise (default biases), 18181-1 F.3
                    qa = np.array(qc, dtype=np.float64)
                    a = np.abs(qa)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        adj = np.where(a == 0, 0.0, np.where(a == 1, np.sign(qa) * _QUANT_BIAS[c], qa - _QUANT_BIAS[3] / qa))
                    mul = x_mul if c == 0 else b_mul if c == 2 else 1.0
                    coefs[c] = adj * _W[wofs[c]:wofs[c] + size] / (scale * q) / mul
                # chroma from luma on the AC coefficients
                tx, ty = bx // 8, by // 8
                coefs[0] += (cfl[0, ty, tx] / 84.0) * coefs[1]
                coefs[2] += (1.0 + cfl[1, ty, tx] / 84.0) * coefs[1]
                # lowest frequencies from the DC image
                for c in range(3):
                    dcv = []
                    for j in range(covered):
                        yy, xx = (by + j, bx) if cy == 2 else (by, bx + j)
                        v = qdc[c, yy, xx] * dc_step[c]
                        if c == 2:
                            v += qdc[1, yy, xx] * dc_step[1]  # default DC correlation of B: 1.0
                        dcv.append(v)
                    if covered == 1:
                        coefs[c, 0] = dcv[0]
                    else:
                        coefs[c, 0] = 0.5 * (dcv[0] + dcv[1])
                        coefs[c, 1] = 0.5 * (dcv[0] - dcv[1]) / _LLF_SCALE
                    px = _block_pixels(name, coefs[c])
                    xyb[c, by * 8:by * 8 + px.shape[0], bx * 8:bx * 8 + px.shape[1]] = px
    _finish(rd)
    out.quant_dc_image, out.strategy, out.raw_quant, out.cfl = qdc, strat, rawq, cfl
    out.xyb = xyb[:, :ys, :xs]
    out.linear_rgb = xyb_to_linear(out.xyb)
    return out


def _u64(br):
    sel = br.read(2)
    if sel == 0:
        return 0
    if sel == 1:
        return br.read(4) + 1
    if sel == 2:
        return br.read(8) + 17
    v = br.read(12)
    shift = 12
    while br.read(1):
        if shift == 60:
            v |= br.read(4) << shift
            break
        v |= br.read(8) << shift
        shift += 8
    return v


_OPSIN = np.array([[0.30, 1.0 - 0.078 - 0.30, 0.078], [0.23, 1.0 - 0.078 - 0.23, 0.078],
                   [0.24342268924547819, 0.20476744424496821, 1.0 - 0.24342268924547819 - 0.20476744424496821]])
_OPSIN_BIAS = 0.0037930732552754493


def xyb_to_linear(xyb):
    cb = _OPSIN_BIAS ** (1.0 / 3)
    lms = np.stack([xyb[1] + xyb[0], xyb[1] - xyb[0], xyb[2]]) + cb
    mixed = lms ** 3 - _OPSIN_BIAS
    return np.tensordot(np.linalg.inv(_OPSIN), mixed, axes=1)


def psnr_db(reference_linear, decoded_linear):
    """PSNR of the display-referred (gamma 2.2) signals, peak 1."""
    g = lambda v: np.clip(v, 0, None) ** (1 / 2.2)  # noqa: E731
    mse = float(np.mean((g(np.asarray(reference_linear, dtype=np.float64)) - g(decoded_linear)) ** 2))
    return 10 * math.log10(1.0 / max(mse, 1e-20))


def psnr_opsin_db(reference_linear, decoded_linear):
    """PSNR in the cube-root LMS domain the codec works in (no singularity at black, unlike a
    display gamma); peak = the range of that signal for linear values in [0, 1]."""
    def f(v):
        mixed = np.tensordot(_OPSIN, np.asarray(v, dtype=np.float64), axes=1) + _OPSIN_BIAS
        return np.cbrt(np.maximum(mixed, 0.0))
    peak = (1.0 + _OPSIN_BIAS) ** (1.0 / 3) - _OPSIN_BIAS ** (1.0 / 3)
    mse = float(np.mean((f(reference_linear) - f(decoded_linear)) ** 2))
    return 10 * math.log10(peak * peak / max(mse, 1e-20))
