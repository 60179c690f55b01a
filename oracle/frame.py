"""oracle.frame -- TEST INFRASTRUCTURE: CPU restatement of the Snappy FRAMING format.

Source status: **absent from /root/reference** (the reference only names framing as a goal,
README:11-17).  What is restated is google/snappy's public framing_format.txt (2013-10-25):
stream identifier ff 06 00 00 "sNaPpY"; chunks = type | 24-bit LE length | data; 0x00 compressed
(masked CRC-32C of the uncompressed bytes + one Snappy block), 0x01 uncompressed, 0xfe padding,
0x80-0xfd skippable, 0x02-0x7f unskippable (error); at most 65536 uncompressed bytes per chunk;
mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8.
Pinned by: the CRC-32C known answers of RFC 3720 B.4 and "123456789" -> e3069283, the
identifier bytes and hand-assembled streams of the spec (tests/test_oracle.py); the chunk BODIES
are the block format the rest of the oracle pins against the compiled reference.  No third-party
framing encoder is in the image to cross-check whole streams: framing parity is spec-pinned.
"""
import numpy as np

CHUNK = 65536
STREAM_ID = bytes([0xff, 0x06, 0x00, 0x00]) + b"sNaPpY"
E_OK, E_NO_IDENTIFIER, E_BAD_CHUNK, E_CRC, E_OUTPUT_INSUF, E_DATA = 0, -201, -202, -203, -204, -205

_POLY = 0x82F63B78
_T = np.zeros(256, dtype=np.uint32)
for _b in range(256):
    _c = _b
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _T[_b] = _c
_TL = [int(x) for x in _T]


def crc32c(data):
    """CRC-32C (Castagnoli), reflected, init and final xor ffffffff (RFC 3720 B.4)."""
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = _TL[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _chunk(ctype, body):
    return bytes([ctype]) + len(body).to_bytes(3, "little") + body


def encode(data, compress, store_raw_when_not_smaller=True):
    """Frame `data`: `compress(chunk) -> Snappy block` (e.g. oracle.Port().compress(x, p))."""
    data = bytes(data)
    out = [STREAM_ID]
    for s in range(0, len(data), CHUNK):
        x = data[s:s + CHUNK]
        c = compress(x)
        crc = mask(crc32c(x)).to_bytes(4, "little")
        if store_raw_when_not_smaller and len(c) >= len(x):
            out.append(_chunk(0x01, crc + x))
        else:
            out.append(_chunk(0x00, crc + c))
    return b"".join(out)


def decode(stream, decompress, uncompressed_length, dst_cap=None):
    """-> (rc, bytes).  decompress(block, n) -> (rc, bytes); uncompressed_length(block) -> (hdr, n)."""
    s = bytes(stream)
    if len(s) < 10 or s[:10] != STREAM_ID:
        return E_NO_IDENTIFIER, b""
    pos, pieces = 0, []
    while pos < len(s):
        if len(s) - pos < 4:
            return E_BAD_CHUNK, b""
        t, ln = s[pos], int.from_bytes(s[pos + 1:pos + 4], "little")
        if len(s) - pos - 4 < ln:
            return E_BAD_CHUNK, b""
        d = s[pos + 4:pos + 4 + ln]
        pos += 4 + ln
        if t == 0xFF:
            if d != b"sNaPpY":
                return E_BAD_CHUNK, b""
            continue
        if t >= 0x80:
            continue
        if t > 0x01 or ln < 4:
            return E_BAD_CHUNK, b""
        crc, body = int.from_bytes(d[:4], "little"), d[4:]
        if t == 0x00:
            hdr, ulen = uncompressed_length(body)
            if hdr < 0:
                return E_BAD_CHUNK, b""
        else:
            ulen = len(body)
        if ulen > CHUNK:
            return E_BAD_CHUNK, b""
        pieces.append((t, crc, body, ulen))
    total = sum(p[3] for p in pieces)
    if dst_cap is not None and total > dst_cap:
        return E_OUTPUT_INSUF, b""
    out, first_err = [], E_OK
    for t, crc, body, ulen in pieces:
        if t == 0x00:
            rc, x = decompress(body, ulen)
            if rc != 0 or len(x) != ulen:
                return E_DATA, b""
        else:
            x = body
        if mask(crc32c(x)) != crc:
            return E_CRC, b""
        out.append(x)
    return first_err, b"".join(out)
