"""oracle -- TEST INFRASTRUCTURE (the CPU checker), not product code.

ctypes bindings for
  * ``liboracle.so``             the from-scratch restatement in snappy_oracle.c ("port"), and
  * ``_ref/libcsnappy_ref.so``   the real reference compiled from /root/reference ("reference"),
                                 present only when it was built in the build container.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this
package.  Nothing under csnappy_amd/ does.
"""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libcsnappy_ref.so")

E_OK, E_HEADER_BAD, E_OUTPUT_INSUF, E_OUTPUT_OVERRUN, E_DATA_MALFORMED = 0, -1, -2, -3, -5
STREAM, FRAGMENT = 0, 1

_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile the restatement (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "snappy_oracle.c")
    stale = not os.path.exists(_PORT_SO) or os.path.getmtime(_PORT_SO) < os.path.getmtime(src)
    want_ref = os.path.exists("/root/reference/csnappy_compress.c") and not os.path.exists(_REF_SO)
    if force or stale or want_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def _ptr(a):
    return a.ctypes.data_as(_u8p)


def _as_u8(buf):
    if isinstance(buf, np.ndarray):
        assert buf.dtype == np.uint8 and buf.flags.c_contiguous
        return buf
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def max_compressed_length(n):
    return (32 + n + n // 6) & 0xFFFFFFFF


class _Codec:
    """One implementation of the six csnappy.h entry points, called via ctypes."""

    kind = None

    def compress(self, data, p=16):
        raise NotImplementedError

    # -- helpers shared by both implementations -------------------------------------------
    def compress_blocks(self, data, block, p=16, mode=STREAM):
        """Cut `data` into `block`-byte units, compress each; return list of bytes."""
        data = _as_u8(data)
        out = []
        for s in range(0, max(len(data), 1), block):
            chunk = data[s:s + block]
            out.append(self.compress(chunk, p) if mode == STREAM else self.compress_fragment(chunk, p))
        return out


class Port(_Codec):
    kind = "port"

    def __init__(self):
        build()
        L = C.CDLL(_PORT_SO)
        L.orc_max_compressed_length.restype = C.c_uint32
        L.orc_max_compressed_length.argtypes = [C.c_uint32]
        L.orc_compress_fragment.restype = C.c_uint32
        L.orc_compress_fragment.argtypes = [_u8p, C.c_uint32, _u8p, C.c_int]
        L.orc_fragment_table_power.restype = C.c_int
        L.orc_fragment_table_power.argtypes = [C.c_uint32, C.c_int]
        L.orc_compress.restype = None
        L.orc_compress.argtypes = [_u8p, C.c_uint32, _u8p, C.POINTER(C.c_uint32), C.c_int]
        L.orc_get_uncompressed_length.restype = C.c_int
        L.orc_get_uncompressed_length.argtypes = [_u8p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_decompress.restype = C.c_int
        L.orc_decompress.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32]
        L.orc_decompress_noheader.restype = C.c_int
        L.orc_decompress_noheader.argtypes = [_u8p, C.c_uint32, _u8p, C.POINTER(C.c_uint32)]
        vp = C.c_void_p
        L.orc_batch_compress.restype = None
        L.orc_batch_compress.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_int,
                                         C.c_int, vp, vp]
        L.orc_batch_decompress.restype = None
        L.orc_batch_decompress.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp,
                                           C.c_int, vp, vp]
        self.L = L

    def max_compressed_length(self, n):
        return self.L.orc_max_compressed_length(n)

    def compress_fragment(self, data, p):
        data = _as_u8(data)
        out = np.empty(max_compressed_length(len(data)), dtype=np.uint8)
        n = self.L.orc_compress_fragment(_ptr(data), len(data), _ptr(out), p)
        return out[:n].tobytes()

    def compress(self, data, p=16):
        data = _as_u8(data)
        out = np.empty(max_compressed_length(len(data)), dtype=np.uint8)
        n = C.c_uint32(0)
        self.L.orc_compress(_ptr(data), len(data), _ptr(out), C.byref(n), p)
        return out[:n.value].tobytes()

    def get_uncompressed_length(self, src):
        src = _as_u8(src)
        r = C.c_uint32(0xDEADBEEF)
        rc = self.L.orc_get_uncompressed_length(_ptr(src), len(src), C.byref(r))
        return rc, r.value

    def decompress(self, src, dst_len):
        """-> (status, bytes written region of dst_len bytes)"""
        src = _as_u8(src)
        dst = np.zeros(max(dst_len, 1), dtype=np.uint8)
        rc = self.L.orc_decompress(_ptr(src), len(src), _ptr(dst), dst_len)
        return rc, dst[:dst_len].tobytes()

    def decompress_noheader(self, src, dst_cap):
        """-> (status, produced, bytes)"""
        src = _as_u8(src)
        dst = np.zeros(max(dst_cap, 1), dtype=np.uint8)
        n = C.c_uint32(dst_cap)
        rc = self.L.orc_decompress_noheader(_ptr(src), len(src), _ptr(dst), C.byref(n))
        return rc, n.value, dst[:n.value if rc == 0 else 0].tobytes()

    # function pointers for the batch drivers (None = the port's own)
    def _fnptrs(self):
        return None, None, None, None


class Ref(_Codec):
    """The compiled reference.  Raises OSError if oracle/_ref is absent."""
    kind = "reference"

    def __init__(self):
        build()
        L = C.CDLL(_REF_SO)
        cp = C.c_char_p
        L.csnappy_max_compressed_length.restype = C.c_uint32
        L.csnappy_max_compressed_length.argtypes = [C.c_uint32]
        L.csnappy_compress_fragment.restype = C.c_void_p
        L.csnappy_compress_fragment.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.csnappy_compress.restype = None
        L.csnappy_compress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32),
                                       C.c_void_p, C.c_int]
        L.csnappy_get_uncompressed_length.restype = C.c_int
        L.csnappy_get_uncompressed_length.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.csnappy_decompress.restype = C.c_int
        L.csnappy_decompress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.csnappy_decompress_noheader.restype = C.c_int
        L.csnappy_decompress_noheader.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p,
                                                  C.POINTER(C.c_uint32)]
        del cp
        self.L = L
        self._wm = np.zeros(1 << 16, dtype=np.uint8)

    def max_compressed_length(self, n):
        return self.L.csnappy_max_compressed_length(n)

    def compress_fragment(self, data, p):
        data = _as_u8(data)
        out = np.zeros(max_compressed_length(len(data)) + 64, dtype=np.uint8)
        end = self.L.csnappy_compress_fragment(data.ctypes.data, len(data), out.ctypes.data,
                                               self._wm.ctypes.data, p)
        return out[:end - out.ctypes.data].tobytes()

    def compress(self, data, p=16):
        data = _as_u8(data)
        out = np.zeros(max_compressed_length(len(data)) + 64, dtype=np.uint8)
        n = C.c_uint32(0)
        self.L.csnappy_compress(data.ctypes.data, len(data), out.ctypes.data, C.byref(n),
                                self._wm.ctypes.data, p)
        return out[:n.value].tobytes()

    def get_uncompressed_length(self, src):
        src = _as_u8(src)
        pad = np.concatenate([src, np.zeros(8, np.uint8)])
        r = C.c_uint32(0xDEADBEEF)
        rc = self.L.csnappy_get_uncompressed_length(pad.ctypes.data, len(src), C.byref(r))
        return rc, r.value

    def decompress(self, src, dst_len):
        src = _as_u8(src)
        pad = np.concatenate([src, np.zeros(8, np.uint8)])
        dst = np.zeros(max(dst_len, 1) + 64, dtype=np.uint8)
        rc = self.L.csnappy_decompress(pad.ctypes.data, len(src), dst.ctypes.data, dst_len)
        return rc, dst[:dst_len].tobytes()

    def decompress_noheader(self, src, dst_cap):
        src = _as_u8(src)
        pad = np.concatenate([src, np.zeros(8, np.uint8)])
        dst = np.zeros(max(dst_cap, 1) + 64, dtype=np.uint8)
        n = C.c_uint32(dst_cap)
        rc = self.L.csnappy_decompress_noheader(pad.ctypes.data, len(src), dst.ctypes.data, C.byref(n))
        return rc, n.value, dst[:n.value if rc == 0 else 0].tobytes()

    def _fnptrs(self):
        g = lambda name: C.cast(getattr(self.L, name), C.c_void_p).value
        return (g("csnappy_compress"), g("csnappy_compress_fragment"), g("csnappy_decompress"),
                g("csnappy_decompress_noheader"))


def have_ref():
    return os.path.exists(_REF_SO)


def best():
    """The strongest checker available: the compiled reference if present, else the port."""
    return Ref() if have_ref() else Port()


# ---------------------------------------------------------------------------------------------
# Batched drivers over host numpy arrays (same descriptor layout as the HIP batch API): used to
# check whole batches and to time the CPU baseline on `threads` host cores.
# ---------------------------------------------------------------------------------------------
def batch_compress(codec, data, in_off, in_len, out_off, out_bytes, p, mode, threads=1, out=None):
    """`out`: optional preallocated (and already touched) uint8 buffer of out_bytes + 64."""
    port = Port()
    cfn, ffn, _, _ = codec._fnptrs()
    data = _as_u8(data)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    if out is None:
        out = np.zeros(out_bytes + 64, dtype=np.uint8)
    out_len = np.zeros(len(in_len), dtype=np.uint32)
    nb = len(in_len)

    def run(lo, hi):
        port.L.orc_batch_compress(data.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, lo, hi,
                                  out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, p, mode,
                                  cfn, ffn)
    _par(run, nb, threads)
    return out[:out_bytes], out_len


def batch_decompress(codec, data, in_off, in_len, out_off, out_cap, out_bytes, mode, threads=1, out=None):
    """`data` must be readable 16 bytes past its last block when `out` is given (no copy is made)."""
    port = Port()
    _, _, dfn, nfn = codec._fnptrs()
    data = np.concatenate([_as_u8(data), np.zeros(16, np.uint8)]) if out is None else _as_u8(data)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    if out is None:
        out = np.zeros(out_bytes + 64, dtype=np.uint8)
    nb = len(in_len)
    status = np.zeros(nb, dtype=np.int32)
    produced = np.zeros(nb, dtype=np.uint32)

    def run(lo, hi):
        port.L.orc_batch_decompress(data.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, lo, hi,
                                    out.ctypes.data, out_off.ctypes.data, out_cap.ctypes.data,
                                    status.ctypes.data, produced.ctypes.data, mode, dfn, nfn)
    _par(run, nb, threads)
    return out[:out_bytes], status, produced


def _par(run, nb, threads):
    if threads <= 1 or nb < 2 * threads:
        run(0, nb)
        return
    cuts = [nb * i // threads for i in range(threads + 1)]
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda i: run(cuts[i], cuts[i + 1]), range(threads)))
