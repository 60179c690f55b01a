"""ctypes binding of libcsnappy.so (include/csnappy.h + include/csnappy_hip.h).

This module is plumbing for tests and bench.py: it loads the C-ABI library and passes raw
device pointers (torch tensors are used only as owners of HBM allocations and for the stream).
There is no Python or CPU implementation of the codec here -- if the library is missing, loading
fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CSNAPPY_AMD_LIB: development only -- time a differently built library with the same tools)
LIB_PATH = os.environ.get("CSNAPPY_AMD_LIB") or os.path.join(_HERE, "lib", "libcsnappy.so")

STREAM, FRAGMENT = 0, 1
E_OK, E_HEADER_BAD, E_OUTPUT_INSUF, E_OUTPUT_OVERRUN, E_DATA_MALFORMED = 0, -1, -2, -3, -5
E_HIP_UNAVAILABLE = -100
WG_TEXT, WG_LOW, WG_PAGE = 0, 1, 2
FRAGMENT_BYTES = 32768

# every symbol the two headers declare (tests check the library exports exactly these)
LEGACY_SYMBOLS = [
    "csnappy_max_compressed_length", "csnappy_compress_fragment", "csnappy_compress",
    "csnappy_get_uncompressed_length", "csnappy_decompress", "csnappy_decompress_noheader",
]
HIP_SYMBOLS = [
    "csnappy_hip_device_count", "csnappy_hip_last_error", "csnappy_hip_compress_workspace_size",
    "csnappy_hip_compress_workspace_size_for",
    "csnappy_hip_compress_batch", "csnappy_hip_decompress_batch", "csnappy_hip_set_kernel_timing",
    "csnappy_hip_get_kernel_timing", "csnappy_hip_workload_generate", "csnappy_hip_compact_batch",
    "csnappy_workload_generate_host", "csnappy_hip_decompress_stream_workspace_size",
    "csnappy_hip_decompress_stream", "csnappy_hip_decompress_stream_took_fast_path",
    "csnappy_hip_dense_offsets_workspace_size", "csnappy_hip_dense_offsets", "csnappy_hip_gather_layout",
]

FRAME_SYMBOLS = [
    "csnappy_frame_max_compressed_length", "csnappy_frame_compress", "csnappy_frame_uncompressed_length",
    "csnappy_frame_decompress", "csnappy_hip_crc32c_batch", "csnappy_frame_release",
]
FRAME_E_NO_IDENTIFIER, FRAME_E_BAD_CHUNK, FRAME_E_CRC, FRAME_E_OUTPUT_INSUF, FRAME_E_DATA = -201, -202, -203, -204, -205

_lib = None


def lib():
    """Load libcsnappy.so (built by __graft_entry__.build() / csnappy_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (there is no CPU fallback for the codec)")
    # ONE HIP runtime per process: the torch wheel bundles its own libamdhip64.so (SONAME
    # libamdhip64.so.7).  Import torch first so the dynamic linker binds our NEEDED
    # libamdhip64.so.7 to that already-loaded copy; loading /opt/rocm's copy beside it gives two
    # runtimes and the second one to initialise reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.csnappy_max_compressed_length.restype = u32
    L.csnappy_max_compressed_length.argtypes = [u32]
    L.csnappy_compress_fragment.restype = vp
    L.csnappy_compress_fragment.argtypes = [vp, u32, vp, vp, i32]
    L.csnappy_compress.restype = None
    L.csnappy_compress.argtypes = [vp, u32, vp, C.POINTER(u32), vp, i32]
    L.csnappy_get_uncompressed_length.restype = i32
    L.csnappy_get_uncompressed_length.argtypes = [vp, u32, C.POINTER(u32)]
    L.csnappy_decompress.restype = i32
    L.csnappy_decompress.argtypes = [vp, u32, vp, u32]
    L.csnappy_decompress_noheader.restype = i32
    L.csnappy_decompress_noheader.argtypes = [vp, u32, vp, C.POINTER(u32)]
    L.csnappy_hip_device_count.restype = i32
    L.csnappy_hip_last_error.restype = C.c_char_p
    L.csnappy_hip_compress_workspace_size.restype = C.c_size_t
    L.csnappy_hip_compress_workspace_size.argtypes = [u32, u32]
    L.csnappy_hip_compress_workspace_size_for.restype = C.c_size_t
    L.csnappy_hip_compress_workspace_size_for.argtypes = [u32, u32, u32]
    L.csnappy_hip_compress_batch.restype = i32
    L.csnappy_hip_compress_batch.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp, i32, i32, vp,
                                             C.c_size_t, vp]
    L.csnappy_hip_decompress_batch.restype = i32
    L.csnappy_hip_decompress_batch.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp, vp, i32, vp]
    L.csnappy_hip_decompress_stream_workspace_size.restype = C.c_size_t
    L.csnappy_hip_decompress_stream_workspace_size.argtypes = [u32, u32]
    L.csnappy_hip_decompress_stream.restype = i32
    L.csnappy_hip_decompress_stream.argtypes = [vp, u32, u32, vp, vp, vp, vp, C.c_size_t, vp]
    L.csnappy_hip_decompress_stream_took_fast_path.restype = i32
    L.csnappy_hip_decompress_stream_took_fast_path.argtypes = [vp, u32, u32, vp]
    L.csnappy_hip_compact_batch.restype = i32
    L.csnappy_hip_compact_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    L.csnappy_hip_dense_offsets_workspace_size.restype = C.c_size_t
    L.csnappy_hip_dense_offsets_workspace_size.argtypes = [u32]
    L.csnappy_hip_dense_offsets.restype = i32
    L.csnappy_hip_dense_offsets.argtypes = [vp, u32, vp, vp, vp, C.c_size_t, vp]
    L.csnappy_hip_gather_layout.restype = None
    L.csnappy_hip_gather_layout.argtypes = [vp, u32, vp, vp]
    L.csnappy_hip_set_kernel_timing.restype = None
    L.csnappy_hip_set_kernel_timing.argtypes = [i32]
    L.csnappy_hip_get_kernel_timing.restype = None
    L.csnappy_hip_get_kernel_timing.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.csnappy_hip_workload_generate.restype = i32
    L.csnappy_hip_workload_generate.argtypes = [i32, u64, u64, u32, u32, vp, vp]
    L.csnappy_frame_max_compressed_length.restype = C.c_size_t
    L.csnappy_frame_max_compressed_length.argtypes = [C.c_size_t]
    L.csnappy_frame_compress.restype = i32
    L.csnappy_frame_compress.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_size_t), i32]
    L.csnappy_frame_uncompressed_length.restype = i32
    L.csnappy_frame_uncompressed_length.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.csnappy_frame_decompress.restype = i32
    L.csnappy_frame_decompress.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_size_t)]
    L.csnappy_hip_crc32c_batch.restype = i32
    L.csnappy_hip_crc32c_batch.argtypes = [vp, vp, vp, u32, vp, vp]
    L.csnappy_workload_generate_host.restype = None
    L.csnappy_workload_generate_host.argtypes = [i32, u64, u64, u32, u32, vp]
    _lib = L
    return L


def reload_knobs():
    """Tests only: make the library read the CSNAPPY_HIP_* environment knobs again (it reads them
    once, at the first batch call)."""
    f = lib().csnappy_hip_debug_reload_knobs
    f.restype, f.argtypes = None, []
    f()


def device_count():
    return lib().csnappy_hip_device_count()


def require_device():
    if device_count() <= 0:
        raise RuntimeError("no usable HIP device: the csnappy codec only runs on the GPU "
                           "(no CPU fallback)")


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: rc={rc} ({lib().csnappy_hip_last_error().decode()})")


# ---------------------------------------------------------------------------------------------
# the six csnappy.h calls on host buffers (each is one 1-block launch on the GPU)
# ---------------------------------------------------------------------------------------------
def _u8(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf, dtype=np.uint8)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def max_compressed_length(n):
    return lib().csnappy_max_compressed_length(n)


def compress(data, p=16):
    require_device()
    data = _u8(data)
    out = np.empty(max_compressed_length(len(data)) + 8, dtype=np.uint8)
    n = C.c_uint32(0)
    lib().csnappy_compress(data.ctypes.data, len(data), out.ctypes.data, C.byref(n), None, p)
    return out[:n.value].tobytes()


def compress_fragment(data, p):
    require_device()
    data = _u8(data)
    out = np.empty(max_compressed_length(len(data)) + 8, dtype=np.uint8)
    end = lib().csnappy_compress_fragment(data.ctypes.data, len(data), out.ctypes.data, None, p)
    return out[:end - out.ctypes.data].tobytes()


def get_uncompressed_length(src):
    src = _u8(src)
    r = C.c_uint32(0xDEADBEEF)
    rc = lib().csnappy_get_uncompressed_length(src.ctypes.data if len(src) else None, len(src),
                                               C.byref(r))
    return rc, r.value


def decompress(src, dst_len):
    """-> (status, dst bytes[:dst_len])"""
    require_device()
    src = _u8(src)
    dst = np.zeros(max(dst_len, 1), dtype=np.uint8)
    rc = lib().csnappy_decompress(src.ctypes.data if len(src) else None, len(src), dst.ctypes.data,
                                  dst_len)
    return rc, dst[:dst_len].tobytes()


def decompress_noheader(src, dst_cap):
    """-> (status, produced, bytes[:produced])"""
    require_device()
    src = _u8(src)
    dst = np.zeros(max(dst_cap, 1), dtype=np.uint8)
    n = C.c_uint32(dst_cap)
    rc = lib().csnappy_decompress_noheader(src.ctypes.data if len(src) else None, len(src),
                                           dst.ctypes.data, C.byref(n))
    return rc, n.value, dst[:n.value if rc == 0 else 0].tobytes()


# ---------------------------------------------------------------------------------------------
# the framing format (include/csnappy_frame.h) on host buffers
# ---------------------------------------------------------------------------------------------
def frame_compress(data, p=16):
    """-> (rc, framed bytes)"""
    data = _u8(data)
    cap = C.c_size_t(lib().csnappy_frame_max_compressed_length(len(data)))
    out = np.empty(cap.value + 8, dtype=np.uint8)
    rc = lib().csnappy_frame_compress(data.ctypes.data if len(data) else None, len(data), out.ctypes.data,
                                      C.byref(cap), p)
    return rc, out[:cap.value].tobytes() if rc == 0 else b""


def frame_uncompressed_length(stream):
    stream = _u8(stream)
    r = C.c_size_t(0)
    rc = lib().csnappy_frame_uncompressed_length(stream.ctypes.data if len(stream) else None, len(stream), C.byref(r))
    return rc, r.value


def frame_decompress(stream, dst_cap):
    """-> (rc, bytes)"""
    stream = _u8(stream)
    dst = np.zeros(max(dst_cap, 1), dtype=np.uint8)
    n = C.c_size_t(dst_cap)
    rc = lib().csnappy_frame_decompress(stream.ctypes.data if len(stream) else None, len(stream), dst.ctypes.data,
                                        C.byref(n))
    return rc, dst[:n.value].tobytes() if rc == 0 else b""


# ---------------------------------------------------------------------------------------------
# batched API on device tensors
# ---------------------------------------------------------------------------------------------
def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def workspace_size(nblocks, max_in_len, launch_gib=1):
    """Scratch for launches of up to launch_gib GiB of input (0: launches of 32 768 fragments whatever
    their size -- csnappy_hip_compress_workspace_size, the least the batch call accepts)."""
    return lib().csnappy_hip_compress_workspace_size_for(nblocks, max_in_len, launch_gib)


def compress_batch(d_in, in_off, in_len, max_in_len, d_out, out_off, out_len, p, mode, workspace):
    """All arguments are CUDA(HIP) tensors: uint8 data, int64 offsets, int32 lengths."""
    nblocks = in_len.numel()
    rc = lib().csnappy_hip_compress_batch(
        d_in.data_ptr(), in_off.data_ptr(), in_len.data_ptr(), nblocks, max_in_len,
        d_out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), p, mode,
        workspace.data_ptr(), workspace.numel(), _stream())
    _check(rc, "csnappy_hip_compress_batch")


def decompress_batch(d_in, in_off, in_len, d_out, out_off, out_cap, status, produced, mode):
    nblocks = in_len.numel()
    rc = lib().csnappy_hip_decompress_batch(
        d_in.data_ptr(), in_off.data_ptr(), in_len.data_ptr(), nblocks, d_out.data_ptr(),
        out_off.data_ptr(), out_cap.data_ptr(), status.data_ptr(), produced.data_ptr(), mode,
        _stream())
    _check(rc, "csnappy_hip_decompress_batch")


def decompress_stream(d_body, ulength, d_out):
    """One stream body (no length header) of any length, device to device: -> (status, produced,
    took_fast_path).  d_out must hold ulength bytes."""
    import torch
    n = d_body.numel()
    need = lib().csnappy_hip_decompress_stream_workspace_size(n, ulength)
    ws = torch.empty(need + 16, dtype=torch.uint8, device=d_body.device)
    ws = ws[(-ws.data_ptr()) % 16:]
    res = torch.zeros(2, dtype=torch.int32, device=d_body.device)
    rc = lib().csnappy_hip_decompress_stream(d_body.data_ptr(), n, ulength, d_out.data_ptr(),
                                             res.data_ptr(), res.data_ptr() + 4, ws.data_ptr(),
                                             need, _stream())
    _check(rc, "csnappy_hip_decompress_stream")
    fast = lib().csnappy_hip_decompress_stream_took_fast_path(ws.data_ptr(), n, ulength, _stream())
    status, produced = res.cpu().tolist()
    return status, produced & 0xFFFFFFFF, bool(fast)


def dense_offsets(out_len):
    """Exclusive sum of the compressed lengths on the device (csnappy_hip_dense_offsets):
    -> (int64 offsets tensor, total bytes)"""
    import torch
    n = out_len.numel()
    off = torch.empty(max(n, 1), dtype=torch.int64, device=out_len.device)
    total = torch.zeros(1, dtype=torch.int64, device=out_len.device)
    need = lib().csnappy_hip_dense_offsets_workspace_size(n)
    ws = torch.empty(need // 8 + 1, dtype=torch.int64, device=out_len.device)
    rc = lib().csnappy_hip_dense_offsets(out_len.data_ptr(), n, off.data_ptr(), total.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 8, _stream())
    _check(rc, "csnappy_hip_dense_offsets")
    return off[:n], int(total.item())


def gather_layout(rank_bytes):
    """csnappy_hip_gather_layout on a list of per-rank byte counts -> (offsets list, total)"""
    rb = np.asarray(rank_bytes, dtype=np.uint64)
    off = np.zeros(len(rb), dtype=np.uint64)
    total = C.c_uint64(0)
    lib().csnappy_hip_gather_layout(rb.ctypes.data, len(rb), off.ctypes.data, C.byref(total))
    return [int(x) for x in off], int(total.value)


def compact_batch(d_out, out_off, out_len, dense_off, dense):
    rc = lib().csnappy_hip_compact_batch(d_out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(),
                                         dense_off.data_ptr(), out_len.numel(), dense.data_ptr(),
                                         _stream())
    _check(rc, "csnappy_hip_compact_batch")


def set_kernel_timing(on):
    lib().csnappy_hip_set_kernel_timing(1 if on else 0)


def get_kernel_timing():
    """-> {kernel: (total_ms, launches)} since the previous read (waits for the events)."""
    ms, cnt = (C.c_float * 4)(), (C.c_uint32 * 4)()
    lib().csnappy_hip_get_kernel_timing(ms, cnt)
    names = ("snappy_parse_fragments", "snappy_emit_blocks", "snappy_decompress_blocks")
    return {n: (ms[i], cnt[i]) for i, n in enumerate(names)}


def generate(kind, seed, first_block, nblocks, block_len, device="cuda"):
    import torch
    out = torch.empty(nblocks * block_len, dtype=torch.uint8, device=device)
    rc = lib().csnappy_hip_workload_generate(kind, seed, first_block, nblocks, block_len,
                                             out.data_ptr(), _stream())
    _check(rc, "csnappy_hip_workload_generate")
    return out


def generate_host(kind, seed, first_block, nblocks, block_len):
    out = np.zeros(nblocks * block_len, dtype=np.uint8)
    lib().csnappy_workload_generate_host(kind, seed, first_block, nblocks, block_len,
                                         out.ctypes.data)
    return out


# ---------------------------------------------------------------------------------------------
# descriptor helper for a batch of blocks cut from one contiguous buffer
# ---------------------------------------------------------------------------------------------
class Batch:
    """Descriptors (host numpy + device tensors) for `lens` blocks laid end to end in the
    input, each with an output slot of csnappy_max_compressed_length(len) bytes rounded up to
    `slot_align`."""

    def __init__(self, lens, slot_align=64, device="cuda", launch_gib=1):
        """launch_gib: size the workspace for parser launches of up to this many GiB of input (1 is
        the least the batch call accepts; larger launches lose less to their ramp and tail)."""
        import torch
        lens = np.asarray(lens, dtype=np.uint32)
        self.n = len(lens)
        self.in_len = lens
        self.in_off = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64) \
            if self.n else np.zeros(0, np.uint64)
        slots = (32 + lens.astype(np.uint64) + lens.astype(np.uint64) // 6 + slot_align - 1) \
            // slot_align * slot_align
        self.slot = slots
        self.out_off = np.concatenate([[0], np.cumsum(slots[:-1], dtype=np.uint64)]).astype(np.uint64) \
            if self.n else np.zeros(0, np.uint64)
        self.in_bytes = int(lens.sum(dtype=np.uint64))
        self.out_bytes = int(slots.sum(dtype=np.uint64))
        self.max_in_len = int(lens.max()) if self.n else 0
        if device is not None:
            t = lambda a, dt: torch.from_numpy(a.astype(dt)).to(device)
            self.d_in_off = t(self.in_off, np.int64)
            self.d_in_len = t(self.in_len, np.int32)
            self.d_out_off = t(self.out_off, np.int64)
            self.d_out_len = torch.zeros(self.n, dtype=torch.int32, device=device)
            need = workspace_size(self.n, self.max_in_len, launch_gib)
            self.d_ws = torch.empty(need + 256, dtype=torch.uint8, device=device)
            # 256-byte aligned view
            off = (-self.d_ws.data_ptr()) % 256
            self.d_ws = self.d_ws[off:off + need]

    @classmethod
    def uniform(cls, total_bytes, block_len, **kw):
        nfull, rem = divmod(total_bytes, block_len)
        lens = [block_len] * nfull + ([rem] if rem else [])
        return cls(lens, **kw)
