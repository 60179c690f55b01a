"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol the two headers declare, and the host-arithmetic entry points behave like the
reference's.  No compute calls are made here (those are the -m gpu tests)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from csnappy_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csnappy_\w+)\s*\(", text)))


def test_headers_declare_what_the_binding_lists():
    assert declared("csnappy.h") == sorted(api.LEGACY_SYMBOLS)
    assert declared("csnappy_hip.h") == sorted(api.HIP_SYMBOLS)
    assert declared("csnappy_frame.h") == sorted(api.FRAME_SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    L = api.lib()
    for name in api.LEGACY_SYMBOLS + api.HIP_SYMBOLS + api.FRAME_SYMBOLS:
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(api.LEGACY_SYMBOLS + api.HIP_SYMBOLS + api.FRAME_SYMBOLS) <= exported
    # the code object for gfx950 is embedded
    blob = open(api.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_reference_header_compiles_against_our_header_as_c():
    """A C translation unit written against the reference's csnappy.h compiles against ours."""
    src = r"""
    #include "csnappy.h"
    #if CSNAPPY_VERSION != 5 || CSNAPPY_WORKMEM_BYTES != 65536 || CSNAPPY_E_DATA_MALFORMED != -5
    #error macros differ
    #endif
    uint32_t (*a)(uint32_t) = csnappy_max_compressed_length;
    char *(*b)(const char *, const uint32_t, char *, void *, const int) = csnappy_compress_fragment;
    void (*c)(const char *, uint32_t, char *, uint32_t *, void *, const int) = csnappy_compress;
    int (*d)(const char *, uint32_t, uint32_t *) = csnappy_get_uncompressed_length;
    int (*e)(const char *, uint32_t, char *, uint32_t) = csnappy_decompress;
    int (*f)(const char *, uint32_t, char *, uint32_t *) = csnappy_decompress_noheader;
    int main(void) { return a && b && c && d && e && f ? 0 : 1; }
    """
    subprocess.run(["gcc", "-std=gnu89", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", "-I",
                    os.path.join(ROOT, "include"), "-"], input=src, text=True, check=True)


def test_host_arithmetic_entry_points():
    # csnappy_compress.c:612-616 incl. the uint32 wrap
    for n, want in ((0, 32), (1, 33), (4096, 4810), (32768, 38261), (65536, 76490), (0xFFFFFFFF, 715827913)):
        assert api.max_compressed_length(n) == want
    # csnappy_decompress.c:45-71
    assert api.get_uncompressed_length(b"") [0] == -1
    assert api.get_uncompressed_length(bytes.fromhex("ffffffffff01"))[0] == -1
    assert api.get_uncompressed_length(bytes.fromhex("ffffffff7f")) == (5, 0xFFFFFFFF)
    assert api.get_uncompressed_length(bytes.fromhex("80"))[0] == -1
    assert api.get_uncompressed_length(bytes.fromhex("808004")) == (3, 65536)
    assert api.get_uncompressed_length(bytes.fromhex("87ed2a")) == (3, 702087)


def test_no_cpu_fallback_without_a_device():
    if api.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(RuntimeError):
        api.compress(b"abc")
    # the C entry point itself reports the missing device instead of decoding on the host
    L = api.lib()
    src = np.frombuffer(bytes.fromhex("0308616263"), dtype=np.uint8)
    dst = np.zeros(8, np.uint8)
    assert L.csnappy_decompress(src.ctypes.data, 5, dst.ctypes.data, 8) == api.E_HIP_UNAVAILABLE
    assert not dst.any()
    assert L.csnappy_hip_compress_workspace_size(16384, 65536) >= 16384 * 38261


def test_product_never_touches_the_oracle():
    """Nothing under csnappy_amd/, include/ or tools/ may reference oracle/."""
    for base in ("csnappy_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in text.replace("no CPU", ""), os.path.join(dirpath, f)


def test_batch_api_rejects_bad_arguments_before_touching_the_device():
    """Argument checks come first, so they can be exercised without a GPU."""
    L = api.lib()
    E_ARG, E_WS = -101, -103
    ws = L.csnappy_hip_compress_workspace_size(4, 65536)
    fake = 0x10000  # never dereferenced: every call below fails validation first
    call = lambda nblocks, max_len, p, mode, wsp, wsb: L.csnappy_hip_compress_batch(
        fake, fake, fake, nblocks, max_len, fake, fake, fake, p, mode, wsp, wsb, None)
    assert call(4, 65536, 8, api.STREAM, fake, ws) == E_ARG          # p below 9
    assert call(4, 65536, 17, api.STREAM, fake, ws) == E_ARG         # p above 16
    assert call(4, 65536, 16, 7, fake, ws) == E_ARG                  # unknown mode
    assert call(4, 32769, 13, api.FRAGMENT, fake, ws) == E_ARG       # a fragment is at most 32 KiB
    assert call(4, 65536, 16, api.STREAM, fake, ws - 1) == E_WS      # workspace too small
    assert call(4, 65536, 16, api.STREAM, fake + 8, ws) == E_WS      # workspace not 256-byte aligned
    assert call(0, 65536, 16, api.STREAM, fake, ws) == 0             # empty batch: nothing to do
    assert L.csnappy_hip_decompress_batch(fake, fake, fake, 4, fake, fake, fake, fake, fake, 9, None) == E_ARG
    assert L.csnappy_hip_decompress_batch(fake, fake, fake, 0, fake, fake, fake, fake, fake, api.STREAM, None) == 0
    assert L.csnappy_hip_workload_generate(5, 1, 0, 1, 64, fake, None) == E_ARG
    # workspace: 8-byte records (one per 4 input bytes at most) + 2-byte bucket ids per fragment
    assert L.csnappy_hip_compress_workspace_size(16384, 65536) >= 32768 * (8200 * 8 + 65536)
    assert L.csnappy_hip_compress_workspace_size(16384, 4096) < L.csnappy_hip_compress_workspace_size(16384, 65536)
    # a batch is parsed in launches of 1 GiB of input (32768 full fragments): the least workspace stops growing there
    assert L.csnappy_hip_compress_workspace_size(1 << 20, 65536) == L.csnappy_hip_compress_workspace_size(16384, 65536)
    # 4 KiB pages: the least the call accepts is launches of 32768 pages (a caller's pooled scratch of about
    # 0.5 GiB keeps working); launches of 1 GiB of pages are what ..._size_for(.., 1) sizes for
    L.csnappy_hip_compress_workspace_size_for.restype = C.c_size_t
    L.csnappy_hip_compress_workspace_size_for.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    floor = L.csnappy_hip_compress_workspace_size(1 << 22, 4096)
    assert floor == L.csnappy_hip_compress_workspace_size(32768, 4096) and floor < 600 << 20
    assert L.csnappy_hip_compress_workspace_size_for(1 << 22, 4096, 1) > 7 * floor
    assert L.csnappy_hip_compress_workspace_size_for(1 << 20, 65536, 1) == L.csnappy_hip_compress_workspace_size(1 << 20, 65536)


@pytest.mark.parametrize("name,value", [
    ("CSNAPPY_HIP_S_ENTRIES", "0"), ("CSNAPPY_HIP_S_ENTRIES", "48"), ("CSNAPPY_HIP_S_ENTRIES", "8192"),
    ("CSNAPPY_HIP_DENSE_CAP", "0"), ("CSNAPPY_HIP_DENSE_CAP", "100"), ("CSNAPPY_HIP_DENSE_CAP", "99999"),
    ("CSNAPPY_HIP_WGS_PER_CU", "0"), ("CSNAPPY_HIP_WGS_PER_CU", "x"), ("CSNAPPY_HIP_TABLE", "nonsense"),
    ("CSNAPPY_HIP_SPILL_CAP", "100"), ("CSNAPPY_HIP_SPILL_CAP", "16384"),
])
def test_experiment_knobs_are_range_checked(name, value, monkeypatch):
    """The CSNAPPY_HIP_* environment knobs exist for experiments; a value outside its range makes
    the batch call fail with CSNAPPY_HIP_E_ARG instead of reaching a kernel."""
    L = api.lib()
    fake = 0x10000
    ws = L.csnappy_hip_compress_workspace_size(4, 65536)
    call = lambda: L.csnappy_hip_compress_batch(fake, fake, fake, 4, 65536, fake, fake, fake, 16, api.STREAM,
                                                fake, ws, None)
    monkeypatch.setenv(name, value)
    api.reload_knobs()  # the library reads its knobs once; this is the debug entry that re-reads them
    try:
        assert call() == -101
    finally:
        monkeypatch.delenv(name)
        api.reload_knobs()


def test_gather_layout_and_the_c_gather_example():
    """csnappy_hip_gather_layout is host arithmetic; tools/gather_rccl_example.c -- the C sequence
    compact -> size exchange -> grouped ncclSend/ncclRecv that assembles the final stream of a
    block-sharded batch -- must compile against the product header and the image's HIP and RCCL
    headers, warning-free, and the build must have linked it as a program (it runs in the -m gpu
    suite: tests/test_gpu_parity.py::test_the_c_gather_example_runs_with_one_rank)."""
    import subprocess
    assert api.gather_layout([5, 0, 7, 1 << 40]) == ([0, 5, 5, 12], 12 + (1 << 40))
    assert api.gather_layout([]) == ([], 0)
    src = os.path.join(ROOT, "tools", "gather_rccl_example.c")
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I/opt/rocm/include", src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = os.path.join(ROOT, "tools", "gather_rccl_example")
    assert os.path.exists(exe)
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libcsnappy.so" in needed and "librccl.so" in needed
