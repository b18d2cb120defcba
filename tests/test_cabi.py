"""C-ABI surface of libfxrx.so and host-side logic that needs no GPU."""
import ctypes as C
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "fxrx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined", "int", "void", "unsigned", "float") and not n.endswith("_callback")))


def test_library_exports_every_declared_symbol(fx):
    L = fx.lib()
    declared = _declared_functions()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), "include/fxrx.h declares %s but libfxrx.so does not export it" % name
    assert sorted(fx._ffi.EXPORTS) == declared                   # the binding covers the whole header


def test_no_gpu_means_loud_failure_not_fallback(fx):
    L = fx.lib()
    if L.fxrx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        fx.RxContext(1)
    assert L.flexframesync_create(fx._ffi.FRAMESYNC_CALLBACK(lambda *a: 0), None) is None
    pn = np.zeros(64, np.complex64)
    assert L.qdetector_cccf_create_linear(pn.ctypes.data, 64, 7, 2, 7, C.c_float(0.3)) is None


def test_dropin_feed_driver_loads_and_fails_loudly_without_a_gpu(fx):
    """csrc/libdropin_feed.so (the C++ flex_rx shell driven in 256-sample flexframesync_execute calls: bench.py's
    value_through_dropin_abi and tests/test_gpu_boundary.py) is built with the library and needs nothing but libfxrx.so."""
    F = fx._ffi.feed_lib()
    assert hasattr(F, "dropin_feed") and hasattr(F, "dropin_feed_threads")
    assert fx._ffi.fnv1a([b"a", b"bc"]) == fx._ffi.fnv1a([b"abc"]) == 0xe71fa2190541574b
    if fx.lib().fxrx_device_count() == 0:
        x = np.zeros(1024, np.complex64); st = fx._ffi.DropinStats()
        assert F.dropin_feed(x.ctypes.data, len(x), 256, 1, C.byref(st)) == -1
        assert b"no usable HIP device" in fx.lib().fxrx_last_error()


def test_index_maps_match_reference_switch_tables(fx):
    L = fx.lib()
    # lib/flex_tx_impl.cc:75-181 and lib/flex_rx_impl.cc:74-179 (inner list skips V27P34: 3 -> P45)
    assert [L.fxrx_mod_from_index(i) for i in range(11)] == [1, 2, 3, 4, 9, 10, 11, 18, 27, 28, 29]
    assert [L.fxrx_inner_from_index(i) for i in range(7)] == [1, 11, 15, 17, 18, 19, 20]
    assert [L.fxrx_outer_from_index(i) for i in range(8)] == [1, 7, 27, 4, 6, 8, 9, 10]
    for i in range(11): assert L.fxrx_mod_to_index(L.fxrx_mod_from_index(i)) == i
    for i in range(7): assert L.fxrx_inner_to_index(L.fxrx_inner_from_index(i)) == i
    for i in range(8): assert L.fxrx_outer_to_index(L.fxrx_outer_from_index(i)) == i
    assert L.fxrx_mod_to_index(40) == -1 and L.fxrx_inner_to_index(16) == -1 and L.fxrx_mod_from_index(11) == -1


def test_msequence_dropin_matches_oracle_preamble(fx, oracle):
    L = fx.lib()
    ms = L.msequence_create(7, 0x0089, 1)                         # lib/frame_detector_cc_impl.cc:47
    pn = np.zeros(64, np.complex64)
    for i in range(64):                                           # :48-51
        re = np.sqrt(0.5) if L.msequence_advance(ms) else -np.sqrt(0.5)
        im = np.sqrt(0.5) if L.msequence_advance(ms) else -np.sqrt(0.5)
        pn[i] = re + 1j * im
    L.msequence_destroy(ms)
    assert np.array_equal(pn, oracle.table("fxr_preamble_pn", 64))


def test_product_tx_is_bit_identical_to_oracle_tx(fx, oracle):
    rng = np.random.default_rng(3)
    for mod in fx.MOD_BY_INDEX:
        for fec0 in fx.INNER_BY_INDEX + [5, 10, 16]:
            n = int(rng.integers(0, 300))
            pl = rng.integers(0, 256, max(n, 1), dtype=np.uint8)[:n]
            hdr = rng.integers(0, 256, 14, dtype=np.uint8)
            dt = float(rng.uniform(-0.5, 0.5))
            a = oracle.gen_frame(pl, mod=mod, fec0=fec0, fec1=1, check=5, header=hdr, dt=dt)
            b = fx.FrameGen(mod, fec0, 1, 5).frame(pl, header=hdr, dt=dt)
            assert len(a) == len(b) == fx.lib().fxrx_gen_frame_len(mod, 5, fec0, 1, n)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (mod, fec0, n)


def test_product_tx_outer_codes_match_oracle(fx, oracle):
    """every outer code the reference's flex_tx can select (lib/flex_tx_impl.cc:148-181)"""
    rng = np.random.default_rng(5)
    for fec1 in (7, 27, 4, 6, 8, 9, 10, 5):                      # Golay, RS, H74, H128, SECDED22/39/72, H84
        for fec0 in (1, 11, 15):
            n = int(rng.integers(1, 200))
            pl = rng.integers(0, 256, n, dtype=np.uint8)
            a = oracle.gen_frame(pl, mod=27, fec0=fec0, fec1=fec1, check=5)
            b = fx.FrameGen(27, fec0, fec1, 5).frame(pl)
            assert len(a) == len(b) and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (fec0, fec1, n)
    for oc in range(8):                                           # every outer code of lib/flex_tx_impl.cc:148-181
        fx.flex_tx.make(1, 1, oc)
    with pytest.raises(ValueError):
        fx.FrameGen(2, 11, 12, 5)                                 # a liquid FEC id outside the reference's menu (CONV_V29)


def test_flexframegen_dropin_contract(fx):
    L = fx.lib()
    p = fx._ffi.GenProps()
    L.flexframegenprops_init_default(C.byref(p))
    assert (p.check, p.fec0, p.fec1, p.mod_scheme) == (6, 1, 1, 40)          # CRC32, none, none, QPSK
    p.check, p.fec0, p.mod_scheme = 5, 11, 2                                  # lib/flex_tx_impl.cc:52 + (1,1,0)
    g = L.flexframegen_create(C.byref(p))
    assert L.flexframegen_getframelen(g) == 0                                 # nothing assembled yet
    pl = np.arange(1024, dtype=np.uint8)
    hdr = np.zeros(14, np.uint8)
    assert L.flexframegen_assemble(g, hdr.ctypes.data, pl.ctypes.data, 1024) == 0
    n = L.flexframegen_getframelen(g)
    assert n == 17066
    buf = np.zeros(n, np.complex64)
    assert L.flexframegen_write_samples(g, buf.ctypes.data, n - 1) == -1      # short buffer refused
    assert L.flexframegen_write_samples(g, buf.ctypes.data, n) == 1
    assert abs(np.mean(np.abs(buf[200:-200]) ** 2) - 1.0) < 0.05               # unit sample power
    p.fec0 = 99
    assert L.flexframegen_setprops(g, C.byref(p)) != 0                        # unsupported FEC rejected
    L.flexframegen_destroy(g)


def test_synth_stream_is_deterministic(fx):
    a, fa = fx.synth_stream(60000, stream_id=5)
    b, fb = fx.synth_stream(60000, stream_id=5)
    c, _ = fx.synth_stream(60000, stream_id=6)
    assert np.array_equal(a, b) and fa == fb and not np.array_equal(a, c)
    assert len(fa) == 3 and fa[1][0] - fa[0][0] == 17066 + 256


def test_cpp_block_shells_compile_against_the_header(fx, tmp_path):
    """csrc/blocks/fx_blocks.hpp needs nothing but include/fxrx.h and -lfxrx; without a GPU construction throws."""
    import subprocess
    exe = str(tmp_path / "test_blocks")
    lib = os.path.join(ROOT, "gr-liquiddsp_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_blocks.cpp"),
                           "-L" + lib, "-lfxrx", "-Wl,-rpath," + lib])
    if fx.lib().fxrx_device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode != 0 and "no usable HIP device" in r.stderr


def _build_callsites(tmp_path, std="c++11"):
    import subprocess
    exe = str(tmp_path / "test_reference_callsites")
    lib = os.path.join(ROOT, "gr-liquiddsp_amd", "csrc")
    subprocess.check_call(["g++", "-std=" + std, "-Wall", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_reference_callsites.cpp"), "-L" + lib, "-lfxrx", "-Wl,-rpath," + lib])
    return exe


@pytest.mark.parametrize("std", ["c++11", "c++17"])
def test_reference_call_sites_compile_through_the_liquid_shim(fx, tmp_path, std):
    """include/liquid/liquid.h lets the reference's call sequences compile as written (gr_complex* / gr_complex by value /
    framesyncstats_s fields) and link against libfxrx.so alone: tests/cpp/test_reference_callsites.cpp.  Running it needs
    a GPU (tests/test_gpu_parity.py); here it must at least build, link, and fail loudly without a device."""
    import subprocess
    exe = _build_callsites(tmp_path, std)
    r = subprocess.run([exe, "--link-only"], capture_output=True, text=True)
    assert r.returncode == 0 and "LINKED" in r.stdout
    if fx.lib().fxrx_device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no usable HIP device" in r.stderr


def test_liquid_shim_is_valid_c(tmp_path):
    """The same header from C (float _Complex), as liquid.h is a C header first."""
    import subprocess
    src = tmp_path / "shim_c.c"
    src.write_text("#include <liquid/liquid.h>\n"
                   "static int cb(unsigned char *h, int hv, unsigned char *p, unsigned int n, int pv, framesyncstats_s st, void *ud)\n"
                   "{ (void)h; (void)hv; (void)p; (void)n; (void)pv; (void)ud; return (int)crealf(st.framesyms[0]); }\n"
                   "int f(liquid_float_complex *x) { flexframesync q = flexframesync_create(cb, 0); flexframesync_execute(q, x, 256);\n"
                   "  qdetector_cccf d = qdetector_cccf_create_linear(x, 64, LIQUID_FIRFILT_ARKAISER, 2, 7, 0.3f);\n"
                   "  flexframegenprops_s p; flexframegenprops_init_default(&p); p.check = LIQUID_CRC_24; p.fec0 = LIQUID_FEC_CONV_V27;\n"
                   "  return qdetector_cccf_execute(d, x[0]) != 0; }\n")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "shim_c.o")])
