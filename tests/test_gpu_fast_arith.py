"""The kernel's shortened float3::Normalized (csrc/srt_kernel.hip.h: sqrt and three divisions without the rescaling / fix-up
steps of the library expansions, where those are identities) against the library path, bit for bit, on the device:
srt_selftest_arith draws vectors of every kind — moderate magnitudes (the short path runs), any bit pattern, every exponent,
zeros, denormals, infinities, NaNs (the wave-uniform decision sends those down the library path)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 0xC0FFEE])
def test_short_normalize_equals_library_normalize(srt, seed):
    L = srt.load_library()
    bad = C.c_uint64(123)
    assert L.srt_selftest_arith(0, seed, 1 << 28, C.byref(bad)) == 0
    assert bad.value == 0


def test_short_square_root_equals_library_square_root_for_every_float_in_its_window(srt):
    """2^31 vectors: vector i also puts the float with bit pattern i through sqrt_window and sqrtf, if it lies in the window
    (+0, [2^-96, inf), NaNs) — every non-negative float there is."""
    L = srt.load_library()
    bad = C.c_uint64(123)
    assert L.srt_selftest_arith(0, 7, 1 << 31, C.byref(bad)) == 0
    assert bad.value == 0


def test_selftest_argument_checks(srt):
    L = srt.load_library()
    bad = C.c_uint64(0)
    assert L.srt_selftest_arith(0, 1, 0, C.byref(bad)) != 0
    assert L.srt_selftest_arith(0, 1, 16, None) != 0
    assert L.srt_selftest_arith(99, 1, 16, C.byref(bad)) != 0
