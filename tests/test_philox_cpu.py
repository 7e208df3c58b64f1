"""csrc/philox.h on the CPU: the text the kernels compile (m5_philox4x32_10, the uint -> float maps, the element -> (call, lane)
geometry of a torch draw) is compiled as host code with g++ and checked against
  * the Random123 known-answer vectors of philox4x32 with 10 rounds (kat_vectors of the Random123 distribution), and
  * a plain-Python restatement of the engine and of ATen's launch geometry (element e of a draw of n values made by G threads:
    Philox call e // 4G of thread e % G, output (e % 4G) // G; counter = offset / 4 + call, subsequence = thread).
The bit-for-bit check against torch.rand / Tensor.exponential_ themselves needs the GPU (tests/test_gpu_kernels.py,
tests/test_gpu_e2e.py); this file pins the arithmetic those tests rely on without one."""
import ctypes
import math
import os
import random
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mars5-tts_amd", "csrc")

HARNESS = r"""
#include <stdint.h>
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = {x, y}; return r; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }
#include "philox.h"
extern "C" void h_philox(const unsigned* c, const unsigned* k, unsigned* out) {
    const uint4 r = m5_philox4x32_10(make_uint4(c[0], c[1], c[2], c[3]), make_uint2(k[0], k[1]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
extern "C" unsigned h_draw_bits(unsigned long long seed, unsigned long long offset, unsigned long long e, unsigned g) {
    return m5_torch_draw_bits(seed, offset, e, g);
}
extern "C" float h_uniform(unsigned v) { return m5_torch_uniform(v); }
extern "C" float h_exponential(unsigned v) { return m5_torch_exponential1(v); }
"""


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("philox")
    src, so = d / "harness.cpp", d / "libphilox_host.so"
    src.write_text(HARNESS)
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", f"-I{CSRC}", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.h_draw_bits.restype = ctypes.c_uint
    lib.h_draw_bits.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_uint]
    lib.h_uniform.restype = ctypes.c_float
    lib.h_uniform.argtypes = [ctypes.c_uint]
    lib.h_exponential.restype = ctypes.c_float
    lib.h_exponential.argtypes = [ctypes.c_uint]
    return lib


M32 = 0xFFFFFFFF


def philox_py(c, k):
    """Philox4x32-10 (Salmon et al., SC'11): ten rounds of two 32 x 32 -> 64 multiplies, key bumped by the Weyl constants."""
    c, k = list(c), list(k)
    for _ in range(10):
        m0, m1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(m1 >> 32) ^ c[1] ^ k[0], m1 & M32, (m0 >> 32) ^ c[3] ^ k[1], m0 & M32]
        k = [(k[0] + 0x9E3779B9) & M32, (k[1] + 0xBB67AE85) & M32]
    return c


def draw_bits_py(seed, offset, e, g):
    call, rem = divmod(e, 4 * g)
    out, thread = divmod(rem, g)
    ctr = offset // 4 + call
    return philox_py([ctr & M32, ctr >> 32, thread, 0], [seed & M32, seed >> 32])[out]


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


KAT = [   # counter (4 words), key (2 words), expected output: Random123 kat_vectors, "philox4x32 10"
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([M32] * 4, [M32] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


@pytest.mark.parametrize("c,k,want", KAT)
def test_philox_known_answer_vectors(host, c, k, want):
    assert philox_py(c, k) == want                       # the restatement itself
    out = (ctypes.c_uint * 4)()
    host.h_philox((ctypes.c_uint * 4)(*c), (ctypes.c_uint * 2)(*k), out)
    assert list(out) == want                             # csrc/philox.h


def test_draw_geometry_equals_the_restatement(host):
    rng = random.Random(5)
    cases = [(0, 0, 0, 256), (1, 4, 255, 256), (7, 8, 4 * 524288 + 3, 524288)]
    for _ in range(400):
        g = 256 * rng.randint(1, 2048)                   # ATen: 256 threads per block, at most CUs x blocks-per-CU blocks
        cases.append((rng.getrandbits(64), 4 * rng.getrandbits(40), rng.randrange(0, 64 * g), g))
    for seed, off, e, g in cases:
        assert host.h_draw_bits(seed, off, e, g) == draw_bits_py(seed, off, e, g), (seed, off, e, g)
    # consecutive elements of one call's four outputs sit G apart; the next call of a thread starts 4G further on
    g = 1024
    quad = philox_py([3, 0, 17, 0], [99, 0])
    assert [host.h_draw_bits(99, 12, 17 + i * g, g) for i in range(4)] == quad
    assert host.h_draw_bits(99, 12, 17 + 4 * g, g) == philox_py([4, 0, 17, 0], [99, 0])[0]


def test_uint_to_float_maps(host):
    two32 = 2.3283064e-10                                # rocrand's 2^-32 literal (fp32: exactly 2^-32)
    assert f32(two32) == 2.0 ** -32
    rng = random.Random(9)
    for v in [0, 1, 2, 0x7FFFFFFF, 0x80000000, 0xFFFFFF00, 0xFFFFFF7F, 0xFFFFFF80, M32] + [rng.getrandbits(32) for _ in range(2000)]:
        u = f32(f32(two32) + f32(f32(float(v)) * f32(two32)))        # (0, 1]: rocrand's uniform_distribution
        want_u = 0.0 if u == 1.0 else u                               # ATen uniform_(0, 1): the closed end moved to 0
        assert host.h_uniform(v) == want_u
        assert 0.0 <= host.h_uniform(v) < 1.0
        ex = host.h_exponential(v)
        if u >= f32(1.0 - 2.0 ** -24):                                # ATen's guard: log(1) = 0 would be a zero rate sample
            assert ex == 2.0 ** -24
        else:
            assert ex > 0.0 and abs(ex - (-math.log(u))) <= 4e-7 * max(1.0, -math.log(u))
    assert host.h_uniform(M32) == 0.0 and host.h_uniform(0) == 2.0 ** -32
