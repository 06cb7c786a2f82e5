"""The DEVICE seeding code against CPython's own `random` -- on the CPU.  `bg_mt_seed_impl` (balatro_gym_amd/csrc/bg_lib.hip: `random.Random(key)` =
init_by_array([key]), balatro_env_2.py:84-106 / shop.py:96) is plain C++ apart from its `__device__` qualifier, so the very text the GPU runs is compiled
here with g++ and held to `random.Random(key).getstate()` (the whole seeded state) and to the stream's first output words (a shop-stream ring slot:
56 tempered words, the top bytes of the first 24 packed into six words -- what a fresh inventory reads --, the seed).  No GPU, no oracle: CPython is the
reference."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "balatro_gym_amd", "csrc")


def _between(text, a, b):
    i = text.index(a)
    return text[i:text.index(b, i)]


@pytest.fixture(scope="module")
def seeder(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    lib = open(os.path.join(CSRC, "bg_lib.hip")).read()
    dev = open(os.path.join(CSRC, "bg_device.h")).read()
    body = _between(lib, "struct alignas(64) BgG16", "__device__ void bg_mt_seed(uint32_t* __restrict__ p")
    temper = _between(dev, "__device__ __forceinline__ uint32_t bg_temper", "// One word / one random() of a lazy MT19937 stream")
    defs = _between(dev, "#define BG_SW_T ", "#define BG_BF_SHOP_OVF")
    d = tmp_path_factory.mktemp("seed_host")
    src = d / "seed_host.cpp"
    src.write_text("""#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#define __device__
#define __forceinline__ inline
#define BG_MT_N 624
#define BG_MT_M 397
struct uint4 { uint32_t x, y, z, w; }; struct uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
""" + defs + temper + body + """
int main(int argc, char** argv) {
  const uint32_t key = (uint32_t)strtoul(argv[2], 0, 10);
  alignas(16) static uint32_t p[640];
  if (argv[1][0] == 's') { bg_mt_seed_impl<true>(p, key); for (int i = 0; i < BG_SLOT_WORDS; i++) printf("%u\\n", p[i]); }
  else { bg_mt_seed_impl<false>(p, key); for (int i = 0; i < 624; i++) printf("%u\\n", p[i]); }
  return 0;
}
""")
    exe = d / "seed_host"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)])
    return str(exe)


KEYS = [0, 1, 7, 42, 382, 12345, 2 ** 31 - 1, 2 ** 31, 4000000000, 2 ** 32 - 1, 1650520237]


@pytest.mark.parametrize("key", KEYS)
def test_seeded_state_is_cpythons(seeder, key):
    got = [int(x) for x in subprocess.check_output([seeder, "f", str(key)]).split()]
    want = list(random.Random(key).getstate()[1][:624])   # the state init_by_array([key]) leaves, before the first twist
    assert got == want


@pytest.mark.parametrize("key", KEYS)
def test_shop_slot_is_the_streams_first_words(seeder, key):
    got = [int(x) for x in subprocess.check_output([seeder, "s", str(key)]).split()]
    r = random.Random(key)
    words = [r.getrandbits(32) for _ in range(56)]
    assert got[:56] == words                                    # BG_SW_T finished (tempered) output words
    pk = [0] * 6
    for k in range(24):
        pk[k >> 2] |= (words[k] >> 24) << (8 * (k & 3))
    assert got[56:62] == pk                                      # BG_SW_PK: the top bytes a fresh inventory classifies (bg_shop_inventory)
    assert got[62] == key and got[63] == 0                       # BG_SW_SEED, padding
