"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/balatro_mi355x.h declares, fails loudly
without a HIP device, and the host-side pieces (observation layout, sharding) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "balatro_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bg_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from balatro_gym_amd import _native as nat
    L = nat.load()
    syms = _declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/balatro_mi355x.h but not exported"
    assert set(nat.EXPORTS) <= set(syms)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from balatro_gym_amd import _native as nat
    L = nat.load()
    h = C.c_void_p()
    rc = L.bg_create(4, 0, 0, 0, C.byref(h))
    assert rc != 0 and b"no HIP device" in L.bg_last_error(None)
    from balatro_gym_amd import BalatroVecEnv
    with pytest.raises(nat.NativeError):
        BalatroVecEnv(4, [1, 2, 3, 4])


def test_product_does_not_touch_oracle():
    """The shipped package must never import / link the checker (oracle/)."""
    pkg = os.path.join(ROOT, "balatro_gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_obs_layout_matches_reference_dtypes():
    import torch
    from balatro_gym_amd import _native as nat
    from balatro_gym_amd.vec_env import ObsBuffers
    from tests.helpers import load_trace
    ob = ObsBuffers(5, torch.device("cpu"))
    tr = load_trace("c1_small_only")
    total = 0
    for k in nat.OBS_KEYS:
        t = ob.tensors[k]
        ref = tr["obs0_" + k]
        assert tuple(t.shape[1:]) == ref.shape[1:], k
        assert str(t.dtype).replace("torch.", "") == str(ref.dtype), k
        total += t[0].numel() * t.element_size()
    assert total == nat.OBS_BYTES == 330
    ob3 = ObsBuffers(5, torch.device("cpu"), steps=3)
    assert ob3.tensors["action_mask"].shape == (3, 5, 60)


def test_shard_range_partitions_everything():
    from balatro_gym_amd.sharded import shard_range
    for total in (1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(total, world, r)
                cover.extend(range(lo, hi))
            assert cover == list(range(total))


def test_sb3_fixed_observation_space_and_transform():
    """BalatroEnvFixed's observation fixes (train_balatro_fixed.py:29-207), batched: 51 keys, scalars -> (1,), MultiBinary
    -> int8, int16 array keys of the upgrade list -> int32, never-produced keys -> zeros of the declared shape."""
    import torch
    from balatro_gym_amd import _native as nat
    from balatro_gym_amd.sb3_adapter import FIXED_SPEC, fix_observation
    assert len(FIXED_SPEC) == 51 and list(FIXED_SPEC)[:31] == nat.OBS_KEYS
    assert FIXED_SPEC["ante"] == ("int16", (1,), True)              # scalar first: dtype kept
    assert FIXED_SPEC["chips_scored"] == ("int64", (1,), True)
    assert FIXED_SPEC["joker_ids"] == ("int32", (10,), True)        # int upgrade
    assert FIXED_SPEC["shop_costs"][0] == "int32" and FIXED_SPEC["shop_items"][0] == "int32"
    assert FIXED_SPEC["consumables"] == ("int16", (5,), True)       # not in the upgrade list
    assert FIXED_SPEC["selected_cards"] == ("int8", (8,), True) and FIXED_SPEC["action_mask"] == ("int8", (60,), True)
    assert FIXED_SPEC["hand_one_hot"] == ("float32", (8, 52), False) and FIXED_SPEC["win_probability"] == ("float32", (1,), False)
    n = 5
    tdt = {"int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64, "float32": torch.float32}
    obs = {}
    for k in nat.OBS_KEYS:
        dt, shape = nat.OBS_SPEC[k]
        obs[k] = (torch.arange(n * int(np.prod(shape, dtype=np.int64))).reshape((n,) + tuple(shape)) % 3).to(tdt[dt])
    fixed = fix_observation(obs)
    assert set(fixed) == set(FIXED_SPEC)
    for k, (dt, shape, produced) in FIXED_SPEC.items():
        assert tuple(fixed[k].shape) == (n,) + shape and fixed[k].dtype == tdt[dt], k
        if produced:
            assert torch.equal(fixed[k].reshape(n, -1).to(torch.int64), obs[k].reshape(n, -1).to(torch.int64)), k
        else:
            assert not fixed[k].any()


def test_sb3_fixed_space_matches_reference_fixture():
    """FIXED_SPEC (the adapter's statement of BalatroEnvFixed's observation space) key by key against the space the reference's
    own BalatroEnvFixed built (tests/golden/sb3_fixed.npz, generated by oracle/gen_golden.py gen_sb3_fixed from
    train_balatro_fixed.py:20-124): same 51 keys in the same order, same dtypes, same shapes."""
    from balatro_gym_amd.sb3_adapter import FIXED_SPEC
    from tests.helpers import GOLD
    with np.load(os.path.join(GOLD, "sb3_fixed.npz")) as z:
        keys, dtypes, shapes = [str(k) for k in z["keys"]], [str(d) for d in z["dtypes"]], [str(x) for x in z["shapes"]]
    assert list(FIXED_SPEC) == keys
    for k, dt, sh in zip(keys, dtypes, shapes):
        want_shape = tuple(int(x) for x in sh.split(",") if x)
        assert FIXED_SPEC[k][0] == dt and FIXED_SPEC[k][1] == want_shape, (k, FIXED_SPEC[k], dt, want_shape)


def test_drop_in_observation_space_declares_all_51_keys():
    """`BalatroEnv.observation_space` is the reference's `_create_observation_space` (balatro_env_2.py:386-470): 51 keys -- the 31 that
    `_get_observation` fills followed by the 20 it never does -- so that the reference's own wrapper, which walks
    `self.env.observation_space.spaces.items()` (train_balatro_fixed.py:31-75), builds the SAME fixed space over this drop-in as over
    the reference env.  The wrapper's rule is applied here to `make_observation_space()` and compared, key by key, with the space the
    reference's `BalatroEnvFixed(BalatroEnv)` built (tests/golden/sb3_fixed.npz)."""
    from balatro_gym_amd import _native as nat
    from balatro_gym_amd.env import make_observation_space, _DECLARED_ONLY, _spaces
    from tests.helpers import GOLD
    space = make_observation_space()
    assert len(space.spaces) == 51 and list(space.spaces)[:31] == nat.OBS_KEYS and list(space.spaces)[31:] == list(_DECLARED_ONLY)
    upgrade = {"chips_scored", "round_chips_scored", "chips_needed", "shop_costs", "shop_items", "joker_ids", "shop_rerolls",
               "hand_potential_scores", "best_hand_this_ante", "money", "hands_played", "ante", "round_chips_scored_rank"}
    fixed = {}
    for k, sp in space.spaces.items():                      # train_balatro_fixed.py:31-100
        if isinstance(sp, _spaces.MultiBinary):
            fixed[k] = ("int8", (int(sp.n),))               # MultiBinary -> Box(int8)
        else:
            dt, shape = np.dtype(sp.dtype), tuple(sp.shape)
            if shape == ():
                fixed[k] = (dt.name, (1,))                  # scalar -> (1,), dtype kept
            elif dt in (np.dtype(np.int16), np.dtype(np.int8)) and k in upgrade:
                fixed[k] = ("int32", shape)                 # int upgrade of the array keys on the list
            else:
                fixed[k] = (dt.name, shape)
    with np.load(os.path.join(GOLD, "sb3_fixed.npz")) as z:
        keys, dtypes, shapes = [str(k) for k in z["keys"]], [str(d) for d in z["dtypes"]], [str(x) for x in z["shapes"]]
    assert list(fixed) == keys
    for k, dt, sh in zip(keys, dtypes, shapes):
        assert fixed[k] == (dt, tuple(int(x) for x in sh.split(",") if x)), (k, fixed[k], dt, sh)
    # bounds of the declared-only keys as the reference states them (:439-468)
    assert space.spaces["hand_one_hot"].shape == (8, 52) and np.dtype(space.spaces["hand_one_hot"].dtype) == np.float32
    assert space.spaces["hand_potential_scores"].shape == (12,) and np.dtype(space.spaces["hand_potential_scores"].dtype) == np.int32
    assert float(np.max(space.spaces["avg_score_per_hand"].high)) == 10000.0 and float(np.max(space.spaces["suit_counts"].high)) == 8.0


def test_integration_snippet_structs_match_the_header():
    """INTEGRATION.md's reference-side ctypes binding declares `InfoPtrs` / `ObsPtrs` with exactly the members of the header's
    `bg_info_ptrs` / `bg_obs_ptrs`, in order (a struct that is one pointer short hands the library a garbage last member)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "balatro_mi355x.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()

    def members(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return re.findall(r"\*\s*(\w+)\s*;", body)

    info_doc = re.search(r"class InfoPtrs\(C\.Structure\).*?\[(\"final_score\".*?)\]\]", doc, re.S).group(1)
    assert re.findall(r'"(\w+)"', info_doc) == members("bg_info_ptrs")
    obs_doc = re.search(r"OBS_KEYS = \[(.*?)\]", doc, re.S).group(1)
    assert re.findall(r'"(\w+)"', obs_doc) == members("bg_obs_ptrs")


def test_packed_record_layout_matches_header():
    """The record offsets the Python views use are the BG_ROW_* constants of include/balatro_mi355x.h, every key is
    naturally aligned and no two fields overlap."""
    import re
    from balatro_gym_amd import _native as nat
    text = open(os.path.join(ROOT, "include", "balatro_mi355x.h")).read()
    hdr = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define BG_ROW_([A-Z_0-9]+)\s+(\d+)", text)}
    assert hdr.pop("bytes") == nat.ROW_BYTES == 352
    offs = dict(nat.ROW_OFFSETS)
    offs.update({k: v[0] for k, v in nat.ROW_EXTRA.items()})
    assert offs == hdr
    spans = []
    for k, off in offs.items():
        dt, shape = nat.OBS_SPEC[k] if k in nat.OBS_SPEC else (nat.ROW_EXTRA[k][1], ())
        item = np.dtype(dt).itemsize
        size = item * int(np.prod(shape, dtype=np.int64))
        assert off % item == 0, k
        spans.append((off, off + size, k))
    spans.sort()
    for (a0, a1, ka), (b0, b1, kb) in zip(spans, spans[1:]):
        assert a1 <= b0, (ka, kb)
    assert spans[-1][1] <= nat.ROW_BYTES


def test_bench_gpus_flag_without_devices_fails_cleanly():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts the ranks itself; with fewer than 2 visible GPUs (none in the build
    container) it must say so and exit 2 -- not die on an assertion, not re-exec."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the flag would really start two ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 2, (out.returncode, out.stderr[-500:])
    assert "GPU(s) are visible" in out.stderr and "Traceback" not in out.stderr


def test_blob_geometry_constants_match_the_device_header():
    """balatro_gym_amd._native's state-blob constants are the device header's (csrc/bg_device.h)."""
    from balatro_gym_amd import _native as nat
    text = open(os.path.join(ROOT, "balatro_gym_amd", "csrc", "bg_device.h")).read()

    def define(name):
        m = re.search(rf"^#define {name} (\S+)", text, flags=re.M)
        assert m, name
        return m.group(1)
    assert int(define("BG_NHOT")) == nat.BLOB_NHOT and int(define("BG_NDECK")) == nat.BLOB_NDECK
    assert int(define("BG_NCOLD")) == nat.BLOB_NCOLD and int(define("BG_NTMPL")) == nat.BLOB_NTMPL
    assert int(define("BG_MTS")) == nat.BLOB_MTS and int(define("BG_NCST")) == nat.BLOB_NCST
    assert nat.SHOP_SLOT_SEED_WORD == int(define("BG_SW_SEED")) and nat.SHOP_SLOT_WORDS == int(define("BG_SLOT_WORDS"))
    m = re.search(r"^#define BG_SSEED (\d+)", open(os.path.join(ROOT, "balatro_gym_amd", "csrc", "bg_device.h")).read() +
                  open(os.path.join(ROOT, "balatro_gym_amd", "csrc", "bg_lib.hip")).read(), flags=re.M)
    assert m and int(m.group(1)) == nat.BLOB_SSEED


def test_operator_wrappers_validate_every_tensor():
    """classify_batch / sim_evaluate_batch refuse a length-mismatched or wrong-device `n` / `flags` before any pointer reaches the
    library (a host pointer or a short buffer would be a GPU fault, not an exception)."""
    import torch
    from balatro_gym_amd import vec_env as ve
    cards = torch.zeros((4, 8), dtype=torch.uint8)
    with pytest.raises(ValueError):
        ve.classify_batch(cards, torch.zeros(4, dtype=torch.uint8))           # not on a device at all
    src = open(os.path.join(ROOT, "balatro_gym_amd", "vec_env.py")).read()
    assert "n.device != cards.device" in src and "tuple(n.shape) != (cards.shape[0],)" in src
    assert "n.to(device=hands.device, dtype=torch.int32)" in src and "flags.to(device=hands.device, dtype=torch.int32)" in src


def test_rollout_refuses_per_step_outputs_without_per_step_buffers():
    """bg_rollout writes reward / terminated / actions with the OBSERVATION's row stride: without per-step observation buffers every step lands in
    row 0.  A [T, N] tensor there would come back with one filled row (bench.py's `step_path` replayed such rows of zeros until round 6): the wrapper refuses."""
    import types
    import torch
    from balatro_gym_amd.vec_env import BalatroVecEnv
    fake = types.SimpleNamespace(_obs=None)
    for kw in ("reward", "terminated", "actions"):
        with pytest.raises(ValueError, match="row 0"):
            BalatroVecEnv.rollout(fake, 8, **{kw: torch.zeros((8, 4), dtype=torch.int32)})


def test_build_signature_is_reproducible(tmp_path):
    """The identity of the device code (`bg_build_signature`, what profiles/*_hbm_traffic.json are keyed on) must survive a rebuild of
    unchanged sources: the library reports the sha256 prefix of sources + flags (12 digits) and compiler (4 digits) it was built from, and
    a second build -- whose code-object bytes differ -- reports the same.  The in-tree library is only held to the compiler-independent
    part (it may have been built on a box with another ROCm); without hipcc there is nothing to rebuild."""
    import ctypes as C
    import shutil
    from balatro_gym_amd import _native as nat, build
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available: nothing to rebuild")
    want = build.source_signature()
    assert want == build.source_signature() and len(want) == 16
    assert build.sources_part(nat.device_code_signature()) == build.sources_part(want), \
        "the in-tree library was not built from the sources in the tree (python -m balatro_gym_amd.build --force)"
    other = build.build(force=True, out=str(tmp_path / "rebuilt.so"))
    L = C.CDLL(other)
    L.bg_build_signature.restype = C.c_char_p
    assert L.bg_build_signature().decode() == want
    assert nat.device_code_signature(other) == want
    assert build.library_signature(other) == want          # read out of the file, without loading it (what needs_build compares)
    assert build.library_signature(__file__) is None
