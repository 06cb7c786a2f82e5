"""ctypes binding of libbalatro_mi355x.so (the C ABI in include/balatro_mi355x.h).

The HIP library is the ONLY compute path: loading fails loudly if it is missing or cannot be built, and bg_create
fails if no HIP device is visible.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

OBS_KEYS = [
    "hand", "hand_size", "deck_size", "selected_cards", "chips_scored", "round_chips_scored", "progress_ratio",
    "mult", "chips_needed", "money", "ante", "round", "hands_left", "discards_left", "joker_count", "joker_ids",
    "joker_slots", "consumable_count", "consumables", "consumable_slots", "shop_items", "shop_costs",
    "shop_rerolls", "hand_levels", "phase", "action_mask", "hands_played", "best_hand_this_ante",
    "boss_blind_active", "boss_blind_type", "face_down_cards",
]
# key -> (torch/numpy dtype name, trailing shape) in the reference's dtypes (balatro_env_2.py:1488-1531)
OBS_SPEC = {
    "hand": ("int8", (8,)), "hand_size": ("int8", ()), "deck_size": ("int8", ()), "selected_cards": ("int64", (8,)),
    "chips_scored": ("int64", ()), "round_chips_scored": ("int32", ()), "progress_ratio": ("float32", ()),
    "mult": ("int32", ()), "chips_needed": ("int32", ()), "money": ("int32", ()), "ante": ("int16", ()),
    "round": ("int8", ()), "hands_left": ("int8", ()), "discards_left": ("int8", ()), "joker_count": ("int8", ()),
    "joker_ids": ("int16", (10,)), "joker_slots": ("int8", ()), "consumable_count": ("int8", ()),
    "consumables": ("int16", (5,)), "consumable_slots": ("int8", ()), "shop_items": ("int16", (10,)),
    "shop_costs": ("int16", (10,)), "shop_rerolls": ("int16", ()), "hand_levels": ("int8", (12,)),
    "phase": ("int8", ()), "action_mask": ("int8", (60,)), "hands_played": ("int32", ()),
    "best_hand_this_ante": ("int32", ()), "boss_blind_active": ("int8", ()), "boss_blind_type": ("int8", ()),
    "face_down_cards": ("int64", (8,)),
}
OBS_BYTES = 330
# packed record of bg_rollout_rows (include/balatro_mi355x.h BG_ROW_*): key -> byte offset; reward / action / terminated ride along
ROW_BYTES = 352
ROW_STRIDE_LINES = 384   # BG_RECORD_STRIDE_LINES: records as three whole 128-byte lines (the fast layout of bg_rollout_rows)
ROW_OFFSETS = {
    "selected_cards": 0, "face_down_cards": 64, "chips_scored": 128, "round_chips_scored": 144, "progress_ratio": 148,
    "mult": 152, "chips_needed": 156, "money": 160, "hands_played": 164, "best_hand_this_ante": 168, "action_mask": 176,
    "joker_ids": 236, "shop_items": 256, "shop_costs": 276, "consumables": 296, "ante": 306, "shop_rerolls": 308,
    "hand": 310, "hand_levels": 318, "hand_size": 330, "deck_size": 331, "round": 332, "hands_left": 333,
    "discards_left": 334, "joker_count": 335, "joker_slots": 336, "consumable_count": 337, "consumable_slots": 338,
    "phase": 339, "boss_blind_active": 340, "boss_blind_type": 341,
}
ROW_EXTRA = {"reward": (136, "float64"), "action": (172, "int32"), "terminated": (342, "uint8")}
INFO_KEYS = ["final_score", "error", "flags", "aux", "hand_type", "cards_played", "reward_terms", "score_breakdown"]
INFO_SPEC = {"final_score": ("int64", ()), "error": ("int32", ()), "flags": ("int32", ()), "aux": ("int32", ()),
             "hand_type": ("int8", ()), "cards_played": ("int8", ()), "reward_terms": ("float64", (8,)), "score_breakdown": ("float64", (8,))}

FLAG_SCORER_JOKERS = 1
FLAG_AUTORESET = 2
FLAG_CARD_STATES = 4
POLICY_UNIFORM, POLICY_SMALL_ONLY, POLICY_CYCLE3 = 0, 1, 2

EXPORTS = ["bg_create", "bg_destroy", "bg_last_error", "bg_num_envs", "bg_max_fused_steps", "bg_state_bytes", "bg_seed", "bg_reset",
           "bg_step", "bg_observe", "bg_rollout", "bg_rollout_rows", "bg_inject", "bg_inject_cards", "bg_inject_consumables", "bg_state_blob_bytes", "bg_get_state", "bg_set_state",
           "bg_refill", "bg_check", "bg_set_profiling", "bg_get_profile", "bg_set_max_ante", "bg_inject_deck",
           "bg_classify_batch", "bg_score_hand_batch", "bg_classify_batch_ex", "bg_score_hand_batch_ex", "bg_bench_copy", "bg_bench_fill", "bg_step_many",
           "bg_sim_evaluate_batch", "bg_sim_score_batch", "bg_create_ex", "bg_step_rows", "bg_observe_rows"]
# state-blob geometry (csrc/bg_device.h; tests/test_cabi_and_host.py checks these against the header): 16-byte chunks per env of the
# hot / deck / cold / template arrays, words per stored MT19937 block, words per shop-stream ring slot and where its seed sits
BLOB_NHOT, BLOB_NDECK, BLOB_NCOLD, BLOB_NTMPL, BLOB_NCST, BLOB_MTS, BLOB_SSEED = 8, 4, 7, 2, 7, 640, 128
SHOP_SLOT_WORDS, SHOP_SLOT_SEED_WORD = 64, 62
SCORE_CASE_WORDS, SCORE_OUT_WORDS = 40, 8
SIM_EVAL_BYTES, SIM_CASE_WORDS = 128, 64


class ObsPtrs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in OBS_KEYS]


class InfoPtrs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in INFO_KEYS]


class RolloutStats(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("episodes", C.c_uint64), ("plays", C.c_uint64), ("score_sum", C.c_int64),
                ("reward_bits", C.c_uint64), ("obs_hash", C.c_uint64)]


class NativeError(RuntimeError):
    pass


_lib = None


def lib_path() -> str:
    # BALATRO_MI355X_LIB points at an installed / experimental build of the same C ABI (still the HIP library: there is
    # no other implementation to fall back to)
    return os.environ.get("BALATRO_MI355X_LIB") or _build.LIB


def device_code_signature(path: str | None = None) -> str:
    """Identifies the DEVICE code that runs, reproducibly: the signature the library was built with (`bg_build_signature`: sha256
    prefix over sources + flags + compiler, `build.source_signature()`), which any rebuild of unchanged sources repeats -- the bytes
    of the code object do not (three rebuilds of unchanged sources gave three .hip_fatbin hashes).  bench.py only quotes a committed
    PMC traffic measurement (profiles/*_hbm_traffic.json) taken on exactly this signature.  Ad-hoc builds ("unsigned", e.g.
    tools/build_variant.sh) fall back to the sha256 of their .hip_fatbin section."""
    path = path or lib_path()
    try:
        L = C.CDLL(path) if (_lib is None or path != lib_path()) else _lib
        L.bg_build_signature.restype = C.c_char_p
        sig = L.bg_build_signature().decode()
        if sig and sig != "unsigned":
            return sig
    except (OSError, AttributeError):
        pass
    return fatbin_sha(path)


def fatbin_sha(path: str) -> str:
    import hashlib
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return "not-elf64"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = sh(shstrndx)
    for i in range(shnum):
        name, off, size = sh(i)
        end = data.index(b"\0", stroff + name)
        if data[stroff + name:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()[:16]
    return "no-fatbin"


def load(build_if_missing: bool = True):
    """Load the HIP library; raises NativeError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if build_if_missing and path == _build.LIB and _build.needs_build():
        try:
            _build.build()
        except Exception as exc:  # no hipcc on the box and no prebuilt .so
            if not os.path.exists(path):
                raise NativeError(f"libbalatro_mi355x.so is missing and could not be built: {exc}") from exc
    if not os.path.exists(path):
        raise NativeError(f"{path} not found: build it with `python -m balatro_gym_amd.build` (no CPU fallback exists)")
    # PyTorch-ROCm ships its own libamdhip64 (torch/lib); the process must hold ONE HIP runtime, and device tensors come
    # from torch, so torch's copy has to be the one that is resident when this library's libamdhip64.so.7 dependency is
    # resolved.  (Loading this library first brought in /opt/rocm's runtime as well: bg_create then saw no device.)
    import torch  # noqa: F401
    try:
        L = C.CDLL(path)
    except OSError as exc:
        raise NativeError(f"cannot load {path}: {exc}") from exc
    for name in EXPORTS:
        if not hasattr(L, name):
            raise NativeError(f"{path} does not export {name}")
    vp, i32, u32, u64, i64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_int64
    L.bg_create.argtypes = [i32, i32, u32, i32, C.POINTER(vp)]
    L.bg_create_ex.argtypes = [i32, i32, u32, i32, i32, C.POINTER(vp)]
    L.bg_destroy.argtypes = [vp]
    L.bg_last_error.restype = C.c_char_p
    L.bg_last_error.argtypes = [vp]
    L.bg_num_envs.argtypes = [vp]
    L.bg_max_fused_steps.argtypes = [vp]
    L.bg_state_bytes.restype = u64
    L.bg_state_bytes.argtypes = [vp]
    L.bg_seed.argtypes = [vp, vp, vp, i32, vp]
    L.bg_reset.argtypes = [vp, vp, C.POINTER(ObsPtrs), vp]
    L.bg_step.argtypes = [vp, vp, C.POINTER(ObsPtrs), vp, vp, vp, C.POINTER(InfoPtrs), vp]
    L.bg_observe.argtypes = [vp, C.POINTER(ObsPtrs), vp]
    L.bg_step_rows.argtypes = [vp, vp, vp, u64, vp, vp, vp, C.POINTER(InfoPtrs), vp]
    L.bg_observe_rows.argtypes = [vp, vp, u64, vp]
    L.bg_step_many.argtypes = [vp, i32, vp, C.POINTER(ObsPtrs), i32, vp, vp, vp, C.POINTER(InfoPtrs), vp]
    L.bg_rollout.argtypes = [vp, i32, i32, u64, u64, u64, C.POINTER(ObsPtrs), i32, vp, vp, vp, vp, vp]
    L.bg_rollout_rows.argtypes = [vp, i32, i32, u64, u64, u64, vp, u64, i32, vp, vp]
    L.bg_set_gather_peers.argtypes = [vp, vp, i32, i32]
    L.bg_inject.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.bg_inject_cards.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    L.bg_inject_consumables.argtypes = [vp, vp, vp, vp, i32, vp]
    L.bg_state_blob_bytes.restype = u64
    L.bg_state_blob_bytes.argtypes = [vp]
    L.bg_get_state.argtypes = [vp, i32, vp, u64]
    L.bg_set_state.argtypes = [vp, i32, vp, u64]
    L.bg_refill.argtypes = [vp, vp]
    L.bg_check.argtypes = [vp, vp]
    L.bg_set_profiling.argtypes = [vp, i32]
    L.bg_get_profile.argtypes = [vp, C.POINTER(C.c_double)]
    L.bg_set_max_ante.argtypes = [vp, i32, vp, vp, vp]
    L.bg_inject_deck.argtypes = [vp, vp, vp, vp]
    L.bg_classify_batch.argtypes = [vp, vp, vp, i64, vp]
    L.bg_score_hand_batch.argtypes = [vp, vp, i32, vp]
    L.bg_classify_batch_ex.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.bg_score_hand_batch_ex.argtypes = [vp, vp, i32, i32, vp, vp]
    L.bg_sim_evaluate_batch.argtypes = [vp, vp, vp, vp, i32, vp]
    L.bg_sim_score_batch.argtypes = [vp, vp, i32, vp]
    L.bg_bench_copy.argtypes = [vp, vp, u64, i32, C.POINTER(C.c_double), vp]
    L.bg_bench_fill.argtypes = [vp, u64, i32, C.POINTER(C.c_double), vp]
    _lib = L
    return L
