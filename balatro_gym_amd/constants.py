"""Phase / Action numbering of the reference's flat 60-way action space (interface data mirrored from
balatro_gym/constants.py:34-39, 43-81, 108-117 so callers can keep `Action.PLAY_HAND` etc.)."""
from enum import IntEnum


class Phase(IntEnum):
    PLAY = 0
    SHOP = 1
    BLIND_SELECT = 2
    PACK_OPEN = 3


class Action(IntEnum):
    PLAY_HAND = 0
    DISCARD = 1
    SELECT_CARD_BASE = 2
    USE_CONSUMABLE_BASE = 10
    SHOP_BUY_BASE = 20
    SHOP_REROLL = 30
    SHOP_END = 31
    SELL_JOKER_BASE = 32
    SELL_CONSUMABLE_BASE = 37
    SELECT_BLIND_BASE = 45
    SKIP_BLIND = 48
    SELECT_FROM_PACK_BASE = 50
    SKIP_PACK = 55


SELECT_CARD_COUNT = 8
USE_CONSUMABLE_COUNT = 5
SHOP_BUY_COUNT = 10
SELL_JOKER_COUNT = 5
SELL_CONSUMABLE_COUNT = 5
SELECT_BLIND_COUNT = 3
SELECT_FROM_PACK_COUNT = 5
ACTION_SPACE_SIZE = 60

HAND_TYPE_NAMES = ["High Card", "One Pair", "Two Pair", "Three Kind", "Straight", "Flush", "Full House", "Four Kind",
                   "Straight Flush", "Five Kind", "Flush House", "Flush Five"]
BOSS_BLIND_NAMES = [None, "The Hook", "The Wall", "The Wheel", "The House", "The Mark", "The Fish", "The Psychic",
                    "The Goad", "The Water", "The Window", "The Manacle", "The Eye", "The Mouth", "The Plant",
                    "The Serpent", "The Pillar", "The Needle", "The Head", "The Club", "The Tooth", "The Flint",
                    "The Oxide", "The Arm", "The Violet", "The Verdant", "The Amber", "The Crimson", "The Cerulean"]
