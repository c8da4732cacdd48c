"""Enums with the reference's integer values (they index its dispatch tables).

BlockSize   src/partition.rs:130-153      TxSize  src/transform/mod.rs:101-123
TxType      src/transform/mod.rs:56-74    FilterMode  src/mc.rs:100-106
"""
import enum


class BlockSize(enum.IntEnum):
    BLOCK_4X4 = 0
    BLOCK_4X8 = 1
    BLOCK_8X4 = 2
    BLOCK_8X8 = 3
    BLOCK_8X16 = 4
    BLOCK_16X8 = 5
    BLOCK_16X16 = 6
    BLOCK_16X32 = 7
    BLOCK_32X16 = 8
    BLOCK_32X32 = 9
    BLOCK_32X64 = 10
    BLOCK_64X32 = 11
    BLOCK_64X64 = 12
    BLOCK_64X128 = 13
    BLOCK_128X64 = 14
    BLOCK_128X128 = 15
    BLOCK_4X16 = 16
    BLOCK_16X4 = 17
    BLOCK_8X32 = 18
    BLOCK_32X8 = 19
    BLOCK_16X64 = 20
    BLOCK_64X16 = 21

    @property
    def dims(self):
        w, h = self.name.split("_")[1].split("X")
        return int(w), int(h)


class TxSize(enum.IntEnum):
    TX_4X4 = 0
    TX_8X8 = 1
    TX_16X16 = 2
    TX_32X32 = 3
    TX_64X64 = 4
    TX_4X8 = 5
    TX_8X4 = 6
    TX_8X16 = 7
    TX_16X8 = 8
    TX_16X32 = 9
    TX_32X16 = 10
    TX_32X64 = 11
    TX_64X32 = 12
    TX_4X16 = 13
    TX_16X4 = 14
    TX_8X32 = 15
    TX_32X8 = 16
    TX_16X64 = 17
    TX_64X16 = 18

    @property
    def dims(self):
        w, h = self.name.split("_")[1].split("X")
        return int(w), int(h)

    @staticmethod
    def by_dims(w, h):
        return TxSize["TX_%dX%d" % (w, h)]


TX_DIMS = [t.dims for t in TxSize]


class TxType(enum.IntEnum):
    DCT_DCT = 0
    ADST_DCT = 1
    DCT_ADST = 2
    ADST_ADST = 3
    FLIPADST_DCT = 4
    DCT_FLIPADST = 5
    FLIPADST_FLIPADST = 6
    ADST_FLIPADST = 7
    FLIPADST_ADST = 8
    IDTX = 9
    V_DCT = 10
    H_DCT = 11
    V_ADST = 12
    H_ADST = 13
    V_FLIPADST = 14
    H_FLIPADST = 15
    WHT_WHT = 16


class FilterMode(enum.IntEnum):
    REGULAR = 0
    SMOOTH = 1
    SHARP = 2
    BILINEAR = 3


class DistKind(enum.IntEnum):
    SAD = 0
    SATD = 1


def valid_av1_transform(tx_size, tx_type):
    """src/transform/mod.rs:405-417 (+ WHT exists only at 4x4)."""
    w, h = TX_DIMS[int(tx_size)]
    m = max(w, h)
    if int(tx_type) == 16:
        return (w, h) == (4, 4)
    if m == 64:
        return int(tx_type) == 0
    if m == 32:
        return int(tx_type) in (0, 9)
    return True
