"""Host-side model of the LDS bank conflicts of conv3x3_f16dma_kernel's A-fragment reads (ds_read_b128: four groups of 16 lanes, 64 banks of 4 B;
a group is conflict-free iff its 16 addresses fall into 16 different 16-byte bank quads -- MI355X_MICROARCH.md, LDS table).  The halo is stored
pixel-major, 128 B per pixel, chunk slot = chunk ^ swizzle(pixel).  With swizzle = (hp >> 1) & 7 on the halo-pixel index hp (rounds 3 / 4 until
HISTORY E.21) the bank quad of a read is a bijection of hp mod 16, so a group is conflict-free iff its lanes' halo-pixel indices are distinct mod
16: true on 64- and 32-column images (32 lanes = one image row), false on 16-column images (two rows per 32 lanes, row pitch 18 = 2 mod 16: +1
cycle in EVERY group, 4 -> 5) and on 8-column images (four rows, pitch 10: +2, 4 -> 6).  Matches the SQ counters of that build
(profiles/r4_*_fp16_sq_counters.json: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.25 - 0.29 on the <16, .> instantiations, 0.43 on <8, .>, 0.00
on <32, .> / <64, .>).  No XOR of the chunk bits with a per-row constant removes it (searched below: the colliding bit is the pixel's parity); the
swizzle on the COLUMN of the halo pixel that the kernel uses now, ((x + 8 (row & 1)) >> 1) & 7 on 8-column and (x >> 1) & 7 on 16-column images,
does.      python tools/probes/halo_bank_conflicts.py          (tests/test_host_logic.py imports conflicts / swizzle_* from here)"""
import itertools

G0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]       # the lane groups of one half-wave of a ds_read_b128
G1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]


def geo(W):
    """GeoD<W> of csrc/conv3x3_f16dma.hip: image slots per 256-pixel tile, tile rows, halo row pitch, halo rows per slot."""
    nimg = 1 if W * W >= 256 else 256 // (W * W)
    th = 256 // (W * nimg)
    return nimg, th, W + 2, th + 2


def swizzle_by_index(W, row, col, hp):
    return (hp >> 1) & 7


def swizzle_by_column(W, row, col, hp):
    """What the kernel does (a_addr / halo_dma): by the halo-pixel index on 32- / 64-column images, by the column (+ 8 x row parity) below."""
    if W > 16:
        return (hp >> 1) & 7
    return ((col + (8 * (row & 1) if W == 8 else 0)) >> 1) & 7


def conflicts(W, swizzle, pitch=None):
    """(extra LDS cycles, lane groups) over every wave row (4), 32-row block (2), tap (9) and lane group (2) of a tile."""
    nimg, th, wp, hp_rows = geo(W)
    wp = pitch or wp
    extra = total = 0
    for wr in range(4):
        for i in range(2):
            for tt in range(9):
                for g in (G0, G1):
                    quads = {}
                    for lane in g:
                        m = wr * 64 + i * 32 + lane
                        sl, rem = divmod(m, th * W)
                        r, c = divmod(rem, W)
                        row, col = sl * hp_rows + r + tt // 3, c + tt % 3
                        hp = row * wp + col
                        a16 = hp * 8 + swizzle(W, row, col, hp)           # address / 16 of the lane's chunk 0 (K steps and the half-wave XOR constants)
                        quads.setdefault(a16 % 16, set()).add(a16)
                    extra += max(len(v) for v in quads.values()) - 1
                    total += 1
    return extra, total


if __name__ == '__main__':
    for W in (64, 32, 16, 8):
        print(W, 'swizzle on the pixel index: (extra cycles, groups)', conflicts(W, swizzle_by_index), ' on the column:', conflicts(W, swizzle_by_column), flush=True)
    for W, mask in ((16, 1), (16, 3), (8, 1), (8, 3)):
        best = None
        for tab in itertools.product(range(8), repeat=mask + 1):
            if tab[0] != 0:
                continue
            e, _ = conflicts(W, lambda W_, row, col, hp, tab=tab, mask=mask: ((hp >> 1) & 7) ^ tab[row & mask])
            if best is None or e < best[0]:
                best = (e, tab)
        print(W, 'index swizzle ^ per-row constant, table of', mask + 1, ': best', best, flush=True)
    for W in (16, 8):
        ok = [p for p in range(W + 2, W + 18) if conflicts(W, swizzle_by_index, p)[0] == 0]
        print(W, 'halo row pitches (pixels) that would make the index swizzle conflict-free:', ok, flush=True)
