#!/usr/bin/env python
"""CPU model of the LDS bank conflicts of the tile-halo convolution's A-fragment reads (ds_read_b128: four groups of 16 lanes, a
lane reads one 16-byte slot; conflict-free iff the 16 lanes of a group hit 16 distinct slots of the 256-byte bank row; lanes reading
the same address broadcast), on the real tile plans of a synthetic 120k frame (numpy restatement of ls3d_tile_plan).
Compares halo layouts / slot orders.  MI355X_MICROARCH.md §LDS."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidarseg3d_amd import synth
from oracle import ref as orc

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[l + 32 for l in g] for g in G128]


def spread(v):
    v = v & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def plan(coords, tbl, slot_order="mask", halo_order="id"):
    n = len(coords)
    key = (spread(coords[:, 2] >> 2) << 1) | spread(coords[:, 3] >> 2)
    order = np.argsort(key, kind="stable")
    tiles = []
    for t0 in range(0, n, 128):
        rows = order[t0:t0 + 128]
        sub = tbl[rows]
        masks = ((sub >= 0) * (1 << np.arange(27))).sum(1)
        if slot_order == "mask":
            o = np.lexsort((np.arange(len(rows)), -masks))
            rows, sub = rows[o], sub[o]
        halo = np.unique(sub[sub >= 0])
        if halo_order == "spatial":  # halo in the tile-key order (then id)
            hk = key[halo]
            halo = halo[np.lexsort((halo, hk))]
        pos = {int(h): i for i, h in enumerate(halo)}
        loc = np.full(sub.shape, -1, np.int64)
        nz = sub >= 0
        loc[nz] = [pos[int(v)] for v in sub[nz]]
        tiles.append((rows, loc, len(halo)))
    return tiles


def cycles(tiles, layout):
    """mean LDS cycles of one A-fragment ds_read_b128 (ideal 4)"""
    tot, cnt = 0, 0
    for rows, loc, H in tiles:
        R = len(rows)
        for w in range(0, R, 32):
            blk = loc[w:w + 32]
            if len(blk) < 32:
                blk = np.concatenate([blk, np.full((32 - len(blk), 27), -1)])
            for k in range(27):
                li = blk[:, k]
                if (li < 0).all():
                    continue
                c = 0
                for g in G128:
                    slots = {}
                    for lane in g:
                        i, kk = lane & 31, lane >> 5
                        l = int(li[i])
                        l = 448 if (l < 0 or l >= 448) else l
                        if layout == "row32":      # [row][32 B]: slot = (2 row + kk) % 16
                            s, addr = (2 * l + kk) % 16, (l, kk)
                        elif layout == "halves":   # [kk][row][16 B]
                            s, addr = (l + kk * 449) % 16, (l, kk)
                        elif layout == "swz":      # [row][32 B] with halves swapped on odd (row >> 3)
                            s, addr = (2 * l + (kk ^ ((l >> 3) & 1))) % 16, (l, kk)
                        slots.setdefault(s, set()).add(addr)
                    c += max(len(v) for v in slots.values())
                tot += c
                cnt += 1
    return tot / max(cnt, 1)


def main():
    cfg = synth.NUSC
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
    frame = synth.lidar_frame(n, seed=100, **cfg)
    v, c, num = orc.hard_voxelize(frame, cfg["voxel_size"], cfg["pc_range"], 5, 300000)
    coords = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    shape = orc.spatial_shape(cfg["voxel_size"], cfg["pc_range"])
    lvl = 1
    specs = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    for spec in [None] + specs:
        if spec is not None:
            coords, shape, _ = orc.conv_rulebook(coords, shape, *spec)
            lvl += 1
        if lvl < int(os.environ.get("MINLVL", "2")):
            continue
        tbl = orc.subm_rulebook(coords, shape, 3)
        for so in ("mask", "spatial"):
            for ho in ("id", "spatial"):
                tiles = plan(coords, tbl, so, ho)[:: max(1, int(os.environ.get("TSTRIDE", "8")))]
                res = {lay: cycles(tiles, lay) for lay in ("row32", "halves", "swz")}
                print("level %d rows %6d slot order %-7s halo order %-7s : LDS cycles per A read " % (lvl, len(coords), so, ho)
                      + "  ".join("%s %.2f" % kv for kv in res.items()), flush=True)


if __name__ == "__main__":
    main()
