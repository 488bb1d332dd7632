#!/usr/bin/env python
"""How much of the tile kernel's matrix work multiplies absent neighbours, and what any re-grouping of a tile's rows could recover (CPU only).

VERDICT r5 item 2 asked for executed / useful MFMA work 1.57x -> <= 1.25x by (a) 16-row MFMA granularity, (b) ordering a tile's rows so that a
wave's rows share their empty offsets, (c) bigger workgroups.  This script answers (a) and (b) on the bench frame WITHOUT a GPU: it builds the SubM
rulebooks of levels 2 - 4 of the 120 000-point frame with the oracle (the same sites the product computes: bit-exact, tests), cuts them into the
kernel's tiles (Morton order of 4 x 4 (y, x) columns, 128 rows) and counts, per level, the (row group, offset) blocks that hold at least one
pair, for groups of 128 (what the pipelined loop executes: it skips an offset only when the whole tile lacks it), 32 (per-wave skipping), 16
(v_mfma_f32_16x16x32_bf16) and 8 rows, with the rows of a tile in spatial order, sorted by neighbour mask (what k_tile_build does) and
greedily clustered by mask (an upper bound on what any within-tile order can give).

    python tools/analyse_blocks.py [--points 120000] [--seed 100] > profiles/round6_block_analysis.txt"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarseg3d_amd import synth  # noqa: E402
from oracle import ref as orc  # noqa: E402  (test infrastructure: this is an offline analysis, not the product path)


def morton2(y, x):
    def part(v):
        v = v.astype(np.uint64)
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return part(x) | (part(y) << 1)


def blocks(tiles, g):
    """rows executed = active (group of g rows, offset) blocks x g"""
    t, tr, k = tiles.shape
    return int(tiles.reshape(t, tr // g, g, k).any(2).sum()) * g


def greedy(rows, g):
    """groups of g rows grown from the densest remaining row by smallest growth of the union of offsets -> rows executed"""
    left = list(range(rows.shape[0]))
    pc = rows.sum(1)
    cost = 0
    while left:
        seed = max(left, key=lambda i: pc[i])
        left.remove(seed)
        u, n = rows[seed].copy(), 1
        while n < g and left:
            arr = np.array(left)
            j = int(arr[np.argmin((rows[arr] & ~u).sum(1))])
            left.remove(j)
            u |= rows[j]
            n += 1
        cost += int(u.sum()) * g
    return cost


def analyse(name, coords, nbr, tr=128, sample=48):
    v = coords.shape[0]
    order = np.argsort(morton2(coords[:, 2] >> 2, coords[:, 3] >> 2), kind="stable")
    m = nbr[order] >= 0
    pairs = int(m.sum())
    nt = (v + tr - 1) // tr
    tiles = np.concatenate([m, np.zeros((nt * tr - v, m.shape[1]), bool)]).reshape(nt, tr, m.shape[1])
    mval = (tiles * (1 << np.arange(m.shape[1], dtype=np.int64))).sum(2)
    by_mask = np.take_along_axis(tiles, np.argsort(-mval, axis=1, kind="stable")[:, :, None], 1)
    print("%s: %d rows, %d tiles, %d pairs, pair density %.3f (%.2f neighbours per row)" % (name, v, nt, pairs, m.mean(), m.sum(1).mean()))
    print("   rows x offsets executed / pairs:")
    print("   %-52s %6.3f" % ("tile-level skip only (the pipelined loop, 128 rows)", blocks(tiles, tr) / pairs))
    for g in (32, 16, 8):
        print("   %-52s %6.3f   spatial order %6.3f" % ("groups of %d rows, rows sorted by mask" % g, blocks(by_mask, g) / pairs, blocks(tiles, g) / pairs))
    rng = np.random.default_rng(0)
    pick = rng.choice(nt - 1, size=min(sample, nt - 1), replace=False)
    for g in (32, 16):
        base = sum(blocks(by_mask[t:t + 1], g) for t in pick)
        gr = sum(greedy(tiles[t][tiles[t].any(1)], g) for t in pick)
        print("   %-52s %6.3f x the mask-sorted order (%d tiles sampled)" % ("greedy clustering by mask, groups of %d" % g, gr / base, len(pick)))
    # what a row looks like: a random subset of its 27 offsets?  if the offsets of a row were independent with the level's density p, a group
    # of g rows would miss an offset with probability (1 - p)^g
    p = m.mean()
    print("   independent-offsets model: an offset is empty for a whole group of 32 / 16 / 8 rows with probability %.1e / %.1e / %.3f"
          % ((1 - p) ** 32, (1 - p) ** 16, (1 - p) ** 8))
    return pairs, blocks(tiles, tr), blocks(by_mask, 32), blocks(by_mask, 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--seed", type=int, default=100)
    a = ap.parse_args()
    cfg = synth.NUSC
    pts = synth.lidar_frame(a.points, seed=a.seed, **cfg)
    coords = orc.hard_voxelize(pts, cfg["voxel_size"], cfg["pc_range"], 5, 300000)[1]
    c = np.concatenate([np.zeros((coords.shape[0], 1), np.int32), coords], 1)
    shape = orc.spatial_shape(cfg["voxel_size"], cfg["pc_range"])
    levels = [("level 1", c, shape)]
    for pad in (1, 1, (0, 1, 1)):
        c, shape, _ = orc.conv_rulebook(c, shape, 3, 2, pad)
        levels.append(("level %d" % (len(levels) + 1), c, shape))
    cin = {"level 2": 64, "level 3": 128, "level 4": 128}
    tot = np.zeros(4)
    for name, cc, ss in levels[1:]:
        r = analyse(name, cc, orc.subm_rulebook(cc, ss, 3))
        tot += np.array(r, np.float64) * cin[name] ** 2  # flop weight of a 3x3x3 SubM layer of the level (6 of them per level)
    print("flop-weighted over levels 2 - 4 (6 equal layers each): executed / useful = %.3f tile-level, %.3f per wave (32), %.3f per 16 rows"
          % (tot[1] / tot[0], tot[2] / tot[0], tot[3] / tot[0]))


if __name__ == "__main__":
    main()
