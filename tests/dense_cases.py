"""Dense-convolution equivalents of the spconv v1.x primitives: the independent yardstick for the oracle's
restatement and for lidarseg3d_amd.spconv.  A sparse tensor is densified to [B, C, Z, Y, X] (inactive sites = 0):
  SubMConv3d(k)                 == F.conv3d(x, W, padding=k//2) read at the ACTIVE INPUT sites (outputs elsewhere are dropped);
  SparseConv3d(k, s, p)         == F.conv3d(x, W, stride=s, padding=p) at every site whose receptive field holds an active input,
                                   output sites in ascending (b, z, y, x) order (spconv's CUDA order);
  SparseInverseConv3d           == F.conv_transpose3d(y, W', stride=s, padding=p) read at the forward conv's active input sites, with
                                   W'[ci, co, kz, ky, kx] = W[kz, ky, kx, ci, co] (no flip: input i receives output o through the
                                   offset k with i = o*s - p + k, the pair the forward conv stored).
Weights use spconv's (kD, kH, kW, Cin, Cout) layout with ASYMMETRIC random values, so a wrong offset order, a flipped kernel or a
transposed filter shows up."""
import numpy as np
import torch
import torch.nn.functional as F

CASES = [  # name, kernel, stride, padding, grid (Z, Y, X)
    ("k3_s1_subm", (3, 3, 3), (1, 1, 1), (1, 1, 1), (9, 12, 11)),
    ("k3_s2_p1", (3, 3, 3), (2, 2, 2), (1, 1, 1), (9, 12, 11)),
    ("k3_s2_p011", (3, 3, 3), (2, 2, 2), (0, 1, 1), (9, 12, 11)),
    ("k311_s211_p0", (3, 1, 1), (2, 1, 1), (0, 0, 0), (9, 12, 11)),
    ("k133_subm", (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 10, 12)),
    ("k313_subm", (3, 1, 3), (1, 1, 1), (1, 0, 1), (8, 10, 12)),
    ("k311_subm", (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 10, 12)),
    ("k3_s221_p1", (3, 3, 3), (2, 2, 1), (1, 1, 1), (8, 10, 12)),
    ("k2_s2_p0", (2, 2, 2), (2, 2, 2), (0, 0, 0), (8, 10, 12)),
]


def random_sparse(grid, batch=2, density=0.12, cin=5, seed=0):
    rng = np.random.default_rng(seed)
    occ = rng.uniform(size=(batch,) + tuple(grid)) < density
    occ[0, :2, :3, :3] = True  # a dense corner (full neighbourhoods) next to isolated sites
    coords = np.argwhere(occ).astype(np.int32)
    coords = coords[rng.permutation(coords.shape[0])]  # input order is arbitrary in spconv
    feats = rng.normal(size=(coords.shape[0], cin)).astype(np.float32)
    return coords, feats


def densify(coords, feats, grid, batch):
    c = feats.shape[1]
    x = torch.zeros((batch, c) + tuple(grid), dtype=torch.float64)
    i = torch.from_numpy(coords.astype(np.int64))
    x[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] = torch.from_numpy(feats).double()
    return x


def weight(ks, cin, cout, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=tuple(ks) + (cin, cout)).astype(np.float32)


def torch_weight(w):
    """(kD,kH,kW,Cin,Cout) -> conv3d's (Cout,Cin,kD,kH,kW)"""
    return torch.from_numpy(w).double().permute(4, 3, 0, 1, 2).contiguous()


def dense_subm(coords, feats, w, grid, batch):
    ks = w.shape[:3]
    y = F.conv3d(densify(coords, feats, grid, batch), torch_weight(w), padding=tuple(k // 2 for k in ks))
    i = torch.from_numpy(coords.astype(np.int64))
    return y[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]].numpy()


def dense_conv(coords, feats, w, stride, pad, grid, batch):
    """-> out_coords (ascending linear index), out_feats"""
    x = densify(coords, feats, grid, batch)
    y = F.conv3d(x, torch_weight(w), stride=stride, padding=pad)
    occ = F.conv3d((densify(coords, np.ones((coords.shape[0], 1), np.float32), grid, batch) != 0).double(),
                   torch.ones((1, 1) + tuple(w.shape[:3]), dtype=torch.float64), stride=stride, padding=pad) > 0
    oc = torch.nonzero(occ[:, 0])  # (b, z, y, x) ascending = row-major order
    return oc.numpy().astype(np.int32), y[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]].numpy(), tuple(y.shape[2:])


def dense_inverse(out_coords, out_feats, w_inv, stride, pad, coarse_grid, fine_coords, fine_grid, batch):
    """SparseInverseConv3d with weight w_inv (kD,kH,kW,Cin=channels of out_feats,Cout) back onto fine_coords"""
    y = densify(out_coords, out_feats, coarse_grid, batch)
    wt = torch.from_numpy(w_inv).double().permute(3, 4, 0, 1, 2).contiguous()  # (Cin, Cout, kD, kH, kW)
    opad = [fine_grid[a] - ((coarse_grid[a] - 1) * stride[a] - 2 * pad[a] + w_inv.shape[a]) for a in range(3)]
    assert all(0 <= o < max(stride[a], 1) + 1 for a, o in enumerate(opad)), opad
    x = F.conv_transpose3d(y, wt, stride=stride, padding=pad, output_padding=tuple(min(o, stride[a] - 1) for a, o in enumerate(opad)))
    full = torch.zeros((batch, x.shape[1]) + tuple(fine_grid), dtype=torch.float64)
    sz = [min(full.shape[2 + a], x.shape[2 + a]) for a in range(3)]
    full[:, :, :sz[0], :sz[1], :sz[2]] = x[:, :, :sz[0], :sz[1], :sz[2]]
    i = torch.from_numpy(fine_coords.astype(np.int64))
    return full[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]].numpy()
