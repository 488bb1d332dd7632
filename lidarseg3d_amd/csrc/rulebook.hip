// rulebook.hip — coordinate index and rulebooks for SubMConv3d / SparseConv3d / SparseInverseConv3d.
//
// spconv v1.x (third party, pinned @ fad3000 by docs/INSTALL.md:88-99; call sites
// det3d/models/backbones/scn_unet.py:15-24,205) builds per-offset (in,out) PAIR lists and runs
// gather -> GEMM -> scatter-add per offset.  Here the rulebook is OUTPUT-MAJOR instead: one row of
// `kvol` int32 per output site naming the input row that feeds it through each kernel offset (-1 = none).
// That is what an output-stationary gather-GEMM wants (no atomics, no zero-init, fused epilogue), and the
// same table shape serves all three conv types:
//     SubM      nbr[v][k]      = row of coords[v] + k - ksize/2
//     strided   nbr_out[o][k]  = row of o*stride - pad + k
//     inverse   nbr_inv[i][k]  = output row o with o*stride - pad + k == i     (transposed relation)
// Strided-conv output sites come out in ascending linear index (spconv's CUDA ordering) without a sort:
// candidate sites set bits in a dense bitmap of the (small) output grid, a popcount prefix sum ranks them.
// Integer work only; bound by L2 atomics/hash-probe latency, tables stay in L2/MALL.
#include "common.h"

struct Shape3 { int z, y, x; };

__global__ __launch_bounds__(256) void k_index_build(const int32_t *coords, int n, const int32_t *n_dev, Shape3 s, uint64_t *keys,
                                                    int32_t *vals, uint32_t mask) {
  const int N = ls3d_count(n, n_dev);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int32_t *c = coords + 4 * (size_t)i;
    const int slot = ls3d_hash_claim(keys, mask, ls3d_key(c[0], c[1], c[2], c[3], s.z, s.y, s.x));
    vals[slot] = i;
  }
}

__global__ __launch_bounds__(256) void k_subm(const int32_t *coords, int n, const int32_t *n_dev, Shape3 s, Shape3 ks,
                                             const uint64_t *keys, const int32_t *vals, uint32_t mask, int32_t *nbr) {
  const int N = ls3d_count(n, n_dev);
  const int kvol = ks.z * ks.y * ks.x;
  const long long work = (long long)N * kvol;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t / kvol), k = (int)(t % kvol);
    const int kz = k / (ks.y * ks.x), ky = (k / ks.x) % ks.y, kx = k % ks.x;
    const int32_t *c = coords + 4 * (size_t)v;
    const int z = c[1] + kz - ks.z / 2, y = c[2] + ky - ks.y / 2, x = c[3] + kx - ks.x / 2;
    int r = -1;
    if (z >= 0 && z < s.z && y >= 0 && y < s.y && x >= 0 && x < s.x) {
      const int slot = ls3d_hash_find(keys, mask, ls3d_key(c[0], z, y, x, s.z, s.y, s.x));
      if (slot >= 0) r = vals[slot];
    }
    nbr[t] = r;
  }
}

struct ConvGeom {
  Shape3 in, out, ks, st, pd;
};

// output site fed by input (z,y,x) through kernel offset (kz,ky,kx):  o = (i + pad - k) / stride
__device__ __forceinline__ bool conv_out_site(const ConvGeom &g, int z, int y, int x, int kz, int ky, int kx, int &oz, int &oy, int &ox) {
  const int nz = z + g.pd.z - kz, ny = y + g.pd.y - ky, nx = x + g.pd.x - kx;
  if (nz < 0 || ny < 0 || nx < 0) return false;
  if (nz % g.st.z || ny % g.st.y || nx % g.st.x) return false;
  oz = nz / g.st.z; oy = ny / g.st.y; ox = nx / g.st.x;
  return oz < g.out.z && oy < g.out.y && ox < g.out.x;
}

// Marks the output sites in a BYTE map with plain stores (every hit writes the same 1: no atomics, no read-modify-write;
// up to 27 (input, offset) pairs hit one site and the atomicOr version spent 60-130 us per level on them) ...
__global__ __launch_bounds__(256) void k_conv_mark(const int32_t *coords, int n, const int32_t *n_dev, ConvGeom g, uint8_t *bytemap) {
  const int N = ls3d_count(n, n_dev);
  const int kvol = g.ks.z * g.ks.y * g.ks.x;
  const long long work = (long long)N * kvol;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / kvol), k = (int)(t % kvol);
    const int kz = k / (g.ks.y * g.ks.x), ky = (k / g.ks.x) % g.ks.y, kx = k % g.ks.x;
    const int32_t *c = coords + 4 * (size_t)i;
    int oz, oy, ox;
    if (conv_out_site(g, c[1], c[2], c[3], kz, ky, kx, oz, oy, ox)) bytemap[ls3d_key(c[0], oz, oy, ox, g.out.z, g.out.y, g.out.x)] = 1;
  }
}

// ... which this pass packs into the occupancy bitmap (bit = site's linear index) and its per-word population counts.
// bytemap is padded to a multiple of 32 bytes and zero-filled.
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t *bytemap, int nwords, uint32_t *bitmap, int32_t *cnt) {
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x) {
    const uint4 lo = *(const uint4 *)(bytemap + (size_t)w * 32), hi = *(const uint4 *)(bytemap + (size_t)w * 32 + 16);
    const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // bytes are 0 / 1: gather the low bit of each of the 4 bytes of a dword
      const uint32_t x = v[q];
      bits |= ((x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u)) << (4 * q);
    }
    bitmap[w] = bits;
    cnt[w] = __popc(bits);
  }
}

__global__ __launch_bounds__(256) void k_conv_emit(const uint32_t *bitmap, const int32_t *prefix, int nwords, Shape3 o, int out_cap,
                                                  const int32_t *total, int32_t *out_coords, int32_t *n_out, int32_t *overflow) {
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x) {
    uint32_t bits = bitmap[w];
    int rank = prefix[w];
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1;
      if (rank < out_cap) {
        uint64_t lin = ((uint64_t)w << 5) + (uint64_t)b;
        int32_t *c = out_coords + 4 * (size_t)rank;
        c[3] = (int)(lin % (uint64_t)o.x); lin /= (uint64_t)o.x;
        c[2] = (int)(lin % (uint64_t)o.y); lin /= (uint64_t)o.y;
        c[1] = (int)(lin % (uint64_t)o.z); lin /= (uint64_t)o.z;
        c[0] = (int)lin;
      }
      ++rank;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int t = *total;
    *n_out = t < out_cap ? t : out_cap;
    if (t > out_cap && overflow) *overflow = 1;
  }
}

__global__ __launch_bounds__(256) void k_conv_tables(const int32_t *coords, int n, const int32_t *n_dev, ConvGeom g, const uint32_t *bitmap,
                                                    const int32_t *prefix, int out_cap, int32_t *nbr_out, int32_t *nbr_inv) {
  const int N = ls3d_count(n, n_dev);
  const int kvol = g.ks.z * g.ks.y * g.ks.x;
  const long long work = (long long)N * kvol;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / kvol), k = (int)(t % kvol);
    const int kz = k / (g.ks.y * g.ks.x), ky = (k / g.ks.x) % g.ks.y, kx = k % g.ks.x;
    const int32_t *c = coords + 4 * (size_t)i;
    int oz, oy, ox, o = -1;
    if (conv_out_site(g, c[1], c[2], c[3], kz, ky, kx, oz, oy, ox)) {
      const uint64_t lin = ls3d_key(c[0], oz, oy, ox, g.out.z, g.out.y, g.out.x);
      const uint32_t word = bitmap[lin >> 5];
      o = prefix[lin >> 5] + __popc(word & ((1u << (lin & 31)) - 1u));
      if (o < out_cap) nbr_out[(size_t)o * kvol + k] = i;  // each (o,k) has exactly one source: no conflict
      else o = -1;
    }
    if (nbr_inv) nbr_inv[t] = o;
  }
}

// neighbour bitmask of every table row (bit k set <=> tbl[r][k] >= 0), kvol <= 32.  Rows sorted by this key give
// the gather-GEMM tiles whose rows share their empty kernel offsets.
__global__ __launch_bounds__(256) void k_tbl_mask(const int32_t *tbl, int n, const int32_t *n_dev, int kvol, int32_t *mask) {
  const int N = ls3d_count(n, n_dev);
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < N; r += gridDim.x * blockDim.x) {
    uint32_t m = 0;
    for (int k = 0; k < kvol; ++k) m |= (tbl[(size_t)r * kvol + k] >= 0 ? 1u : 0u) << k;
    mask[r] = (int32_t)m;
  }
}

// sort key of every table row for ONE batched sort of several tables: (segment << 27) | (mask ^ flip), kvol <= 27.
// flip = 0x7FFFFFF orders a segment's rows by DESCENDING mask (densest rows first).
__global__ __launch_bounds__(256) void k_tbl_sortkey(const int32_t *tbl, int n, const int32_t *n_dev, int kvol, uint32_t seg_bits, uint32_t flip,
                                                     int32_t *keys) {
  const int N = ls3d_count(n, n_dev);
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    uint32_t m = 0x7FFFFFFu ^ flip;  // rows beyond the device count: the largest key of the segment, they sort last
    if (r < N) {
      m = 0;
      for (int k = 0; k < kvol; ++k) m |= (tbl[(size_t)r * kvol + k] >= 0 ? 1u : 0u) << k;
    }
    keys[r] = (int32_t)(seg_bits | (m ^ flip));
  }
}

// Sort key of the rows of a strided convolution's TRANSPOSED table (what SparseInverseConv3d and the dgrad of SparseConv3d run on) without
// reading the table: input site c feeds output (c + pad - k) / stride through offset k iff that division is exact in every dimension (every
// such output site exists: the rulebook generated it), so a row's offset mask is a function of the residues (c + pad) mod stride - at most
// stride_z stride_y stride_x classes (8 for the UNet's stride-2 layers) instead of a 27-bit mask, i.e. ONE 8-bit radix pass instead of four.
// (Rows at the border of the output grid miss some offsets of their class; they are processed with it.)  lut[residue tuple]: position of the
// class among all classes by descending offset count, so ascending keys put the densest rows first, as the mask order does.
struct ParityGeom { int pad[3], stride[3]; unsigned char lut[64]; };
__global__ __launch_bounds__(256) void k_parity_sortkey(const int32_t *__restrict__ coords, int n, const int32_t *n_dev, ParityGeom g, uint32_t seg_bits,
                                                        uint32_t spare, int32_t *keys) {
  const int N = ls3d_count(n, n_dev);
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    uint32_t cls = spare;  // rows beyond the device count: the largest key of the segment
    if (r < N) {
      cls = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        int res = (coords[(size_t)r * 4 + 1 + d] + g.pad[d]) % g.stride[d];
        if (res < 0) res += g.stride[d];
        cls = cls * (uint32_t)g.stride[d] + (uint32_t)res;
      }
      cls = g.lut[cls];
    }
    keys[r] = (int32_t)(seg_bits | cls);
  }
}

// positions in the concatenation of several segments -> positions inside the own segment (int64 -> int32)
struct SegOffsets { int32_t off[17]; int32_t nseg; };
template <typename T>
__global__ __launch_bounds__(256) void k_segment_local(const T *perm, int n, SegOffsets so, int32_t *out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int s = 0;
    while (s + 1 < so.nseg && i >= so.off[s + 1]) ++s;
    out[i] = (int32_t)(perm[i] - (T)so.off[s]);
  }
}

__global__ __launch_bounds__(256) void k_fill_m1(int32_t *p, long long n) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) p[t] = -1;
}

static inline size_t rb_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int ls3d_index_build(const int32_t *coords, int n, const int32_t *n_dev, const int32_t shape[3], uint64_t *keys,
                                int32_t *vals, int cap, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!coords || !shape || !keys || !vals || n < 0 || cap < 2 || (cap & (cap - 1)) || (long long)cap < 2LL * n) return LS3D_ERR_ARG;
  hipMemsetAsync(keys, 0xFF, (size_t)cap * 8, stream);
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_index_build, ls3d_grid(n), dim3(256), 0, stream, coords, n, n_dev, Shape3{shape[0], shape[1], shape[2]}, keys, vals,
                     (uint32_t)(cap - 1));
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_rulebook_subm(const int32_t *coords, int n, const int32_t *n_dev, const int32_t shape[3], const int32_t ksize[3],
                                  const uint64_t *keys, const int32_t *vals, int cap, int32_t *nbr, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!coords || !shape || !ksize || !keys || !vals || !nbr || n < 0 || (cap & (cap - 1))) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  const int kvol = ksize[0] * ksize[1] * ksize[2];
  hipLaunchKernelGGL(k_subm, ls3d_grid((long long)n * kvol), dim3(256), 0, stream, coords, n, n_dev, Shape3{shape[0], shape[1], shape[2]},
                     Shape3{ksize[0], ksize[1], ksize[2]}, keys, vals, (uint32_t)(cap - 1), nbr);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_rulebook_masks(const int32_t *tbl, int n, const int32_t *n_dev, int kvol, int32_t *mask, ls3d_stream_t stream) {
  if (!tbl || !mask || n < 0 || kvol < 1 || kvol > 31) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_tbl_mask, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, tbl, n, n_dev, kvol, mask);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_rulebook_sort_keys(const int32_t *tbl, int n, const int32_t *n_dev, int kvol, int segment, int descending, int32_t *keys,
                                       ls3d_stream_t stream) {
  if (!tbl || !keys || n < 0 || kvol < 1 || kvol > 27 || segment < 0 || segment > 15) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_tbl_sortkey, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, tbl, n, n_dev, kvol, (uint32_t)segment << 27,
                     descending ? 0x7FFFFFFu : 0u, keys);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_rulebook_parity_keys(const int32_t *coords_in, int n, const int32_t *n_dev, const int32_t ksize_host[3], const int32_t stride_host[3],
                                         const int32_t pad_host[3], int segment, int class_bits, int32_t *keys, ls3d_stream_t stream) {
  if (!coords_in || !keys || !ksize_host || !stride_host || !pad_host || n < 0 || segment < 0 || class_bits < 1 || class_bits > 16) return LS3D_ERR_ARG;
  ParityGeom g;
  int cnt[3][8];
  long long classes = 1;
  for (int d = 0; d < 3; ++d) {
    const int s = stride_host[d], K = ksize_host[d];
    if (s < 1 || K < 1 || pad_host[d] < 0) return LS3D_ERR_ARG;
    if (s > 8) return LS3D_ERR_UNSUPPORTED;
    g.pad[d] = pad_host[d];
    g.stride[d] = s;
    for (int r = 0; r < s; ++r) {
      cnt[d][r] = 0;
      for (int k = 0; k < K; ++k) cnt[d][r] += ((r - k) % s == 0) ? 1 : 0;  // offsets k with (c + pad - k) divisible by the stride
    }
    classes *= s;
  }
  // the top class value is the spare key
  if (classes > 64 || classes >= (1ll << class_bits) || ((long long)segment << class_bits) >= (1ll << 31)) return LS3D_ERR_UNSUPPORTED;
  int total[64];
  for (int c = 0; c < (int)classes; ++c) {
    const int rx = c % g.stride[2], ry = (c / g.stride[2]) % g.stride[1], rz = c / (g.stride[2] * g.stride[1]);
    total[c] = cnt[0][rz] * cnt[1][ry] * cnt[2][rx];
  }
  for (int c = 0; c < 64; ++c) g.lut[c] = 0;
  for (int c = 0; c < (int)classes; ++c) {
    int rk = 0;
    for (int q = 0; q < (int)classes; ++q) rk += (total[q] > total[c] || (total[q] == total[c] && q < c)) ? 1 : 0;
    g.lut[c] = (unsigned char)rk;
  }
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_parity_sortkey, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, coords_in, n, n_dev, g, (uint32_t)segment << class_bits,
                     (1u << class_bits) - 1u, keys);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_segment_local_index(const int64_t *perm, int n, const int32_t *seg_offsets_host, int nseg, int32_t *out, ls3d_stream_t stream) {
  if (!perm || !out || !seg_offsets_host || n < 0 || nseg < 1 || nseg > 16) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  SegOffsets so;
  for (int s = 0; s <= nseg; ++s) so.off[s] = seg_offsets_host[s];
  so.nseg = nseg;
  hipLaunchKernelGGL(k_segment_local<int64_t>, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, perm, n, so, out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_segment_local_index32(const int32_t *perm, int n, const int32_t *seg_offsets_host, int nseg, int32_t *out, ls3d_stream_t stream) {
  if (!perm || !out || !seg_offsets_host || n < 0 || nseg < 1 || nseg > 16) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  SegOffsets so;
  for (int s = 0; s <= nseg; ++s) so.off[s] = seg_offsets_host[s];
  so.nseg = nseg;
  hipLaunchKernelGGL(k_segment_local<int32_t>, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, perm, n, so, out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

static inline long long rb_words(int batch, const int32_t oshape[3]) {
  return ((long long)batch * oshape[0] * oshape[1] * oshape[2] + 31) / 32;
}

extern "C" size_t ls3d_rulebook_conv_workspace_bytes(int batch, const int32_t oshape[3]) {
  const long long nw = rb_words(batch, oshape);
  return rb_align((size_t)nw * 4) * 3 + rb_align(ls3d_scan_tmp_ints(nw) * 4) + 256 + rb_align((size_t)nw * 32);
}

extern "C" int ls3d_rulebook_conv(const int32_t *coords_in, int n_in, const int32_t *n_in_dev, int batch, const int32_t in_shape[3],
                                  const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3], void *workspace,
                                  size_t workspace_bytes, int32_t *out_coords, int out_cap, int32_t *n_out_dev, int32_t *nbr_out,
                                  int32_t *nbr_inv, int32_t *overflow_dev, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!coords_in || !in_shape || !ksize || !stride || !pad || !workspace || !out_coords || !n_out_dev || !nbr_out) return LS3D_ERR_ARG;
  if (n_in < 0 || batch < 1 || out_cap < 1) return LS3D_ERR_ARG;
  ConvGeom g;
  g.in = Shape3{in_shape[0], in_shape[1], in_shape[2]};
  g.ks = Shape3{ksize[0], ksize[1], ksize[2]};
  g.st = Shape3{stride[0], stride[1], stride[2]};
  g.pd = Shape3{pad[0], pad[1], pad[2]};
  if (g.st.z < 1 || g.st.y < 1 || g.st.x < 1) return LS3D_ERR_ARG;
  int32_t os[3];
  for (int a = 0; a < 3; ++a) os[a] = (in_shape[a] + 2 * pad[a] - ksize[a]) / stride[a] + 1;
  if (os[0] < 1 || os[1] < 1 || os[2] < 1) return LS3D_ERR_ARG;
  g.out = Shape3{os[0], os[1], os[2]};
  const long long nw = rb_words(batch, os);
  if (nw > 0x7FFFFFFFLL) return LS3D_ERR_UNSUPPORTED;
  if (workspace_bytes < ls3d_rulebook_conv_workspace_bytes(batch, os)) return LS3D_ERR_WORKSPACE;
  char *base = (char *)workspace;
  uint32_t *bitmap = (uint32_t *)base; base += rb_align((size_t)nw * 4);
  int32_t *cnt = (int32_t *)base; base += rb_align((size_t)nw * 4);
  int32_t *prefix = (int32_t *)base; base += rb_align((size_t)nw * 4);
  int32_t *scan_tmp = (int32_t *)base; base += rb_align(ls3d_scan_tmp_ints(nw) * 4);
  int32_t *total = (int32_t *)base; base += 256;
  uint8_t *bytemap = (uint8_t *)base;
  const int kvol = ksize[0] * ksize[1] * ksize[2];
  hipMemsetAsync(bytemap, 0, (size_t)nw * 32, stream);
  hipLaunchKernelGGL(k_fill_m1, ls3d_grid((long long)out_cap * kvol), dim3(256), 0, stream, nbr_out, (long long)out_cap * kvol);
  if (n_in > 0)
    hipLaunchKernelGGL(k_conv_mark, ls3d_grid((long long)n_in * kvol), dim3(256), 0, stream, coords_in, n_in, n_in_dev, g, bytemap);
  hipLaunchKernelGGL(k_pack_bits, ls3d_grid(nw), dim3(256), 0, stream, (const uint8_t *)bytemap, (int)nw, bitmap, cnt);
  int rc = ls3d_exclusive_scan_i32(cnt, prefix, (int)nw, scan_tmp, total, stream);
  if (rc != LS3D_OK) return rc;
  hipLaunchKernelGGL(k_conv_emit, ls3d_grid(nw), dim3(256), 0, stream, (const uint32_t *)bitmap, (const int32_t *)prefix, (int)nw, g.out,
                     out_cap, (const int32_t *)total, out_coords, n_out_dev, overflow_dev);
  if (n_in > 0)
    hipLaunchKernelGGL(k_conv_tables, ls3d_grid((long long)n_in * kvol), dim3(256), 0, stream, coords_in, n_in, n_in_dev, g,
                       (const uint32_t *)bitmap, (const int32_t *)prefix, out_cap, nbr_out, nbr_inv);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
