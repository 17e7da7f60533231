/*
 * pcc_hip.h -- C ABI of libpcc_hip.so, the MI355X (gfx950) sparse-voxel codec hot path.
 *
 * This is the drop-in boundary.  The reference
 * (ikt-luh/Unified-Point-Cloud-Compression) reaches this path through two Python import
 * surfaces, `import MinkowskiEngine as ME` and `compressai.*`; the Python shims in
 * unified_point_cloud_compression_amd/{MinkowskiEngine,compressai}/ bind these entry points
 * with ctypes.  Every entry point names the reference interface it replaces (file:line in
 * /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with `h_`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - functions return 0 on success, a negative PCC_E* code otherwise; pcc_last_error()
 *     gives the message of the calling thread's last failure; no exceptions, no ownership
 *     transfer: outputs and workspaces are caller-allocated (sizes from the *_ws_bytes /
 *     *_elems queries).  Two exceptions, owned by the library per process: a grow-only device
 *     scratch for the 16-bit operand planes / split-K partial tiles of the convolution entry
 *     points (pcc_conv_fwd, pcc_conv_fwd_pairs, pcc_convt_fwd*, pcc_gdn_fwd) and a small table
 *     buffer of pcc_convt_fwd_csr*.  They are allocated with hipMalloc outside the caller's
 *     allocator, and GROWING one synchronises the device once (hipDeviceSynchronize + hipFree);
 *     otherwise nothing here synchronises the stream;
 *   - threading: ONE stream per device at a time for the convolution entry points (they share
 *     the scratch above in stream order); calls from two host threads or on two streams of the
 *     same device must be serialised by the caller.  The pure coordinate / entropy entry points
 *     keep no state;
 *   - coordinates are packed int64 keys  b<<48 | (x+2^15)<<32 | (y+2^15)<<16 | (z+2^15);
 *     a coordinate set is a strictly ascending key array ("canonical order" = the order
 *     `utils.sort_tensor` produces, utils.py:142-165);
 *   - features are row-major float32 [N, C].
 */
#ifndef PCC_HIP_H
#define PCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCC_OK 0
#define PCC_EINVAL (-1)   /* bad argument / unsupported shape */
#define PCC_EHIP (-2)     /* HIP runtime error              */
#define PCC_EWS (-3)      /* workspace too small            */

#define PCC_ACT_NONE 0
#define PCC_ACT_RELU 1    /* ME.MinkowskiReLU      (model/transforms.py:148,153,158)          */
#define PCC_ACT_LEAKY 2   /* ME.MinkowskiLeakyReLU (model/entropy_models.py:179,181,187,189)  */

#define PCC_MAP_HDR_INTS 512  /* int32 words of a kernel-map header (device resident)        */
#define PCC_MAP_MAX_SEG 8

int pcc_version(void);
const char* pcc_last_error(void);
/* number of compute units / name of the current device (sanity: must be gfx950). */
int pcc_device_info(int* h_cu_count, char* h_arch, int h_arch_len);

/* ------------------------------------------------------------------------------------------
 * a1 / a11  coordinate keys            (ME.SparseTensor ctor: model/model.py:66-70,147-161,227;
 *                                       ME.utils.sparse_quantize: model/model.py:152-156)
 * ---------------------------------------------------------------------------------------- */
/* {min b,x,y,z, max b,x,y,z} of [n,4] coordinates (int32, or float: floored) -- sizes the key range / grid lattices */
int pcc_coords_bounds(const void* coords, int32_t is_float, int64_t n, int32_t* out8 /*device*/, void* stream);
/* dst[i][:] = src[idx[i]][:]: features of user-ordered rows in canonical order (ME.SparseTensor) */
int pcc_rows_gather(const float* src, const int64_t* idx, int64_t m, int32_t c, float* dst, void* stream);
/* int32 [n,4] (b,x,y,z) -> keys */
int pcc_keys_pack_i32(const int32_t* coords, int64_t n, int64_t* keys, void* stream);
/* float [n,4] -> floor -> keys (the reference passes float coordinates, model/model.py:142-149) */
int pcc_keys_pack_f32(const float* coords, int64_t n, int64_t* keys, void* stream);
int pcc_keys_unpack(const int64_t* keys, int64_t n, int32_t* coords, void* stream);
/* Row ranges per batch index of a canonical key array whose row count *d_n may still be on the device: out[e] = first row with batch
 * index >= e for e in [0, entries), entries <= 12 (out_a: entries 0-3, out_b: 4-7, out_c: 8-11; 4 int64 each).  Lets a caller read a derived
 * set's size and its per-batch ranges (reference model/transforms.py:228-254: top-k per batch) in one host read. */
int pcc_batch_bounds(const int64_t* keys, const int64_t* d_n /*nullable: then n_host rows*/, int64_t n_host, int32_t entries,
                     int64_t* out_a, int64_t* out_b, int64_t* out_c, void* stream);

/* LSD radix sort of keys (stable), optional payload perm_out[i] = input index of output i.
 * Only 8-bit digits intersecting `bit_mask` (bits that may differ between keys) are sorted. */
size_t pcc_sort_ws_bytes(int64_t n);
int pcc_sort_keys(const int64_t* keys_in, int64_t n, uint64_t bit_mask, int64_t* keys_out,
                  int32_t* perm_out /*nullable*/, void* ws, size_t ws_bytes, void* stream);

/* adjacent-unique of a sorted key array.  uniq[] (capacity n), first[] (nullable; index in the
 * sorted array of the first occurrence), *d_count = number of unique keys (device int64). */
size_t pcc_unique_ws_bytes(int64_t n);
int pcc_unique_sorted(const int64_t* sorted_keys, int64_t n, int64_t* uniq, int32_t* first,
                      int64_t* d_count, void* ws, size_t ws_bytes, void* stream);

/* *d_flag != 0 when keys[] is strictly ascending, else 0 (device int32). */
int pcc_keys_is_canonical(const int64_t* keys, int64_t n, int32_t* d_flag, void* stream);

/* a2(i)  strided output set, ME stride map: out = unique(floor(c/m)*m), m = new tensor stride
 * (power of two).  (ME.MinkowskiConvolution stride=2: model/transforms.py:33,37,41;
 * model/entropy_models.py:180,182; coordinate-only use model/model.py:227-229 = row a12.) */
size_t pcc_stride_ws_bytes(int64_t n);
int pcc_coords_stride(const int64_t* keys, int64_t n, int32_t new_stride, uint64_t bit_mask,
                      int64_t* out_keys /*cap n*/, int64_t* d_count, void* ws, size_t ws_bytes,
                      void* stream);

/* a3(i)  generative output set: unique{ c + off_k * ts_out }, K = kernel_size^3 offsets
 * (ME.MinkowskiGenerativeConvolutionTranspose: model/transforms.py:129,133,137;
 * model/entropy_models.py:186,188). out_keys capacity n*K. */
size_t pcc_expand_ws_bytes(int64_t n, int32_t kernel_size);
int pcc_coords_expand(const int64_t* keys, int64_t n, int32_t kernel_size, int32_t ts_out,
                      uint64_t bit_mask, int64_t* out_keys /*cap n*K*/, int64_t* d_count, void* ws,
                      size_t ws_bytes, void* stream);

/* a3(i)+(ii) fused: generative output set AND its transposed kernel map in CSR form from ONE sort.
 * Candidates are sorted as 32-bit cell indices of the output lattice h_lattice = {lo_x,lo_y,lo_z, cells_x,cells_y,
 * cells_z, pitch (= ts_out), batches} (needs <= 2^32 cells) with the pair id i*K+k as payload:
 *   out_keys[o]                       the output coordinate set (capacity n*K), *d_count rows
 *   pair_ids[first[o] .. first[o+1])  the (input row, offset) pairs landing on output row o, ascending
 * pair_ids: n*K ints, first: n*K+1 ints. */
size_t pcc_expand_csr_ws_bytes(int64_t n, int32_t kernel_size);
int pcc_coords_expand_csr(const int64_t* keys, int64_t n, int32_t kernel_size, int32_t ts_out,
                          const int32_t* h_lattice, int64_t* out_keys, int64_t* d_count, int32_t* pair_ids,
                          int32_t* first, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * a2(ii) / a3  kernel map              (ME kernel map behind every conv forward)
 *
 * A map is three device arrays:
 *   hdr  int32[PCC_MAP_HDR_INTS]  segment table (written by the build kernels)
 *   nbr  int32[...]               per segment a [k_count][pos_count] table of input rows (-1 none)
 *   rows int32[n_out] or NULL     output row of each position (NULL: position == output row)
 * conv:       one segment, all K offsets, positions = output rows; when `rows` is given (non-transposed build) the
 *             positions are the output rows in Morton (Z-curve) order of their coordinates and rows[] receives that
 *             permutation: tiles of consecutive positions are compact 3-D blobs (L1/L2 locality of the gathers).
 * transposed: outputs are grouped by their residue class modulo the up-sampling stride
 *             (stride^3 segments); a class only lists the offsets that can reach it.
 * ---------------------------------------------------------------------------------------- */
/* number of int32 the nbr array needs (upper bound for transposed maps) */
int64_t pcc_map_nbr_elems(int64_t n_out, int32_t kernel_size, int32_t stride, int32_t transposed);
size_t pcc_map_ws_bytes(int64_t n_out);
/* in = out + off_k*step (conv, step = input tensor stride)
 * out = in + off_k*step (transposed, step = output tensor stride, stride = up-sampling factor)
 * *d_pairs (device int64, nullable) receives the number of valid (in,out) pairs. */
int pcc_kernel_map_build(const int64_t* in_keys, int64_t n_in, const int64_t* out_keys, int64_t n_out,
                         int32_t kernel_size, int32_t step, int32_t stride, int32_t transposed,
                         int32_t* hdr, int32_t* nbr, int32_t* rows /*transposed: required; conv: optional (Morton)*/,
                         int64_t* d_pairs, const uint64_t* grid_bits /*nullable*/, const int32_t* grid_rank,
                         const int32_t* h_grid, void* ws, size_t ws_bytes, void* stream);

/* Grid index of the INPUT set of a map (replaces MinkowskiEngine's coordinate hash map for lookups): an occupancy
 * bitmap over the set's bounding lattice plus an exclusive popcount prefix per 64-bit word.  Canonical order equals
 * ascending cell order, so row(cell) = rank[word] + popcount(bits below) -- two coalesced reads per neighbour query.
 * h_grid (host int32[8]) = {lo_x, lo_y, lo_z, cells_x, cells_y, cells_z, pitch (tensor stride), batches}.
 * Without a grid the map build falls back to binary search over the sorted keys. */
int64_t pcc_grid_words(const int32_t* h_grid);
size_t pcc_grid_ws_bytes(int64_t words);
int pcc_grid_build(const int64_t* keys, int64_t n, const int32_t* h_grid, uint64_t* bits /*[words]*/,
                   int32_t* rank /*[words]*/, void* ws, size_t ws_bytes, void* stream);
/* a2-i through the bitmap: strided set = occupancy of the coarse cells.  keys = the FINE canonical set; h_grid = the
 * COARSE lattice (pitch h_grid[6] = new tensor stride, origin a multiple of it).  Marks, ranks and reads the set back
 * out in bitmap (= canonical) order: out_keys (capacity n), *d_count, plus the coarse set's grid index in bits / rank.
 * Same result as pcc_coords_stride without the sort; ws of pcc_grid_ws_bytes(words). */
/* keys may be in ANY order and hold duplicates (marking a cell is idempotent).  d_n (device int64, nullable): the number
 * of valid rows when it is still on the device -- a set derived a moment ago whose size the host has not read; n is then
 * the capacity of keys[].  A chain of strided sets is queued this way without a host read between its links. */
int pcc_coords_stride_grid(const int64_t* keys, int64_t n, const int64_t* d_n, const int32_t* h_grid, uint64_t* bits,
                           int32_t* rank, int64_t* out_keys, int64_t* d_count, void* ws, size_t ws_bytes, void* stream);
/* a1 through the bitmap: canonical order and first-wins de-duplication of USER-ordered keys without a sort.
 * bits / rank = the set's grid index, out_keys (capacity n) = canonical keys, first_user[canonical position] = smallest
 * user row with that coordinate, d_count[0] = number of distinct coordinates, d_count[1] != 0 when some key is off the
 * lattice (results invalid: use pcc_sort_keys + pcc_unique_sorted).  h_grid must cover all keys. */
int pcc_keys_canonicalize_grid(const int64_t* keys, int64_t n, const int32_t* h_grid, uint64_t* bits, int32_t* rank,
                               int64_t* out_keys, int32_t* first_user, int64_t* d_count, void* ws, size_t ws_bytes,
                               void* stream);
/* a3-i through the bitmaps: generative expansion without sorting the n*K candidates.
 *   pcc_coords_expand_grid     : marks the output lattice h_out (pitch ts_out), ranks it and reads the canonical output
 *                                keys back out; bits / rank = grid index of the output set; *d_count = n_out.
 *   pcc_coords_expand_grid_csr : (n_out known) CSR pair lists of the transposed map by probing the INPUT set's grid at
 *                                c - off_k in ascending input row: first[n_out+1], pair_ids[n_in*K] (pair = i*K + k),
 *                                identical to pcc_coords_expand_csr's. */
int pcc_coords_expand_grid(const int64_t* keys, int64_t n, int32_t kernel_size, const int32_t* h_out, uint64_t* bits,
                           int32_t* rank, int64_t* out_keys, int64_t* d_count, void* ws, size_t ws_bytes, void* stream);
/* Transposed convolution on a SUBSET of its output rows (the rows kept by the top-k pruning), straight from their CSR
 * pair lists (pcc_coords_expand_grid_csr on those rows): pairs bucketed by offset, gathered pair GEMM, sum in CSR order.
 * packed_w: pcc_conv_pack_weights layout.  pairs = host value of first[n_out].  T: pcc_convt_rows_t_elems floats. */
size_t pcc_convt_rows_int_ws_bytes(int64_t pairs, int32_t K);
int64_t pcc_convt_rows_t_elems(int64_t pairs, int32_t K, int32_t cout);
int pcc_convt_fwd_rows(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w, const float* bias,
                       int32_t K, int32_t cout, const int32_t* first, const int32_t* pair_ids, int64_t n_out,
                       int64_t pairs, float* T, float* out, int32_t act, float slope, void* int_ws, size_t int_ws_bytes,
                       int32_t arith /*PCC_ARITH_*, see the convolution section*/, int32_t* d_guard /*nullable*/, void* stream);
/* conv-form map (one segment, nbr[k*n_out + o] = input row or -1) from CSR pair lists: evaluates a transposed conv on a
 * subset of its output rows with pcc_conv_fwd / pcc_conv_fwd_pairs.  hdr: PCC_MAP_HDR_INTS ints, nbr: K*n_out ints. */
int pcc_map_from_csr(const int32_t* first, const int32_t* pair_ids, int64_t n_out, int32_t kernel_size, int32_t* hdr,
                     int32_t* nbr, void* stream);
size_t pcc_expand_grid_csr_ws_bytes(int64_t n_out);
/* d_total (device int64, nullable) receives first[n_out], the number of pairs: lets the caller read that size together
 * with others it is waiting for instead of indexing first[] from the host. */
int pcc_coords_expand_grid_csr(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                               const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in, int64_t n_in,
                               int32_t* first, int32_t* pair_ids, int64_t* d_total, void* ws, size_t ws_bytes, void* stream);
/* the same lists, kernel offsets of the pair ids numbered z fastest (iz + KS*iy + KS*KS*ix) -- for product buffers laid out
 * [input row][kx][ky][kz][c]: the composite levels gather z-runs of output rows from adjacent memory */
/* The same pair lists in ONE pass (round 4): the 256 rows of a workgroup write into a slot of their own, pair_ids[w * SLOT ...) with
 * SLOT = 256 x the most pairs a row can have ((k+1)/2)^3, so no prefix sum over all rows (and no second probing pass) is needed.
 * first[o] (n_out entries) = absolute start of row o's list; the list ends at first[o + 1], except for the last row of a workgroup
 * (o % 256 == 255, or the last row): wg_end[o / 256].  Same pairs, same order inside a row as pcc_coords_expand_grid_csr[_zk].
 * kernel_size 5 or 7, input pitch >= 2 x output pitch.  pair_ids: pcc_expand_grid_csr_slot_elems(n_out, kernel_size) ints (sized for
 * the worst case; only the written part is touched); wg_end: ceil(n_out / 256) ints; d_total (nullable): receives the pair total. */
int64_t pcc_expand_grid_csr_slot_elems(int64_t n_out, int32_t kernel_size);
int pcc_coords_expand_grid_csr_slots(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                                     const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in, int64_t n_in,
                                     int32_t* first, int32_t* pair_ids, int32_t* wg_end, int64_t* d_total /*nullable*/, int32_t zk,
                                     void* stream);
int pcc_coords_expand_grid_csr_zk(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                               const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in, int64_t n_in,
                               int32_t* first, int32_t* pair_ids, int64_t* d_total, void* ws, size_t ws_bytes, void* stream);
/* dense [K][n_out] view of any map (testing / inspection): -1 where no pair */
int pcc_map_to_dense(const int32_t* hdr, const int32_t* nbr, const int32_t* rows, int64_t n_out,
                     int32_t K, int32_t* dense, void* stream);

/* ------------------------------------------------------------------------------------------
 * a2(iii) / a3  sparse convolution forward, output stationary, fp32 MFMA
 *   out[o] = act( bias + sum_k feat_in[nbr_k(o)] @ W[k] )
 * (ME.MinkowskiConvolution / MinkowskiGenerativeConvolutionTranspose forward, 17+5 sites:
 *  model/transforms.py:33-43,127-166; model/entropy_models.py:178-190.)
 * Weights are consumed in a packed layout produced once per parameter update.
 * ---------------------------------------------------------------------------------------- */
/* MFMA-path arithmetic is an ARGUMENT of every convolution entry point (`arith`), never process state: two threads, or the
 * encoder and the decoder of one process, may use different forms at the same time, and a caller that must reproduce a result
 * bit for bit elsewhere (the hyper-synthesis h_s, whose scales / means select the rANS table rows on both sides:
 * model/entropy_models.py:371-400,438-484) names the form it was produced in.
 *   PCC_ARITH_F32  fp32-input MFMA instructions only (v_mfma_f32_32x32x2_f32);
 *   PCC_ARITH_BF6  every fp32 product on the bf16 matrix pipe from an exact three-way bf16 split of both operands (six cross
 *                  terms, fp32 accumulation: 24 bits per element, no range condition, 2.67x the fp32-MFMA rate);
 *   PCC_ARITH_H3   as BF6 for the gathered 3x3x3 convolutions and GDN; the products whose output row depends on ONE input row
 *                  (dense products of the generative transposed convolutions, pair-list GEMMs) as row / column-scaled fp16
 *                  pairs, three MFMA terms, under the range guard below.
 * Range guard of the three-term form.  The form carries every element within 2^-18 of its row / column
 * maximum to >= 22 bits and smaller ones with an absolute error of 2^-28 of that maximum, so a product of depth cin is off by
 * at most cin * 2^-27 * max|row| * max|column| beyond fp32 behaviour.  With `d_guard` non-NULL every fp16-pair launch of the
 * call ORs 1 into *d_guard (device int32, zeroed by the caller) when the scales of one of its tiles admit more than
 * PCC_H_GUARD_BUDGET (absolute, in output units); the caller reads the word with a size it reads anyway and repeats THAT call
 * with PCC_ARITH_BF6.  d_guard is ignored by the other two forms. */
#define PCC_ARITH_F32 0
#define PCC_ARITH_BF6 1
#define PCC_ARITH_H3 2
#define PCC_H_GUARD_BUDGET 2.5e-5f   /* a quarter of the 1e-4 parity bar: max|row| * max|column| > 26 at cin = 128 */
/* 4-channel inputs (the codec's first layer, 4 -> 128, 5x5x5, stride 2): from `rows` output rows on, the (offset, channel)
 * pairs are flattened into one reduction axis and the convolution runs in 32-wide chunks of 8 offsets on the six-term bf16
 * form (default 65536; env PCC_IN4_MIN_ROWS; negative: never).  Tests lower it to reach the path on small inputs. */
int pcc_set_in4_min_rows(int64_t rows);
/* pcc_conv_thin_grid_fwd with one output channel over 16 input channels (the last level's occupancy head): from `rows` rows on
 * the projection pass pre-adds a column's three z terms for the middle row and the gather reads one value per (dx, dy) column
 * (default 2^20; env PCC_THIN_Z_MIN_ROWS; negative: never).  Tests lower it to reach the path on small inputs. */
int pcc_set_thin_z_min_rows(int64_t rows);
int64_t pcc_conv_packed_elems(int32_t K, int32_t cin, int32_t cout);
/* W: ME layout [K, cin, cout] row-major (state_dict `kernel`).  packed_cap: floats available at
 * `packed`; a buffer smaller than pcc_conv_packed_elems(K, cin, cout) is refused (PCC_EWS), never written past. */
int pcc_conv_pack_weights(const float* W, int32_t K, int32_t cin, int32_t cout, float* packed,
                          int64_t packed_cap, void* stream);
/* The same pack read through a transposed / offset-reversed view of a stored kernel: W'[k][ci][co] = W[flip ? K-1-k : k][co][ci]
 * when `transpose` (W stored [K][cout][cin]), else W[flip ? K-1-k : k][ci][co] -- the kernels of the data gradient
 * (reference train.py:221-227 back-propagates through every ME convolution) without a copy in front.  MFMA layout only
 * (cin a multiple of 32, cout > 16); other shapes are refused. */
int pcc_conv_pack_weights_ex(const float* W, int32_t K, int32_t cin, int32_t cout, int32_t transpose, int32_t flip, float* packed,
                             int64_t packed_cap, void* stream);
size_t pcc_conv_ws_bytes(int64_t n_in, int32_t K, int32_t cin, int32_t cout);
/* Occupancy head `predict_i` (model/transforms.py:141-160) in one pass over the features:
 *   logits = conv_k3(relu(conv_k3(x; W0, b0)); W2, b2),  W0: cin -> cmid (4 < cmid <= 16), W2: cmid -> 1.
 * packed_w0: pcc_conv_pack_weights(27, cin, cmid) layout; w2: the second kernel as [27][cmid] (its
 * pcc_conv_pack_weights(27, cmid, 1) layout); hdr/nbr: the canonical 3x3x3 map of the set onto itself;
 * tiles/n_tiles: optional band-ordered tile table (pcc_band_tiles_build) or both NULL; ws: pcc_conv_head_ws_bytes(n). */
int pcc_conv_head_supported(int32_t cin, int32_t cmid);
size_t pcc_conv_head_ws_bytes(int64_t n);
int pcc_conv_head_fwd(const float* feat, int64_t n, int32_t cin, const float* packed_w0, const float* bias0 /*nullable*/,
                      int32_t cmid, const float* w2, const float* bias2 /*nullable [1]*/, const int32_t* hdr,
                      const int32_t* nbr, const int32_t* tiles, const int32_t* n_tiles, float* logits, void* ws,
                      size_t ws_bytes, void* stream);
/* Band-ordered tiles (<= 16 consecutive canonical rows each, word = row0 | (rows-1) << 27) of a single-batch canonical
 * set for the stencil kernels: y cut into `nbands` bands, tiles numbered band-major then x ascending, so that a
 * contiguous tile range sweeps x inside one band and the dx = +-1 neighbour slabs stay in an XCD's L2.
 * lo_x/lo_y: smallest coordinate per axis, nx/ny: lattice cells per axis at pitch ts.  n < 2^27. */
int64_t pcc_band_tiles_cap(int64_t n, int32_t nx, int32_t nbands);
size_t pcc_band_tiles_ws_bytes(int32_t nx, int32_t nbands);
int pcc_band_tiles_build(const int64_t* keys, int64_t n, int32_t lo_x, int32_t nx, int32_t lo_y, int32_t ny, int32_t ts,
                         int32_t nbands, int32_t* tiles, int64_t tiles_cap, int32_t* n_tiles /*device*/, void* ws,
                         size_t ws_bytes, void* stream);
int pcc_conv_fwd(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                 const float* bias /*nullable [cout]*/, int32_t K, int32_t cout, const int32_t* hdr,
                 const int32_t* nbr, const int32_t* rows, int64_t n_out, float* out, int32_t act,
                 float slope, void* ws, size_t ws_bytes, int32_t arith, int32_t* d_guard /*nullable*/, void* stream);
/* Pair-list form of the same convolution for maps with mostly empty (offset, row) slots (5x5x5 kernels on surfaces):
 * the pairs of each offset are compacted and padded to 128-pair tiles, T[p] = feat[in(p)] @ W[k(p)] runs as a gathered
 * GEMM in which every MFMA row is a real pair, and out[o] = act(bias + sum_k T[pos(k,o)]) is summed in ascending k.
 * Conv maps only (one segment, nbr[k*n_out + o], no row list).  Same results as pcc_conv_fwd up to fp32 summation order.
 *   pcc_pair_plan_rank : pos[K*n_out], pstart[K+1], info[3] = {padded pairs, tiles, pairs}      (device arrays)
 *   pcc_pair_plan_fill : pair_in[padded pairs] (-1 = padding), tile_k[tiles]   -- after the host has read info
 *   pcc_conv_fwd_pairs : T is scratch of padded_pairs*cout floats */
int pcc_conv_pairs_supported(int32_t K, int32_t cin, int32_t cout);
size_t pcc_pair_plan_ws_bytes(int64_t n_out, int32_t K);
int pcc_pair_plan_rank(const int32_t* nbr, int64_t n_out, int32_t K, int32_t* pos, int32_t* pstart, int64_t* info,
                       void* ws, size_t ws_bytes, void* stream);
int pcc_pair_plan_fill(const int32_t* nbr, const int32_t* pos, const int32_t* pstart, int64_t n_out, int32_t K,
                       int64_t padded_pairs, int32_t* pair_in, int32_t* tile_k, void* stream);
int pcc_conv_fwd_pairs(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w, const float* bias,
                       int32_t K, int32_t cout, const int32_t* pair_in, const int32_t* tile_k, const int64_t* d_info,
                       int64_t padded_pairs, const int32_t* pos, int64_t n_out, float* T, float* out, int32_t act,
                       float slope, int32_t arith, int32_t* d_guard /*nullable*/, void* stream);


/* a3  generative transposed convolution, input stationary (ME.MinkowskiGenerativeConvolutionTranspose forward:
 * model/transforms.py:129,133,137; model/entropy_models.py:186,188).  Every (input row, offset) is one pair, so
 *   T[i][k][:] = feat_in[i] @ W[k]       one dense [n_in,cin] x [cin,K*cout] GEMM on the fp32 MFMA
 *   out[o]     = act(bias + sum_k T[nbr_k(o)][k][:])   ordered gather-sum through the transposed map
 * T: caller scratch of n_in*K*cout floats.  hdr/nbr/rows: a transposed map from pcc_kernel_map_build. */
int64_t pcc_convt_packed_elems(int32_t K, int32_t cin, int32_t cout);
int pcc_convt_pack_weights(const float* W, int32_t K, int32_t cin, int32_t cout, float* packed,
                           int64_t packed_cap, void* stream);
int pcc_convt_fwd(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                  const float* bias /*nullable [cout]*/, int32_t K, int32_t cout, const int32_t* hdr,
                  const int32_t* nbr, const int32_t* rows, int64_t n_out, float* T, float* out, int32_t act,
                  float slope, int32_t arith, int32_t* d_guard /*nullable*/, void* stream);

/* same with the CSR pair lists of pcc_coords_expand_csr (row pair_id of T); outputs in canonical row order */
int pcc_convt_fwd_csr(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                      const float* bias /*nullable [cout]*/, int32_t K, int32_t cout, const int32_t* first,
                      const int32_t* pair_ids, int64_t n_out, float* T, float* out, int32_t act, float slope,
                      const int32_t* ex_nbr /*nullable [ex_K][n_out]*/, int32_t ex_K,
                      const float* ex_bias /*[ex_K][cout]*/, int32_t arith, int32_t* d_guard /*nullable*/, void* stream);

/* pcc_convt_fwd_csr with the constant-per-existing-neighbour term (ex_bias [27][cout]) keyed on the OUTPUT set's own grid
 * index (out_keys + pcc_grid_build arrays) instead of a [27][n_out] neighbour table; and a 3x3x3 convolution to <= 4 channels
 * whose neighbours also come straight from the grid index (packed_w: pcc_conv_pack_weights(27, cin, cout) thin layout).
 * Together they spare the composite up+head convolutions the 3x3x3 kernel map of their candidate set. */
int pcc_convt_fwd_csr_grid(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w, const float* bias /*nullable*/,
                           int32_t K, int32_t cout, const int32_t* first, const int32_t* pair_ids, int64_t n_out, float* T,
                           float* out, int32_t act, float slope, const int64_t* out_keys, const uint64_t* out_bits,
                           const int32_t* out_rank, const int32_t* h_out, const float* ex_bias,
                           const int32_t* wg_end /*nullable: first / pair_ids are the slotted lists of pcc_coords_expand_grid_csr_slots*/,
                           int32_t arith, int32_t* d_guard /*nullable*/, void* stream);
/* Chunked form of pcc_convt_fwd_csr for 7x7x7 composite levels: the per-pair products never exist as a whole.  Parent rows
 * are processed in chunks whose products fit the Infinity Cache (pcc_set_t_chunk_bytes, default 96 MiB; env PCC_T_CHUNK_MIB):
 * GEMM chunk -> staging buffer T (pcc_convt_chunk_t_bytes) -> ordered gather-sum of the children that chunk reaches, partial
 * sums carried in `out`.  Same summation order per output row as the one-pass form (bit-identical result).  in_keys/out_keys:
 * canonical keys of the input / output rows; ts_out: output pitch; ws: pcc_convt_chunk_ws_bytes.  ex_bias (nullable) as in
 * pcc_convt_fwd_csr_grid (then out_bits/out_rank/h_out are the output set's pcc_grid_build arrays). */
int pcc_set_t_chunk_bytes(int64_t bytes);
size_t pcc_convt_chunk_t_bytes(int64_t n_in, int32_t K, int32_t cout);
size_t pcc_convt_chunk_ws_bytes(int64_t n_in, int32_t K, int32_t cout);
int pcc_convt_fwd_csr_chunked(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w, const float* bias /*nullable*/,
                              int32_t K, int32_t cout, const int32_t* first, const int32_t* pair_ids, int64_t n_out,
                              const int64_t* in_keys, const int64_t* out_keys, int32_t ts_out, float* T, size_t t_bytes,
                              float* out, int32_t act, float slope, const uint64_t* out_bits /*nullable*/,
                              const int32_t* out_rank /*nullable*/, const int32_t* h_out /*nullable*/,
                              const float* ex_bias /*nullable*/, void* ws, size_t ws_bytes, int32_t arith,
                              int32_t* d_guard /*nullable*/, void* stream);
size_t pcc_thin_grid_ws_bytes(int64_t n, int32_t cout);
int pcc_conv_thin_grid_fwd(const float* feat, int64_t n, int32_t cin, const float* packed_w, const float* bias /*nullable*/,
                           int32_t cout, const int64_t* keys, const uint64_t* bits, const int32_t* rank, const int32_t* h_grid,
                           float* out, void* ws, size_t ws_bytes, void* stream);

/* a5  fused GDN / IGDN (GDN1 form), MinkowskiGDN.forward model/blocks.py:26-57:
 *   norm = beta + |x| @ gamma^T ; out = x / norm (inverse=0) or x * norm (inverse=1)
 * beta_raw/gamma_raw are the raw (un-reparametrised) CompressAI parameters; the
 * NonNegativeParametrizer is applied by pcc_gdn_pack. packed: pcc_gdn_packed_elems(c) floats
 * (0 = unsupported channel count), beta_eff: c floats. */
int64_t pcc_gdn_packed_elems(int32_t c);
int pcc_gdn_pack(const float* beta_raw, const float* gamma_raw, int32_t c, float beta_min,
                 float* packed, int64_t packed_cap, float* beta_eff, void* stream);
int pcc_gdn_fwd(const float* x, int64_t n, int32_t c, const float* packed, const float* beta_eff,
                int32_t inverse, float* out, int32_t arith, void* stream);

/* ------------------------------------------------------------------------------------------
 * Frame intake and hand-over of UnifiedModel.compress / decompress (model/model.py:141-161, 240-250), one launch each.
 * pcc_frame_intake: pc [n,6] fp32 rows (x y z r g b) -> keys of (0, floor x, floor y, floor z), features [n,4] = (1, r, g, b),
 *   out12 (device int32[12]): [0..3] min of (b, floored x y z), [4..7] MINUS their max, [8] != 0: rows already in
 *   canonical (strictly ascending key) order.  ws: pcc_frame_intake_ws_bytes() (per-workgroup
 *   partial results; no global atomics).
 * pcc_decode_finish: keys [n] + colour features [n,3] -> out [n,6] = (x, y, z, clamp(round(255 f), 0, 255) / 255).
 * ---------------------------------------------------------------------------------------- */
/* pcc_coords_intake_i32: int32 [n,4] rows (b, x, y, z) -> keys, out12 as above with [0] / [4] = min b / MINUS max b (the latent
 *   coordinates `decompress` is handed, model/model.py:218-229): pack + bounds + order check in one pass. */
size_t pcc_frame_intake_ws_bytes(void);
int pcc_coords_intake_i32(const int32_t* coords, int64_t n, int64_t* keys, int32_t* out12, void* ws, size_t ws_bytes,
                          void* stream);
int pcc_frame_intake(const float* pc, int64_t n, int64_t* keys, float* feats, int32_t* out12, void* ws, size_t ws_bytes,
                     void* stream);
int pcc_decode_finish(const int64_t* keys, const float* feats3, int64_t n, float* out6, void* stream);

/* ------------------------------------------------------------------------------------------
 * a4  top-k occupancy mask + pruning   (model/transforms.py:228-282, ME.MinkowskiPruning :163)
 * Total order: logit descending, then canonical row ascending.
 * seg_begin: h_ host array [nb+1] of row ranges per batch index, h_k: host array [nb].
 * ---------------------------------------------------------------------------------------- */
size_t pcc_topk_ws_bytes(int64_t n);
int pcc_topk_mask(const float* logits, int64_t stride_elems, const int64_t* h_seg_begin,
                  const int64_t* h_k, int32_t nb, uint8_t* mask, void* ws, size_t ws_bytes,
                  void* stream);
/* the same selection with the kept rows' keys compacted in the same pass (the decoder's "top-k, then prune the coordinates",
 * model/transforms.py:246-282 without the features): keys [rows], keys_out capacity sum_b min(max(k_b,0), rows_b), written
 * batch after batch in canonical order; mask as pcc_topk_mask. */
int pcc_topk_prune_keys(const float* logits, int64_t stride_elems, const int64_t* h_seg_begin,
                        const int64_t* h_k, int32_t nb, const int64_t* keys, uint8_t* mask, int64_t* keys_out,
                        void* ws, size_t ws_bytes, void* stream);
/* row compaction by mask: keys_out/feat_out capacity n; *d_count device int64 */
size_t pcc_prune_ws_bytes(int64_t n);
int pcc_prune_rows(const uint8_t* mask, int64_t n, const int64_t* keys, const float* feat, int32_t c,
                   int64_t* keys_out, float* feat_out, int64_t* d_count, void* ws, size_t ws_bytes,
                   void* stream);

/* a6  SparseTensor.features_at_coordinates for on-grid queries
 * (model/entropy_models.py:294,381,446): out[q] = feat[row(query_keys[q])] or zeros. */
int pcc_lookup_gather(const int64_t* keys, int64_t n, const float* feat, int32_t c,
                      const int64_t* query_keys, int64_t nq, float* out, void* stream);
/* row index of each query (-1 absent) */
int pcc_lookup_rows(const int64_t* keys, int64_t n, const int64_t* query_keys, int64_t nq,
                    int32_t* rows_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * a7  Gaussian conditional, fused      (compressai GaussianConditional as used at
 *                                       model/entropy_models.py:299-333,396-400,468-484)
 * y [n,c]; params [n,2c] = (scales_hat | means_hat) as h_s emits them; gain [nb,c] nullable
 * (scale_nn(q)+eps rows, indexed by the batch field of keys[]; NULL = 1).
 *   s   = max(scales*gain, 0.11)
 *   sym = rint(y*gain - means*gain)            idx = 63 - #{t<63 : s <= table[t]}
 *   lik = max(Phi((.5-|sym|)/s) - Phi((-.5-|sym|)/s), 1e-9)
 * Any of sym/idx/lik may be NULL.
 * ---------------------------------------------------------------------------------------- */
int pcc_gauss_encode(const float* y, const float* params, const int64_t* keys, const float* gain,
                     int64_t n, int32_t c, const float* table, int32_t n_table, int32_t* sym,
                     int32_t* idx, float* lik, void* stream);
/* decode side: y_hat = sym + means*gain (no offsets, model/entropy_models.py:484), idx as above */
int pcc_gauss_decode(const int32_t* sym, const float* params, const int64_t* keys, const float* gain,
                     int64_t n, int32_t c, const float* table, int32_t n_table, float* y_hat,
                     int32_t* idx, void* stream);

/* Differentiable Gaussian likelihood of the TRAINING forward (GaussianConditional._likelihood + lower bound 1e-9;
 * model/entropy_models.py:312-316,327-331; rate term loss.py:77-79) on n elements:
 *   lik = max(Phi((.5 - |v - mean|)/s) - Phi((-.5 - |v - mean|)/s), 1e-9),  s = max(scale, 0.11)
 * and its gradients w.r.t. v, scale and mean (CompressAI LowerBound gradient rule on both bounds); mean nullable. */
int pcc_gauss_lik_fwd(const float* v, const float* scale, const float* mean, int64_t n, float* lik, void* stream);
int pcc_gauss_lik_bwd(const float* v, const float* scale, const float* mean, const float* grad_lik, int64_t n,
                      float* dv /*nullable*/, float* dscale /*nullable*/, float* dmean /*nullable*/, void* stream);

/* a8  factorised prior on z            (compressai EntropyBottleneck, model/entropy_models.py:272,
 *                                       282-285,371-372,438); filters (3,3,3,3).
 * eb_packed [c,58]: softplus(matrices) (3,9,9,9,3) | biases (3,3,3,3,1) | tanh(factors) (3,3,3,3);
 * medians [c]. sym = rint(z - med), z_hat = sym + med, lik = |sig(s*u) - sig(s*l)| >= 1e-9. */
int pcc_eb_encode(const float* z, int64_t n, int32_t c, const float* eb_packed, const float* medians,
                  int32_t* sym, float* z_hat, float* lik, void* stream);

/* Differentiable factorised-prior likelihood of the TRAINING forward (EntropyBottleneck._likelihood + lower bound 1e-9;
 * model/entropy_models.py:272,282-285; rate term loss.py:77-79), [n, c] rows, filters (3,3,3,3):
 *   lik = max(|sigmoid(sg up) - sigmoid(sg lo)|, 1e-9),  lo / up = logits(v -+ .5),  sg = -sign(lo + up) (no gradient)
 * backward: dv [n, c] (nullable) and d_packed [c, 58], the gradient of the packed parameters (eb_packed layout: softplus /
 * tanh already applied -- the caller differentiates those reparametrisations), rows summed in a fixed order. */
/* Quantisation-offset network of the training forward (reference model/entropy_models.py:210-233 `quant_nn`, call sites
 * :318-322): a 2 -> 10 -> 10 -> 1 perceptron with ReLUs per element on (scale, stddev), one kernel per direction.
 * params [pcc_quant_mlp_params() = 151]: W1 [10][2] | b1 [10] | W2 [10][10] | b2 [10] | W3 [10] | b3 (torch.nn.Linear layouts).
 * bwd: d_scale / d_stddev nullable; d_params [151] summed in a fixed order (deterministic). */
/* GDN backward, element-wise parts (reference model/blocks.py:38-57; the products n = beta + |x| gamma^T, v = u gamma and
 * d gamma = u^T |x| run on the convolution / weight-gradient entry points):
 *   pre : GDN y = x / n: u = -g x / n^2, dx0 = g / n;  IGDN y = x n: u = g x, dx0 = g n        (elems = rows * C, a multiple of 4)
 *   post: dx += sign(x) v
 *   gamma_eff [C][C] = CompressAI's reparametrisation of gamma_raw (as pcc_gdn_pack applies it)
 *   reparam_bwd: gradients of the RAW parameters from those of the effective ones (LowerBound's rule); d_gamma_t is [ci][co]
 *   as pcc_conv_wgrad(|x|, u) returns it, gamma_raw / d_gamma_raw are [co][ci]. */
int pcc_gdn_bwd_pre(const float* x, const float* g, const float* n, int64_t elems, int32_t inverse, float* u, float* dx0, void* stream);
int pcc_gdn_bwd_post(float* dx, const float* x, const float* v, int64_t elems, void* stream);
int pcc_gdn_gamma_eff(const float* gamma_raw, int32_t c, float* gamma_eff, void* stream);
int pcc_gdn_reparam_bwd(const float* beta_raw, const float* gamma_raw, const float* d_beta, const float* d_gamma_t, int32_t c,
                        float beta_min, float* d_beta_raw, float* d_gamma_raw, void* stream);
/* Focal loss rows of one occupancy level (reference loss.py:115-157 Multiscale_FocalLoss): f[i] = -(occ ? alpha : 1-alpha) *
 * (1-pt)^gamma * log(pt) * q_map[batch(i)][0] with pt = clip(occ ? p : 1-p, 1e-2, 1), p = sigmoid(logit), and df[i] = d f[i] /
 * d logit[i].  occ_row: pcc_lookup_rows of the candidate keys in the ground-truth set (>= 0: occupied); keys: the candidates. */
int pcc_focal_rows(const float* logits, int64_t stride_elems, const int32_t* occ_row, const int64_t* keys, int64_t n,
                   const float* q_map, int32_t q_pitch, float alpha, float gamma, float* f, float* df, void* stream);
int32_t pcc_quant_mlp_params(void);
size_t pcc_quant_mlp_ws_bytes(int64_t n);
int pcc_quant_mlp_fwd(const float* scale, const float* stddev, int64_t n, const float* params, float* out, void* stream);
int pcc_quant_mlp_bwd(const float* scale, const float* stddev, const float* grad_out, int64_t n, const float* params,
                      float* d_scale, float* d_stddev, float* d_params, void* ws, size_t ws_bytes, void* stream);
int pcc_eb_lik_fwd(const float* v, int64_t n, int32_t c, const float* eb_packed, float* lik, void* stream);
int pcc_eb_lik_bwd(const float* v, const float* grad_lik, int64_t n, int32_t c, const float* eb_packed, float* dv,
                   float* d_packed, void* stream);

/* ------------------------------------------------------------------------------------------
 * training (BASELINE config 4; reference train.py:221-227 back-propagates through every ME convolution)
 *   data gradient  : the forward entry points on the inverse map with transposed weights (no extra kernel)
 *   weight gradient: dW[k][ci][co] = sum over pairs (i,o) of offset k of feat_in[i][ci] * grad_out[o][co]
 *                    MFMA GEMM whose reduction runs over the pair list; partial tiles per position slice are summed
 *                    in slice order (deterministic).  hdr/nbr/rows: the FORWARD map (NULL hdr: K = 1 identity).
 *   pcc_convt_scatter_rows: dT[pair] = grad_out[output row of pair] for the input-stationary transposed conv.
 * ---------------------------------------------------------------------------------------- */
size_t pcc_conv_wgrad_ws_bytes(int64_t n_out, int32_t K, int32_t cin, int32_t cout);
int pcc_conv_wgrad(const float* feat_in, int64_t n_in, int32_t cin, const float* grad_out, int64_t n_out, int32_t cout,
                   int32_t K, const int32_t* hdr, const int32_t* nbr, const int32_t* rows, float* dW /*[K,cin,cout]*/,
                   void* ws, size_t ws_bytes, void* stream);
int pcc_convt_scatter_rows(const float* grad_out, const int32_t* first, const int32_t* pair_ids, int64_t n_out,
                           int32_t cout, float* dT /*[n_in*K, cout]*/, void* stream);
/* Convolution over a set mapped onto ITSELF with 1 or 16 output channels (reference model/transforms.py:146-161, predict_*[0] /
 * predict_*[2]; odd K <= 27, stride 1; cout 1 with cin in {16,32,64}, cout 16 with cin in {16,32}): dW[k][ci][co] = sum_i feat[i][ci] * grad_out[nbr_{K-1-k}(i)][co] --
 * input-stationary, the feature rows are streamed once, only the thin gradient rows are gathered.  hdr/nbr: the map of the set
 * onto ITSELF (the caller guarantees in_set == out_set). */
int pcc_conv_wgrad_self_supported(int32_t K, int32_t cin, int32_t cout);
size_t pcc_conv_wgrad_self_ws_bytes(int64_t n, int32_t K, int32_t cin, int32_t cout);
int pcc_conv_wgrad_self(const float* feat, int64_t n, int32_t cin, const float* grad_out /*[n,cout]*/, int32_t cout, int32_t K,
                        const int32_t* hdr, const int32_t* nbr, float* dW /*[K,cin,cout]*/, void* ws, size_t ws_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * 8f-1  rANS entropy coder + CDF tables  (CompressAI `_CXX`: `pmf_to_quantized_cdf`, `BufferedRansEncoder`,
 *       `RansDecoder`; reference call sites model/entropy_models.py:371-372,397-400,438,471,484, model/model.py:30-34)
 * rans64 scheme: 64-bit state, 32-bit words, 16-bit probabilities, 4-bit bypass digits outside a table.
 * Tables: cdf [rows, cdf_stride] int32, sizes[rows] (= pmf length + 2), offsets[rows]; a symbol with table row r is
 * coded as value = symbol - offsets[r] in [0, sizes[r]-2), anything else through the bypass sentinel.
 * ---------------------------------------------------------------------------------------- */
int pcc_pmf_to_quantized_cdf(const float* h_pmf, int32_t n, int32_t precision, int32_t* h_cdf /*[n+1]*/);
/* single stream on HOST buffers: the byte layout of BufferedRansEncoder.flush() */
int64_t pcc_rans_max_bytes(int64_t n);
int pcc_rans_encode_host(const int32_t* h_sym, const int32_t* h_idx, int64_t n, const int32_t* h_cdf,
                         int32_t cdf_stride, const int32_t* h_sizes, const int32_t* h_offsets, uint8_t* h_out,
                         int64_t cap, int64_t* h_nbytes);
int pcc_rans_decode_host(const uint8_t* h_data, int64_t nbytes, const int32_t* h_idx, int64_t n,
                         const int32_t* h_cdf, int32_t cdf_stride, const int32_t* h_sizes, const int32_t* h_offsets,
                         int32_t* h_sym);
/* GPU, n_groups * n_segments independent streams over a row-major [n, channels] int32 symbol matrix.  The matrix is
 * cut into n_groups channel groups (channels/n_groups adjacent channels each, a power of two) and n_segments row
 * segments of R = ceil(n / n_segments) rows; stream s = segment * n_groups + group codes its tile row by row, with table
 * row idx[same element] (idx NULL: row = channel, the factorised prior).  Container (device buffer `out`, capacity
 * pcc_rans_container_max_bytes(pcc_rans_stream_symbols(...), n_groups * n_segments)):
 *   u32 n_streams | u32 nwords[n_streams] | stream words.
 * Every stream is byte-identical to the host single-stream coder run on that tile.  With n_groups = channels and
 * n_segments = 1 the concatenation of the streams is the reference's [1,C,N] channel-major order.  Each stream costs
 * 12 bytes of framing: few streams for rate, many for speed (one lane per stream).  Limits: 65536 streams,
 * n * channels < 2^31. */
int64_t pcc_rans_stream_symbols(int64_t n, int32_t channels, int32_t n_groups, int32_t n_segments); /* longest stream; -1: bad geometry */
int64_t pcc_rans_container_max_bytes(int64_t stream_symbols, int32_t n_streams);
size_t pcc_rans_streams_ws_bytes(int64_t stream_symbols, int32_t n_streams);
/* enc_table (device, required): division-free entries from pcc_rans_build_enc_table (16 bytes per (row, value),
 * layout [rows][cdf_stride]). */
int pcc_rans_build_enc_table(const int32_t* h_cdf, int32_t rows, int32_t cdf_stride, const int32_t* h_sizes,
                             void* h_table /*16*rows*cdf_stride bytes*/);
/* Payload estimate (in 1/256 bit) of a symbol matrix under the tables, independent of any stream geometry and of the
 * order of evaluation (integer sum): the encoder sizes its stream count from it (framing <= 2 % of the payload). */
int pcc_rans_estimate_bits(const int32_t* sym, const int32_t* idx, int64_t n, int32_t channels, const int32_t* cdf,
                           int32_t cdf_stride, const int32_t* sizes, const int32_t* offsets, int64_t* d_bits256 /*device*/,
                           void* stream);
int pcc_rans_encode_streams(const int32_t* sym, const int32_t* idx, int64_t n, int32_t channels,
                            int32_t n_groups, int32_t n_segments, const int32_t* cdf, int32_t cdf_stride,
                            const int32_t* sizes, const int32_t* offsets, const void* enc_table, uint8_t* out,
                            int64_t* d_nbytes, void* ws, size_t ws_bytes, void* stream);
/* compact decoder table (host): 16-bit CDF rows back to back + a 256-bucket start table per row, one blob of
 * pcc_rans_dec_table_bytes() bytes; kept in LDS by pcc_rans_decode_streams when it fits (<= 150 KB), so the per-symbol
 * CDF search runs at LDS latency.  dec_table may be NULL (binary search over the int32 tables in global memory). */
int64_t pcc_rans_dec_table_bytes(int32_t rows, const int32_t* h_sizes);
int pcc_rans_build_dec_table(const int32_t* h_cdf, int32_t rows, int32_t cdf_stride, const int32_t* h_sizes,
                             void* h_blob);
/* *d_status != 0 after the kernel: malformed container.  `data` must be readable 8 bytes past nbytes (look-ahead). */
int pcc_rans_decode_streams(const uint8_t* data, int64_t nbytes, const int32_t* idx, int64_t n, int32_t channels,
                            int32_t n_groups, int32_t n_segments, const int32_t* cdf, int32_t cdf_stride,
                            const int32_t* sizes, const int32_t* offsets, const void* dec_table /*nullable, device*/,
                            int64_t dec_bytes, int32_t* sym_out, int32_t* d_status, void* stream);

/* ------------------------------------------------------------------------------------------
 * 8f-3  lossless coder for the stride-8 latent coordinates (replaces the PLY file + `tmc3` subprocess round trip of
 *       UnifiedModel.gpcc_encode / gpcc_decode, model/model.py:388-486).  Host buffers.  Octree occupancy bits under an
 *       adaptive binary range coder; stream = u32 n | u8 depth | payload.  Cells are (x,y,z) in [0, 2^depth), unique;
 *       the decoder returns them in Morton order (x most significant).  Not G-PCC syntax; lossless.
 * ---------------------------------------------------------------------------------------- */
int64_t pcc_octree_max_bytes(int64_t n, int32_t depth);
int pcc_octree_encode_host(const int32_t* h_cells /*[n,3]*/, int64_t n, int32_t depth, uint8_t* h_out, int64_t cap,
                           int64_t* h_nbytes);
/* h_cells NULL: only *h_n / *h_depth are returned (size query) */
int pcc_octree_decode_host(const uint8_t* h_data, int64_t nbytes, int32_t* h_cells, int64_t cap_points, int64_t* h_n,
                           int32_t* h_depth);

/* ------------------------------------------------------------------------------------------
 * 8f-4  nearest-neighbour association of the distortion report (replaces the two Open3D KD-trees and the per-point
 *       Python loop of PointCloudMetric.__init__, metrics/metric.py:36-43).  Exact, integer coordinates.
 *       a_xyz [n_a,3] int32 queries (any order; canonical order is fastest); b_xyz [n_b,3] int32 SORTED BY x ascending
 *       (canonical key order is).  d2[i] = min_j |a_i - b_j|^2, nn[i] = the smallest such j.
 * ---------------------------------------------------------------------------------------- */
int pcc_nn_sorted_x(const int32_t* a_xyz, int64_t n_a, const int32_t* b_xyz, int64_t n_b, int64_t* d2, int32_t* nn,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * measurement support: per-launch HIP-event timing of the conv kernel (bench.py roofline)
 * ---------------------------------------------------------------------------------------- */
int pcc_prof_enable(int32_t on);
/* sums over launches since the last reset; also resets. flops = 2*P*Cin*Cout needs the pair
 * counts, which the caller accumulates itself (d_pairs). Synchronises the recorded events. */
int pcc_prof_collect(double* h_conv_ms, int64_t* h_conv_launches);
/* The same timings split by the kernel form each launch actually took (so that a report never has to mirror the dispatch
 * rules): arrays of PCC_FORM_COUNT entries.  flops / bytes: algorithmic work of the dense products (rows x columns x depth,
 * 4 bytes per operand / result element), 0 for the gathered forms whose pair counts live on the device.  Also resets. */
#define PCC_FORM_OTHER 0      /* anything else that is event-timed */
#define PCC_FORM_GEMM_H2 1    /* k_gemm_h2: dense products, scaled fp16 pairs, 3 MFMA terms */
#define PCC_FORM_GEMM_BF2 2   /* k_gemm_bf2: dense products, bf16 split, 6 terms */
#define PCC_FORM_PAIR_H2 3    /* k_pair_h2: gathered pair GEMM, 3 terms */
#define PCC_FORM_PAIR_BF 4    /* gathered pair GEMM, 6 terms */
#define PCC_FORM_CONV_BF 5    /* k_conv_mfma_bf: gathered output-stationary convolution, 6 terms */
#define PCC_FORM_CONV_F32 6   /* k_conv_mfma: fp32-input MFMA */
#define PCC_FORM_WAVE16 7     /* k_conv_wave16 / wave16z: narrow outputs, fp32-input 16x16x4 MFMA */
#define PCC_FORM_GATHER_CSR 8 /* k_convt_gather_csr of a composite level (pcc_convt_fwd_csr_grid): not an MFMA launch; timed because
                                 dense products + gather-sum are ONE unit of SURVEY 8d's accounting */
#define PCC_FORM_COUNT 9
int pcc_prof_collect_forms(double* h_ms, int64_t* h_launches, double* h_flops, double* h_bytes);
/* forms of the timed launches recorded so far, in launch order (up to cap entries); returns their number; no reset */
int64_t pcc_prof_sequence(int32_t* h_forms, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PCC_HIP_H */
