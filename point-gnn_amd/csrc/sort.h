// Device-side primitives shared by the graph builders: stable LSD radix sort of
// (key, value) pairs and an exclusive scan.  Everything is enqueued on the
// caller's stream; scratch comes from the caller's arena.
#pragma once
#include "pgnn_common.h"

namespace pgnn {

#ifndef PGNN_SORT_TILE
#define PGNN_SORT_TILE 2048
#endif
constexpr int kSortTile = PGNN_SORT_TILE;  // keys per block per pass

// bytes of scratch radix_sort_pairs needs for n pairs (excluding the ping-pong
// key/value buffers, which the caller provides)
size_t radix_sort_scratch_bytes(int64_t n);

// Stable sort of n (key, value) pairs by the low `nbits` bits of key.
// keys_a/vals_a hold the input; *_b are same-sized temporaries.  On return
// *keys_out / *vals_out point at whichever buffer holds the sorted result.
// n_dev (nullable, device memory): the pair count when only the device knows
// it; n is then the capacity the launches and the scratch are sized for, and
// min(*n_dev, n) pairs are sorted.
int radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b,
                     uint32_t *vals_b, int64_t n, int nbits, void *scratch,
                     size_t scratch_bytes, uint32_t **keys_out,
                     uint32_t **vals_out, hipStream_t stream,
                     const int32_t *n_dev = nullptr);

// out[i] = sum_{j<i} in[j] for i in [0, n]; out has n + 1 entries (the last is
// the total).  in/out may not alias.  scratch: scan_scratch_bytes(n).
size_t scan_scratch_bytes(int64_t n);
int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n,
                       void *scratch, size_t scratch_bytes,
                       hipStream_t stream);

// In-place exclusive scan of data[0..n) by one workgroup (small n).
int exclusive_scan_inplace_i32(int32_t *data, int64_t n, hipStream_t stream);

}  // namespace pgnn
