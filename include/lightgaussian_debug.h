/* lightgaussian_debug.h -- diagnostic entry points of liblightgaussian_hip.so.
 *
 * NOT part of the drop-in boundary (include/lightgaussian.h): nothing the reference's FFI for this path would bind.  They exist
 * for tests/ (sort passes on their own, tile-list dumps for the two-stage-sort cross-check, the failure path of the look-back,
 * K7's wave reduction on one wave, the id of every pixel's last contributor) and tools/ (micro-benchmarks, activation probe).
 * Same conventions as the main header: extern "C", plain pointers and sizes, device pointers unless stated, `stream` = hipStream_t. */
#ifndef LIGHTGAUSSIAN_DEBUG_H
#define LIGHTGAUSSIAN_DEBUG_H

#include "lightgaussian.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostics: the fused-getter activations on their own (exp; sigmoid in two forms; normalize in four summation orders), n
 * values each: out_s [n], out_r [4][n][4], out_o [2][n] -- compared with torch's own results by tools/activation_probe.py */
int lg_debug_activations(int32_t n, const float* s, const float* r, const float* o, float* out_s, float* out_r, float* out_o, void* stream);

/* diagnostics: K4 on its own -- stable ascending sort of bits [begin_bit, end_bit) of n < 2^30 64-bit keys (keys_in preserved);
 * temp: lg_debug_sort_temp_bytes(n) device bytes */
size_t lg_debug_sort_temp_bytes(int64_t n);
int lg_debug_sort_keys(int64_t n, const uint64_t* keys_in, uint64_t* keys_out, int32_t begin_bit, int32_t end_bit, void* temp,
                       void* stream);

/* diagnostics: out_ids [H*W] uint32 = Gaussian id of every pixel's last contributor (0xFFFFFFFF: none), from the buffers a forward
 * with this view saved (geom, binning, img, num_rendered as handed to lg_backward).  n_contrib itself is a position in this
 * library's culled tile lists and cannot be compared across implementations; the id can (tests/test_gpu_full_size.py). */
int lg_debug_last_contributor(const lg_view* view, int32_t N, const void* geom, const void* binning, const void* img, int64_t num_rendered,
                              uint32_t* out_ids, void* stream);

/* diagnostics: the tile lists a forward with this view left in its binning buffer -- out_ranges [tiles][2] uint32 {begin, end} and
 * out_entries [num_rendered] uint64 sorted keys (tile | depth | Gaussian id; only the first R = end of the last non-empty tile are
 * meaningful).  The tests compare the default two-stage sort with the one-stage scheme (LG_FLAG_SORT_ALL_BITS) entry by entry. */
int lg_debug_tile_lists(const lg_view* view, const void* binning, int64_t num_rendered, uint32_t* out_ranges, uint64_t* out_entries,
                        void* stream);

/* diagnostics: the 16 per-view words a forward leaves in its binning buffer (0 = work items of the backward, 1 = longest list, 2 = segment
 * length, 3 = par_min, 4 = items of the parallel long-tile walk, 5 = pixels the significance pass resolved through its exact fix-up) */
int lg_debug_view_meta(const lg_view* view, const void* binning, int64_t num_rendered, uint32_t* out_meta16, void* stream);

/* diagnostics: the failure path of the sort's look-back -- one digit pass whose only tile has a predecessor that never publishes.
 * Must return LG_ERR_DEVICE (error word set, no hang, no silent wrong order).  temp: lg_debug_sort_temp_bytes(2 * 8192). */
int lg_debug_sort_orphan(int64_t n, const uint64_t* keys_in, uint64_t* keys_out, void* temp, void* stream);

/* diagnostics: the packed wave reduction used by the backward blend, on one wave: in [64][9] -> out [9] */
int lg_debug_reduce9(const float* in_64x9, float* out_9, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTGAUSSIAN_DEBUG_H */
