// kmg_kernels.h -- launchers of the gfx950 kernels (internal to libkmeans_hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmg {

// Device centroid entry: (L, a, b, C = sqrt(a^2 + b^2)).  C is hoisted out of the per-pixel loop
// (delta_e.wgsl:9 recomputes it for every pixel-centroid pair).
struct alignas(16) Centroid { float L, a, b, C; };

constexpr int kBlock = 256;          // threads per workgroup (4 waves of 64)

// What the launchers size grids and LDS requests by, per device ORDINAL: a process may drive several devices (kmg_group's worker
// threads) and they need not be alike (a partitioned or mixed node).  Of the calling thread's current device; read once per ordinal.
struct DeviceInfo { uint32_t cus; size_t lds_max; };
const DeviceInfo &device_info();



// Number of workgroups the assign/accumulate pass uses for n pixels (also the number of rows of
// the per-workgroup partial-sum slab).
uint32_t assign_grid(uint64_t n_pixels);

hipError_t launch_rgb_to_lab(const uint32_t *rgba, uint64_t n, const float *lut, float *lab3,
                             hipStream_t st);

// labels and/or partial sums.  partials: [assign_grid(n)][k][4] int64, fully overwritten.
hipError_t launch_assign(const uint32_t *rgba, uint64_t n, const Centroid *cent, uint32_t k,
                         const float *lut, uint32_t *labels, int64_t *partials, hipStream_t st);

// acc[k][4] = sum over rows of partials
hipError_t launch_reduce_partials(const int64_t *partials, uint32_t rows, uint32_t k,
                                  int64_t *acc, hipStream_t st);

// choose_centroid.wgsl `pick` for all k
// Small images (assign_loop_fits): one launch per Lloyd iteration -- every workgroup updates its own copy of the centroids from
// acc_in (do_update; workgroup 0 writes them to cent_out and the convergence count to n_converged), assigns, adds its sums to
// acc_out (zero on entry) and workgroup 0 clears acc_clear.  The caller rotates three k x 4 sum buffers and two centroid
// buffers (assign_loop_scratch_bytes(k) bytes hold the three sum buffers and the second centroid buffer).
bool assign_loop_fits(uint64_t n_pixels);
size_t assign_loop_scratch_bytes(uint32_t k);
hipError_t launch_assign_loop(const uint32_t *rgba, uint64_t n, const Centroid *cent, Centroid *cent_out, uint32_t k, const float *lut,
                              uint32_t *labels, const int64_t *acc_in, int64_t *acc_out, int64_t *acc_clear, int do_update,
                              float convergence, uint32_t *n_converged, hipStream_t st);
// rows x k x 4 values small enough (reduce_update_fits): reduction and -- do_update -- the update in one launch of one workgroup
bool reduce_update_fits(uint32_t rows, uint32_t k);
hipError_t launch_reduce_update(const int64_t *partials, uint32_t rows, uint32_t k, int64_t *acc, int do_update, float convergence,
                                Centroid *cent, uint32_t *n_converged, hipStream_t st);
hipError_t launch_update(const int64_t *acc, uint32_t k, float convergence, Centroid *cent,
                         uint32_t *n_converged, hipStream_t st);

// farthest-point init
hipError_t launch_init_first(const uint32_t *rgba, uint64_t index, const float *lut,
                             Centroid *cent, unsigned long long *key, hipStream_t st);
// pick = false: pass j into *key (atomicMax).  pick = true (whole image, one device): `key` = init_slots_bytes() of slots; the
// launch first picks centroid j - 1 from the slots of launch j - 1 (j >= 2), runs pass j and leaves its keys in the other slot
// set; launch_init_pick_slots picks the last centroid (j = k - 1) from the slots of the last pass
hipError_t launch_init_pass(const uint32_t *rgba, uint64_t n, const float *lut,
                            Centroid *cent, uint32_t j, float *dist,
                            unsigned long long *key, uint64_t first_index, hipStream_t st, bool pick = false);
size_t init_slots_bytes();
// Whole image on one device, SEVERAL centroids per launch (kmg_kernels.hip k_init_multi): launch L = 1, 2, ... picks up to four
// centroids from what launch L - 1 left in `scratch` (init_multi_bytes(n); centroid 0 is there: launch_init_first) and sweeps the
// running distances against all of them; a launch that finds the table complete does nothing.  *init_multi_count(scratch, n, L)
// (device memory) = centroids chosen after launch L: the host enqueues launches in chunks and reads it in between.
size_t init_multi_bytes(uint64_t n);
hipError_t launch_init_multi(const uint32_t *rgba, uint64_t n, const float *lut, Centroid *cent, uint32_t k, uint32_t launch,
                             float *dist, void *scratch, hipStream_t st);
const uint32_t *init_multi_count(const void *scratch, uint64_t n, uint32_t launch);
hipError_t launch_init_pick_slots(const uint32_t *rgba, uint64_t n, const float *lut, const unsigned long long *slots, Centroid *cent,
                                  uint32_t j, hipStream_t st);
// sharded init (row bands): publish the colour of the pixel named by an all-reduced key; set one centroid
hipError_t launch_init_pick_band(const uint32_t *rgba, uint64_t n, uint64_t first_index,
                                 const unsigned long long *key, uint32_t *colour2, hipStream_t st);
hipError_t launch_set_centroid_rgba(const uint32_t *colour, const float *lut, Centroid *cent, uint32_t j,
                                    hipStream_t st);

hipError_t launch_resize(const uint32_t *rgba, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh,
                         uint32_t *out, hipStream_t st);
// The same resize for a row band of a sharded image: `band` holds the image rows from src_row0 on (at least every source row
// the requested output rows sample: resize_source_row(gy) and the row below it), `out` receives output rows
// [out_row0, out_row0 + out_rows).  Same values as launch_resize of the whole image, row for row.
hipError_t launch_resize_band(const uint32_t *band, uint32_t w, uint32_t h, uint32_t src_row0, uint32_t nw, uint32_t nh,
                              uint32_t out_row0, uint32_t out_rows, uint32_t *out, hipStream_t st);
uint32_t resize_source_row(uint32_t gy, uint32_t h, uint32_t nh);

// replace / dither output pass.  pal: k+1 RGBA8 words (entry k = the converted sentinel).
hipError_t launch_apply(const uint32_t *rgba, uint32_t w, uint32_t rows, uint32_t row0,
                        const Centroid *cent, uint32_t k, const float *lut, const uint32_t *pal,
                        bool dither, float threshold, uint32_t *out, hipStream_t st);

// thr[256] (device): the linear-channel thresholds of the 256 sRGB8 output bytes (k_meld), made by the device's own encode
hipError_t launch_encode_thresholds(float *thr, hipStream_t st);
// out[0..2] (device; the caller sets {0, ~0, 0}): mismatches of div_const(x, c) against x / c over every binary32 x, and the
// smallest / largest |x| bit pattern among them
hipError_t launch_division_check(float c, unsigned long long *out, hipStream_t st);
// *bad (device, zeroed by the caller) += the floats (all of them, NaN aside) whose table byte is not the encode's byte
hipError_t launch_encode_check(const float *thr, unsigned long long *bad, hipStream_t st);
// meld output pass (mix_colors.wgsl main_meld + lab_to_rgb.wgsl); lut: 256 decode entries followed by the 256 thresholds
// masks: NULL, or per colour cell the candidate centroids of kmg_table.h's launch_meld_candidates
hipError_t launch_meld(const uint32_t *rgba, uint64_t n, const Centroid *cent, uint32_t k, const float *lut,
                       const uint64_t *masks, uint32_t *out, hipStream_t st);

}  // namespace kmg
