// pt_kernels.h -- host-callable launchers of the kernels in pt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_device.h"
#include "../../include/pbrtgpu.h"

hipError_t ptk_trace(hipStream_t st, int grid, int grid_dist, const PtScene& sc, const PtPaths& P, const PtQueues& Q, PtCounters* cnt, uint32_t* spill,
                     uint32_t spill_depth, uint32_t* err, int far = 0);
bool ptk_trace_has_far(const PtScene& sc);
hipError_t ptk_trace_batch(hipStream_t st, int grid, const PtScene& sc, uint32_t n, const float* o, const float* d, const float* tmax, pt_hit* out,
                           uint8_t* occ, int any_hit, uint32_t* ticket, PtCounters* cnt, uint32_t* spill, uint32_t spill_depth, uint32_t* err);
hipError_t ptk_gen(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, const uint32_t* pixels, uint32_t n_pix,
                   uint32_t s0, uint32_t n_samples, PtCounters* cnt);
hipError_t ptk_nee_resolve(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q);
hipError_t ptk_prep(hipStream_t st, const PtQueues& Q, int mode);
hipError_t ptk_shade(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, PtCounters* cnt, int nee_split, int local_sort = 0);
int ptk_nee_split_default();
int ptk_trace_wide();
hipError_t ptk_film(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const uint32_t* pixels, uint32_t n_pix, uint32_t n_samples,
                    float4* own, float4* spill, float* radiance_out, uint32_t s0, uint32_t spp_total);
hipError_t ptk_film_xyzw(hipStream_t st, const float4* own, const float4* spill, float4* xyzw, uint32_t n);
hipError_t ptk_film_add(hipStream_t st, float4* xyzw, const float4* other, uint32_t n);
hipError_t ptk_film_rgb(hipStream_t st, const float4* xyzw, float* rgb, uint32_t n, float scale);
hipError_t ptk_light_grid(hipStream_t st, const PtScene& sc, float* data, uint32_t n_vox, const uint32_t* vox_list = nullptr);
hipError_t ptk_grid_mark(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, int32_t* row_of, uint32_t* todo, uint32_t* todo_count);
hipError_t ptk_grid_assign(hipStream_t st, int32_t* row_of, const uint32_t* todo, uint32_t n, uint32_t row0, uint32_t* todo_count);
hipError_t ptk_camera_rays(hipStream_t st, const PtScene& sc, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, float* o, float* d,
                           float* pf);
hipError_t ptk_bsdf_eval(hipStream_t st, const PtScene& sc, uint32_t material, uint32_t n, const float* wo, const float* wi, uint32_t flags, float* f,
                         float* pdf);
hipError_t ptk_bsdf_sample(hipStream_t st, const PtScene& sc, uint32_t material, uint32_t n, const float* wo, const float* u, uint32_t flags, float* f,
                           float* wi, float* pdf, uint32_t* type);
hipError_t ptk_sobol_samples(hipStream_t st, const PtScene& sc, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, const uint32_t* dim,
                             float* out);
hipError_t ptk_ao_tag(hipStream_t st, int grid, const PtPaths& P, uint32_t n_pix, uint32_t n_paths, uint32_t s0);
hipError_t ptk_ao_rays(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, uint32_t n_paths, float4* ao_o, float4* ao_d, float* ao_w,
                       uint32_t* counter, PtCounters* cnt);
hipError_t ptk_ao_queue(hipStream_t st, const PtQueues& Q, const uint32_t* counter, uint32_t n_s);
hipError_t ptk_iota(hipStream_t st, int grid, uint32_t* out, uint32_t n);
hipError_t ptk_ao_resolve(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, uint32_t n_paths, const float* ao_w, const uint8_t* occ);
hipError_t ptk_expand_tiles(hipStream_t st, const int4* tiles, const uint32_t* tile_off, uint32_t n_tiles, int32_t sb_x0, int32_t sb_y0, uint32_t sb_w,
                            uint32_t* pixels, uint32_t* bitmap, uint32_t* err);
hipError_t ptk_wavefront_results(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, uint32_t n, const uint8_t* kind, pt_hit* out, uint8_t* occ);
int ptk_shade_prof_read(unsigned long long* out16);
// pt_raysort.hip: the shadow rays of a launch ordered by origin cell and direction octant
size_t ptk_sort_rays_temp_bytes(uint32_t cap);
hipError_t ptk_sort_shadow_rays(hipStream_t st, uint32_t* ids, uint32_t* ids_alt, uint32_t* keys, uint32_t* keys_alt, void* temp, size_t temp_bytes, uint32_t n,
                                uint32_t** sorted);
hipError_t ptk_cont_keys(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const uint32_t* list, uint32_t n, uint32_t* keys);
size_t ptk_sort_rays_keep_temp_bytes(uint32_t cap);
hipError_t ptk_sort_rays_keep(hipStream_t st, const uint32_t* ids, uint32_t* ids_out, const uint32_t* keys, uint32_t* keys_out, void* temp, size_t temp_bytes, uint32_t n);
int ptk_trace_dist_blocks_per_cu();      // blocks per CU the pooled-leaf traversal kernels were built for (LDS budget)
hipError_t ptk_rec_init(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtRec& R, uint32_t n);
hipError_t ptk_rec_enter(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, const PtQueues& Qn, const PtRec& R, PtCounters* cnt,
                         uint32_t lights_per_node);
hipError_t ptk_rec_next(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, const PtRec& R);
