// oracle/primref.hip -- TEST INFRASTRUCTURE ONLY (never linked into or loaded by the product).
//
// The two library algorithms the reference leans on, called the way the reference calls them, through hipCUB (the
// image's own implementation of the CUB interface on top of rocPRIM):
//   cub::DevicePartition::Flagged            tt_embeddings_cuda.cu:1437-1478 (colidx, rowidx, cache_locations)
//   cub::DeviceRadixSort::SortPairsDescending tt_embeddings_cuda.cu:1280-1308 (int64 keys = cache_freq, int64 values =
//                                             hashtbl, bits [0, 64))
// The product replaces both with its own kernels (csrc/ttx_cache.hip: rowidx_update / partition_scatter, radix_*); the
// -m gpu tests of tests/test_primref_gpu.py compare them with these calls, so rows a6 / a13 are pinned to a library's
// statement of the contract (selected items in order at the front, rejected items REVERSED at the rear; a stable
// descending sort), not only to a paragraph restating it.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

template <typename T>
static int partition_flagged(const T* in, const bool* flags, T* out, int32_t* num_selected_dev, int n) {
  size_t bytes = 0;
  hipError_t e = hipcub::DevicePartition::Flagged(nullptr, bytes, in, flags, out, num_selected_dev, n, 0);
  if (e != hipSuccess) return (int)e;
  void* tmp = nullptr;
  if ((e = hipMalloc(&tmp, bytes ? bytes : 1)) != hipSuccess) return (int)e;
  e = hipcub::DevicePartition::Flagged(tmp, bytes, in, flags, out, num_selected_dev, n, 0);
  hipError_t e2 = hipDeviceSynchronize();
  (void)hipFree(tmp);
  return (int)(e != hipSuccess ? e : e2);
}

extern "C" {

int primref_partition_flagged_i64(const int64_t* in, const bool* flags, int64_t* out, int32_t* num_selected_dev, int n) {
  return partition_flagged<int64_t>(in, flags, out, num_selected_dev, n);
}

int primref_partition_flagged_i32(const int32_t* in, const bool* flags, int32_t* out, int32_t* num_selected_dev, int n) {
  return partition_flagged<int32_t>(in, flags, out, num_selected_dev, n);
}

int primref_sort_pairs_desc_i64(const int64_t* keys_in, int64_t* keys_out, const int64_t* vals_in, int64_t* vals_out, int n) {
  size_t bytes = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0,
                                                              (int)sizeof(int64_t) * 8, 0);
  if (e != hipSuccess) return (int)e;
  void* tmp = nullptr;
  if ((e = hipMalloc(&tmp, bytes ? bytes : 1)) != hipSuccess) return (int)e;
  e = hipcub::DeviceRadixSort::SortPairsDescending(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0,
                                                   (int)sizeof(int64_t) * 8, 0);
  hipError_t e2 = hipDeviceSynchronize();
  (void)hipFree(tmp);
  return (int)(e != hipSuccess ? e : e2);
}

}  // extern "C"
