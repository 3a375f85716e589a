// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// extern "C" driver around the *unmodified* reference C++ core, compiled in
// place from /root/reference (see oracle/Makefile, target `_ref`).  It exposes
// the two native entry points of the reference on raw pointers so that ctypes
// can call them (the reference's own CPython wrappers do not compile against
// numpy 2.x):
//
//   ref_batch_query      -> batch_nanoflann_neighbors
//        (cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211,
//         called from cpp_neighbors/wrapper.cpp:198)
//   ref_subsample_batch  -> batch_grid_subsampling
//        (cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109,
//         called from cpp_subsampling/wrapper.cpp)
//
// No reference source is copied: this file only #includes the reference
// headers by path (the Makefile passes -I to the reference tree).
#include <cstring>
#include <vector>

#include "cpp_neighbors/neighbors/neighbors.h"
#include "cpp_subsampling/grid_subsampling/grid_subsampling.h"

extern "C" {

// Returns max_count (row width of the reference output).  If `out` is non-null
// and cap >= nq*max_count the [nq, max_count] int32 matrix is copied to it.
long ref_batch_query(const float* q, int nq, const float* s, int ns,
                     const int* qb, const int* sb, int nb, float radius,
                     int* out, long cap) {
  std::vector<PointXYZ> queries(nq), supports(ns);
  std::memcpy(queries.data(), q, sizeof(float) * 3 * (size_t)nq);
  std::memcpy(supports.data(), s, sizeof(float) * 3 * (size_t)ns);
  std::vector<int> q_batches(qb, qb + nb), s_batches(sb, sb + nb);
  std::vector<int> neighbors;
  batch_nanoflann_neighbors(queries, supports, q_batches, s_batches, neighbors,
                            radius);
  long max_count = nq > 0 ? (long)(neighbors.size() / (size_t)nq) : 0;
  if (out && cap >= (long)neighbors.size())
    std::memcpy(out, neighbors.data(), sizeof(int) * neighbors.size());
  return max_count;
}

// Returns the number of subsampled points; out_p must hold 3*n floats,
// out_b nb ints.
long ref_subsample_batch(const float* p, int n, const int* b, int nb, float dl,
                         int max_p, float* out_p, int* out_b) {
  std::vector<PointXYZ> original(n), subsampled;
  std::memcpy(original.data(), p, sizeof(float) * 3 * (size_t)n);
  std::vector<float> of, sf;
  std::vector<int> oc, sc;
  std::vector<int> ob(b, b + nb), sb;
  batch_grid_subsampling(original, subsampled, of, sf, oc, sc, ob, sb, dl,
                         max_p);
  std::memcpy(out_p, subsampled.data(), sizeof(float) * 3 * subsampled.size());
  for (int i = 0; i < nb; ++i) out_b[i] = sb[i];
  return (long)subsampled.size();
}

}  // extern "C"
