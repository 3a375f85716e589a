/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's two
 * native preprocessing operators.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path never
 * does (it fails loudly when the HIP extension is missing).
 *
 * Parity status: PINNED.  Both functions are checked in tests/test_oracle_*.py
 * against (a) oracle/_ref (the reference's own C++ compiled in place) when
 * /root/reference is present and (b) the committed fixtures under
 * tests/golden/ that were generated from it (oracle/gen_golden.py).
 *
 * Plain C11, single threaded like the reference extension, compiled with
 * -ffp-contract=off so every float32 operation rounds exactly as in the
 * reference's default x86-64 build.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* libstdc++ std::unordered_map<size_t, T> iteration-order model              */
/* ------------------------------------------------------------------------- */
/*
 * The reference emits subsampled points by iterating a
 * std::unordered_map<size_t, SampledData>
 * (cpp_subsampling/grid_subsampling/grid_subsampling.cpp:48,85), so its output
 * ORDER is the libstdc++ hashtable's node-list order.  That order is a pure
 * function of the sequence of distinct keys in first-insertion order:
 *
 *  - identity hash, bucket = key % bucket_count;
 *  - a node whose bucket is empty is linked at the global list head, otherwise
 *    directly after the bucket's "before" node, i.e. at the head of the
 *    bucket's chain (hashtable.h: _M_insert_bucket_begin);
 *  - when size+1 would exceed bucket_count (max_load_factor 1.0) the table is
 *    rehashed BEFORE the insertion to the next entry of the prime schedule
 *    below, re-linking the nodes in current list order with the same rule
 *    (hashtable.h: _M_rehash_aux, unique keys).
 *
 * Bucket schedule of libstdc++ (GCC 11.4, _Prime_rehash_policy, growth 2):
 * probed in this container with a real std::unordered_map<size_t,int>; the
 * test-suite re-checks it against oracle/_ref.
 */
static const size_t k_bucket_schedule[] = {
    1,      13,     29,      59,      127,     257,     541,     1109,
    2357,   5087,   10273,   20753,   42043,   85229,   172933,  351061,
    712697, 1447153, 2938679, 5967347, 12117689, 24607243, 49969847};
#define K_NSCHED (sizeof(k_bucket_schedule) / sizeof(k_bucket_schedule[0]))

typedef struct {
  size_t nb;       /* bucket count                                          */
  int sched;       /* index into k_bucket_schedule                          */
  int* before;     /* per bucket: node before its first node; -1 = list     */
                   /* head sentinel (before_begin); -2 = empty bucket       */
  int* next;       /* per node: next node or -1                             */
  const uint64_t* keys; /* per node key                                     */
  int head;        /* first node of the list or -1                          */
  int size;
} umap_model;

static void umap_link(umap_model* m, int node) {
  size_t b = (size_t)(m->keys[node] % m->nb);
  if (m->before[b] != -2) {
    int bef = m->before[b];
    if (bef == -1) {
      m->next[node] = m->head;
      m->head = node;
    } else {
      m->next[node] = m->next[bef];
      m->next[bef] = node;
    }
  } else {
    m->next[node] = m->head;
    m->head = node;
    if (m->next[node] != -1) {
      size_t ob = (size_t)(m->keys[m->next[node]] % m->nb);
      m->before[ob] = node;
    }
    m->before[b] = -1;
  }
}

static int umap_rehash(umap_model* m, size_t nb_new) {
  int* nb = (int*)malloc(sizeof(int) * nb_new);
  if (!nb) return -1;
  for (size_t i = 0; i < nb_new; ++i) nb[i] = -2;
  free(m->before);
  m->before = nb;
  m->nb = nb_new;
  int p = m->head;
  m->head = -1;
  while (p != -1) {
    int nx = m->next[p];
    umap_link(m, p);
    p = nx;
  }
  return 0;
}

/* Insert node `node` (a NEW distinct key).  Mirrors _M_insert_unique_node. */
static int umap_insert(umap_model* m, int node) {
  /* _M_need_rehash(n_bkt, n_elt, 1): the first insertion goes 1 -> 13. */
  if ((size_t)m->size + 1 > m->nb || m->size == 0) {
    if ((size_t)(m->sched + 1) >= K_NSCHED) return -1;
    m->sched += 1;
    if (umap_rehash(m, k_bucket_schedule[m->sched])) return -1;
  }
  umap_link(m, node);
  m->size += 1;
  return 0;
}

/* Iteration order of a libstdc++ unordered_map after inserting the n distinct
 * keys keys[0..n) in this order.  order[i] = index (into keys) of the i-th
 * element visited.  Returns 0 on success. */
int spr_oracle_umap_order(const uint64_t* keys, int n, int* order) {
  umap_model m;
  m.nb = 1;
  m.sched = 0;
  m.before = (int*)malloc(sizeof(int));
  m.next = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  m.keys = keys;
  m.head = -1;
  m.size = 0;
  if (!m.before || !m.next) return -1;
  m.before[0] = -2;
  for (int i = 0; i < n; ++i)
    if (umap_insert(&m, i)) {
      free(m.before);
      free(m.next);
      return -1;
    }
  int p = m.head, k = 0;
  while (p != -1) {
    order[k++] = p;
    p = m.next[p];
  }
  free(m.before);
  free(m.next);
  return k == n ? 0 : -1;
}

/* ------------------------------------------------------------------------- */
/* grid subsampling                                                           */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint64_t key;
  int idx;
} key_idx;

static int cmp_key_idx(const void* a, const void* b) {
  const key_idx* x = (const key_idx*)a;
  const key_idx* y = (const key_idx*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/*
 * One cloud.  Follows grid_subsampling()
 * (cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106):
 *   origin = floor(min * (1/dl)) * dl                       (:27)
 *   nx = floor((max.x - origin.x)/dl) + 1, ny likewise      (:30-31)
 *   key = ix + nx*iy + nx*ny*iz, i* = floor((p - origin)/dl) (:53-56)
 *   barycentre = (in-order float32 sum) * (float)(1.0/count) (:87,
 *                grid_subsampling.h:74-79, cloud.h operator*)
 * order_mode 0: reference order (libstdc++ unordered_map iteration order)
 * order_mode 1: canonical order (ascending voxel key)
 * Optionally returns per output voxel: its key and the index (cloud-local) of
 * its first point.
 */
static long subsample_one(const float* p, int n, float dl, int order_mode,
                          float* out, uint64_t* out_keys, int* out_first) {
  if (n <= 0) return 0;
  float mn[3] = {p[0], p[1], p[2]}, mx[3] = {p[0], p[1], p[2]};
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      float v = p[3 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  float inv = 1 / dl; /* float division, as (1/sampleDl) in the reference */
  float org[3];
  for (int d = 0; d < 3; ++d) org[d] = floorf(mn[d] * inv) * dl;
  size_t nx = (size_t)floorf((mx[0] - org[0]) / dl) + 1;
  size_t ny = (size_t)floorf((mx[1] - org[1]) / dl) + 1;

  key_idx* ki = (key_idx*)malloc(sizeof(key_idx) * (size_t)n);
  if (!ki) return -1;
  for (int i = 0; i < n; ++i) {
    size_t ix = (size_t)floorf((p[3 * i + 0] - org[0]) / dl);
    size_t iy = (size_t)floorf((p[3 * i + 1] - org[1]) / dl);
    size_t iz = (size_t)floorf((p[3 * i + 2] - org[2]) / dl);
    ki[i].key = (uint64_t)(ix + nx * iy + nx * ny * iz);
    ki[i].idx = i;
  }
  qsort(ki, (size_t)n, sizeof(key_idx), cmp_key_idx);

  /* voxels in ascending key order; sums in original point order */
  int nv = 0;
  for (int i = 0; i < n; ++i)
    if (i == 0 || ki[i].key != ki[i - 1].key) nv++;
  float* bary = (float*)malloc(sizeof(float) * 3 * (size_t)nv);
  uint64_t* vkey = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)nv);
  int* vfirst = (int*)malloc(sizeof(int) * (size_t)nv);
  if (!bary || !vkey || !vfirst) return -1;
  int v = -1, cnt = 0;
  float s[3] = {0, 0, 0};
  for (int i = 0; i <= n; ++i) {
    if (i == n || i == 0 || ki[i].key != ki[i - 1].key) {
      if (v >= 0) {
        float a = (float)(1.0 / cnt);
        bary[3 * v + 0] = s[0] * a;
        bary[3 * v + 1] = s[1] * a;
        bary[3 * v + 2] = s[2] * a;
      }
      if (i == n) break;
      v++;
      vkey[v] = ki[i].key;
      vfirst[v] = ki[i].idx;
      s[0] = s[1] = s[2] = 0.f;
      cnt = 0;
    }
    const float* q = p + 3 * ki[i].idx;
    s[0] += q[0];
    s[1] += q[1];
    s[2] += q[2];
    cnt++;
  }

  int* perm = (int*)malloc(sizeof(int) * (size_t)nv);
  if (!perm) return -1;
  if (order_mode == 0) {
    /* distinct keys in first-occurrence order -> unordered_map order */
    key_idx* fo = (key_idx*)malloc(sizeof(key_idx) * (size_t)nv);
    uint64_t* fkeys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)nv);
    int* ord = (int*)malloc(sizeof(int) * (size_t)nv);
    if (!fo || !fkeys || !ord) return -1;
    for (int i = 0; i < nv; ++i) {
      fo[i].key = (uint64_t)vfirst[i];
      fo[i].idx = i;
    }
    qsort(fo, (size_t)nv, sizeof(key_idx), cmp_key_idx);
    for (int i = 0; i < nv; ++i) fkeys[i] = vkey[fo[i].idx];
    if (spr_oracle_umap_order(fkeys, nv, ord)) return -1;
    for (int i = 0; i < nv; ++i) perm[i] = fo[ord[i]].idx;
    free(fo);
    free(fkeys);
    free(ord);
  } else {
    for (int i = 0; i < nv; ++i) perm[i] = i;
  }
  for (int i = 0; i < nv; ++i) {
    memcpy(out + 3 * i, bary + 3 * perm[i], sizeof(float) * 3);
    if (out_keys) out_keys[i] = vkey[perm[i]];
    if (out_first) out_first[i] = vfirst[perm[i]];
  }
  free(ki);
  free(bary);
  free(vkey);
  free(vfirst);
  free(perm);
  return nv;
}

/*
 * Batched version: batch_grid_subsampling()
 * (cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-204), including
 * the max_p truncation (:181-204; max_p < 1 means "no limit", :131-132).
 * out_pts must hold 3*n floats, out_lens nb ints; out_keys / out_first may be
 * NULL (n entries otherwise).  Returns the number of output points or -1.
 */
long spr_oracle_grid_subsample(const float* pts, int n, const int* lens, int nb,
                               float dl, int max_p, int order_mode,
                               float* out_pts, int* out_lens,
                               uint64_t* out_keys, int* out_first) {
  long total = 0;
  int off = 0;
  if (max_p < 1) max_p = n;
  for (int b = 0; b < nb; ++b) {
    float* tmp = (float*)malloc(sizeof(float) * 3 * (size_t)(lens[b] + 1));
    uint64_t* tk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(lens[b] + 1));
    int* tf = (int*)malloc(sizeof(int) * (size_t)(lens[b] + 1));
    if (!tmp || !tk || !tf) return -1;
    long nv = subsample_one(pts + 3 * (size_t)off, lens[b], dl, order_mode, tmp,
                            tk, tf);
    if (nv < 0) return -1;
    if (nv > max_p) nv = max_p;
    memcpy(out_pts + 3 * total, tmp, sizeof(float) * 3 * (size_t)nv);
    if (out_keys) memcpy(out_keys + total, tk, sizeof(uint64_t) * (size_t)nv);
    if (out_first) memcpy(out_first + total, tf, sizeof(int) * (size_t)nv);
    out_lens[b] = (int)nv;
    total += nv;
    off += lens[b];
    free(tmp);
    free(tk);
    free(tf);
  }
  (void)n;
  return total;
}

/* ------------------------------------------------------------------------- */
/* radius neighbours                                                          */
/* ------------------------------------------------------------------------- */
typedef struct {
  float d2;
  int idx;
} dist_idx;

static int cmp_dist_idx(const void* a, const void* b) {
  const dist_idx* x = (const dist_idx*)a;
  const dist_idx* y = (const dist_idx*)b;
  if (x->d2 != y->d2) return x->d2 < y->d2 ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/*
 * Brute-force restatement of batch_nanoflann_neighbors()
 * (cpp_neighbors/neighbors/neighbors.cpp:211-332):
 *   r2 = radius*radius in float32                                   (:226)
 *   d2 = ((qx-sx)^2 + (qy-sy)^2) + (qz-sz)^2, float32, query minus support,
 *        accumulated in x,y,z order   (nanoflann.hpp:432-440 evalMetric)
 *   keep d2 < r2 (strict)                          (nanoflann.hpp:249)
 *   sort ascending by d2                           (nanoflann.hpp:1287)
 *   global index = local + cloud offset; pad with Ns_total  (:309-327)
 *   row width = max count over ALL queries                    (:296,:304)
 * The reference sorts with std::sort on d2 only, so rows containing EXACT
 * float32 d2 ties may be permuted inside a tie run; this restatement (and the
 * HIP kernel) break ties by ascending index.  Tests canonicalise tie runs.
 *
 * Two-call protocol: call with out == NULL to obtain max_count (return
 * value), then with out of size nq*max_count.  `limit` > 0 additionally keeps
 * only the first `limit` columns (kpconv.py:259-260); the returned width is
 * then min(max_count, limit).  *max_count_out always gets the untruncated max.
 */
long spr_oracle_radius_neighbors(const float* q, int nq, const float* s, int ns,
                                 const int* qb, const int* sb, int nb,
                                 float radius, int limit, int* out,
                                 int* max_count_out) {
  float r2 = radius * radius;
  int cap = 64;
  dist_idx* buf = (dist_idx*)malloc(sizeof(dist_idx) * (size_t)cap);
  if (!buf) return -1;
  int max_count = 0;
  int width = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int qoff = 0, soff = 0;
    for (int b = 0; b < nb; ++b) {
      for (int i = qoff; i < qoff + qb[b]; ++i) {
        int cnt = 0;
        const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
        for (int j = soff; j < soff + sb[b]; ++j) {
          float dx = qx - s[3 * j], dy = qy - s[3 * j + 1],
                dz = qz - s[3 * j + 2];
          float d2 = 0.f;
          d2 += dx * dx;
          d2 += dy * dy;
          d2 += dz * dz;
          if (d2 < r2) {
            if (cnt == cap) {
              cap *= 2;
              buf = (dist_idx*)realloc(buf, sizeof(dist_idx) * (size_t)cap);
              if (!buf) return -1;
            }
            buf[cnt].d2 = d2;
            buf[cnt].idx = j;
            cnt++;
          }
        }
        if (pass == 0) {
          if (cnt > max_count) max_count = cnt;
        } else {
          qsort(buf, (size_t)cnt, sizeof(dist_idx), cmp_dist_idx);
          for (int k = 0; k < width; ++k)
            out[(size_t)i * width + k] = k < cnt ? buf[k].idx : ns;
        }
      }
      qoff += qb[b];
      soff += sb[b];
    }
    if (pass == 0) {
      width = (limit > 0 && limit < max_count) ? limit : max_count;
      if (max_count_out) *max_count_out = max_count;
      if (!out) break;
    }
  }
  free(buf);
  (void)ns;
  return width;
}
