// hnsw_front.hpp -- C++ host-side mirror of the reference's module interface for the search path,
// over the C ABI of include/hnsw_mi355x.h (header only; link with libhnsw_mi355x.so).
//
//   Hnsw::Ohnsw::knn / knn_batch_bigarray / build_batch_bigarray / distance_l2   lib/ohnsw.ml:840-899
//   Hnsw::Ba::knn / knn_batch                                                    lib/hnsw.ml:763-777
//   Hnsw::Ohnsw::search_k / search_one, Hnsw::Ba::search                         lib/ohnsw.ml:492-588, lib/hnsw_algo.ml:350-437
//   Hnsw::MultiHgraph (one process, several GPUs)                                SURVEY 8e
//   Hnsw::Stats::compute (Hgraph.Stats)                                          lib/hnsw.ml:353-375
//   Hnsw::HostMat (a page-locked Lacaml-shaped matrix the device accesses directly: hnsw_host_alloc)
//
// Same names, argument meaning and error behaviour: OCaml Invalid_argument -> std::invalid_argument
// ("knn: empty hgraph", lib/ohnsw.ml:862), Failure -> std::runtime_error.  A `Mat` is a
// Lacaml.S.mat (dim x n, Fortran layout): in memory n rows of dim contiguous floats.
#pragma once
#include "../../include/hnsw_mi355x.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace Hnsw {

struct Mat {
    const float *data;
    int64_t dim2; // number of vectors (Lacaml.S.Mat.dim2)
    int32_t dim1; // dimension (Lacaml.S.Mat.dim1)
};

inline void check(int rc) {
    if (rc == HNSW_OK) return;
    const std::string msg = hnsw_last_error();
    if (rc == HNSW_ERR_BAD_ARG || rc == HNSW_ERR_EMPTY_INDEX || rc == HNSW_ERR_DEGREE_OVERFLOW) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// Flattened Ohnsw.Hgraph.t / Hnsw.Ba.Hgraph.t resident on the device.
class Hgraph {
public:
    Hgraph() = default;
    Hgraph(const Hgraph &) = delete;
    Hgraph &operator=(const Hgraph &) = delete;
    Hgraph(Hgraph &&o) noexcept : h_(o.h_), id_base_(o.id_base_), d_(o.d_) { o.h_ = nullptr; }
    ~Hgraph() { if (h_) hnsw_index_destroy(h_); }

    // flatten + upload (hnsw_index_create)
    static Hgraph create(const hnsw_index_desc &desc, int device = 0) {
        Hgraph g;
        check(hnsw_index_create(&desc, device, &g.h_));
        g.id_base_ = desc.id_base; g.d_ = desc.d;
        return g;
    }
    hnsw_index *handle() const { return h_; }
    int id_base() const { return id_base_; }
    int dim() const { return d_; }

private:
    friend struct Builder;
    hnsw_index *h_ = nullptr;
    int id_base_ = 0, d_ = 0;
public:
    static Hgraph adopt(hnsw_index *h, int id_base, int d) { Hgraph g; g.h_ = h; g.id_base_ = id_base; g.d_ = d; return g; }
};

struct value_distance { int node; float distance_to_target; }; // lib/hnsw_algo.ml:85

// A Lacaml-shaped float32 matrix (dim x n) in page-locked memory the library allocates (hnsw_host_alloc): knn_batch*
// reads such a query matrix straight from the device, without an upload step.
class HostMat {
public:
    HostMat(int32_t dim, int64_t n) : dim_(dim), n_(n) {
        void *p = nullptr;
        check(hnsw_host_alloc(&p, (int64_t)sizeof(float) * dim * (n > 0 ? n : 1)));
        data_ = static_cast<float *>(p);
    }
    HostMat(const HostMat &) = delete;
    HostMat &operator=(const HostMat &) = delete;
    ~HostMat() { if (data_) hnsw_host_free(data_); }
    float *data() { return data_; }
    float *col(int64_t j) { return data_ + j * dim_; }          // Lacaml.S.Mat.col m (j + 1)
    Mat mat() const { return Mat{data_, n_, dim_}; }
private:
    float *data_ = nullptr;
    int32_t dim_;
    int64_t n_;
};

// Hgraph.Stats (lib/hnsw.ml:353-375): per layer its size and the mima record {min; max; mean; isolated}; `isolated` is the
// reference's list -- node ids, descending (consed during the ascending fold of min_max_connectivity).
namespace Stats {
struct mima { int min, max; double mean; std::vector<int64_t> isolated; };
struct t { int64_t num_nodes; std::vector<int64_t> layer_sizes; std::vector<mima> layer_connectivity; };
inline t compute(const Hgraph &g) {
    hnsw_index_info inf{};
    check(hnsw_index_get_info(g.handle(), &inf));
    t out{inf.n, {}, {}};
    for (int layer = 0; layer <= inf.max_layer; ++layer) {
        hnsw_layer_stats s{};
        check(hnsw_index_layer_stats(g.handle(), layer, &s));
        mima m{s.min_degree, s.max_degree, s.mean_degree, std::vector<int64_t>((size_t)s.num_isolated)};
        int64_t cnt = 0;
        check(hnsw_index_layer_isolated(g.handle(), layer, m.isolated.data(), s.num_isolated, &cnt));
        out.layer_sizes.push_back(s.num_nodes);
        out.layer_connectivity.push_back(std::move(m));
    }
    return out;
}
} // namespace Stats

namespace detail {
inline void search(const Hgraph &g, const Mat &batch, int ef, int k, int fill, std::vector<int32_t> &ids, std::vector<float> &dist, int sem = HNSW_SEM_OHNSW) {
    ids.assign((size_t)batch.dim2 * k, -1);
    dist.assign((size_t)batch.dim2 * k, 0.f);
    hnsw_search_params p{ef, k, fill, sem};
    check(hnsw_search_batch(g.handle(), batch.data, batch.dim2, batch.dim1, &p, ids.data(), dist.data(), nullptr, nullptr));
}
} // namespace detail

namespace Ohnsw {

// Ohnsw.build_batch_bigarray distance batch ~num_connections ~num_nodes_search_construction
// (lib/ohnsw.ml:840-857), batched on the device.
inline Hgraph build_batch_bigarray(const Mat &batch, int num_connections, int num_nodes_search_construction,
                                   uint64_t seed = 0, int metric = HNSW_METRIC_L2, int device = 0, int expected_ef = 0) {
    hnsw_build_params p{num_connections, num_nodes_search_construction, metric, 0, seed, 0, 0, expected_ef, HNSW_SEM_OHNSW};
    hnsw_index *h = nullptr;
    check(hnsw_build(batch.data, batch.dim2, batch.dim1, batch.dim1, &p, device, &h));
    return Hgraph::adopt(h, 0, batch.dim1);
}

// Ohnsw.knn hgraph visited ~k target (lib/ohnsw.ml:859-875): the MinQueue popped ascending.
inline std::vector<value_distance> knn(const Hgraph &g, int k, const float *target) {
    std::vector<int32_t> ids; std::vector<float> dist;
    detail::search(g, Mat{target, 1, g.dim()}, k, k, HNSW_FILL_OHNSW, ids, dist);
    std::vector<value_distance> out;
    for (int i = 0; i < k && ids[(size_t)i] >= g.id_base(); ++i) out.push_back({ids[(size_t)i], dist[(size_t)i]});
    return out;
}

// Ohnsw.knn_batch_bigarray hgraph ~k batch -> (ids, distances) (lib/ohnsw.ml:877-897):
// ids nq x k (-1 filled), distances k x nq Fortran = [nq][k] (NaN filled).
inline std::pair<std::vector<std::vector<int>>, std::vector<float>> knn_batch_bigarray(const Hgraph &g, int k, const Mat &batch) {
    std::vector<int32_t> ids; std::vector<float> dist;
    detail::search(g, batch, k, k, HNSW_FILL_OHNSW, ids, dist);
    std::vector<std::vector<int>> out((size_t)batch.dim2, std::vector<int>((size_t)k));
    for (int64_t j = 0; j < batch.dim2; ++j) for (int i = 0; i < k; ++i) out[(size_t)j][(size_t)i] = ids[(size_t)(j * k + i)];
    return {std::move(out), std::move(dist)};
}

// Ohnsw.search_k layer distance value visited start_nodes target k (lib/ohnsw.ml:543-588): one target,
// the start queue as a list of nodes; result_minq popped ascending.  `sem` = HNSW_SEM_FUNCTOR gives
// Hnsw_algo.Search.search (lib/hnsw_algo.ml:350-391).
inline std::vector<value_distance> search_k(const Hgraph &g, int layer, const std::vector<int64_t> &start_nodes,
                                            const float *target, int k, int sem = HNSW_SEM_OHNSW) {
    std::vector<int32_t> ids((size_t)k); std::vector<float> dist((size_t)k);
    int32_t cnt = 0;
    hnsw_search_params p{k, k, HNSW_FILL_OHNSW, sem};
    check(hnsw_search_layer_batch(g.handle(), layer, target, 1, g.dim(), start_nodes.data(), (int32_t)start_nodes.size(), &p,
                                  ids.data(), dist.data(), &cnt, nullptr, nullptr));
    std::vector<value_distance> out;
    for (int i = 0; i < cnt; ++i) out.push_back({ids[(size_t)i], dist[(size_t)i]});
    return out;
}

// Ohnsw.search_one layer distance value visited start_node target (lib/ohnsw.ml:492-512)
inline value_distance search_one(const Hgraph &g, int layer, int64_t start_node, const float *target) {
    int64_t node = 0; float d = 0.f;
    check(hnsw_search_one_batch(g.handle(), layer, target, 1, g.dim(), &start_node, &node, &d));
    return {(int)node, d};
}

// Ohnsw.distance_l2 a b (lib/ohnsw.ml:899), batched: out[q][j] = distance(batch[q], value ids[q][j])
inline std::vector<float> distance_l2(const Hgraph &g, const Mat &batch, const int32_t *ids, int m) {
    std::vector<float> out((size_t)batch.dim2 * m);
    check(hnsw_distance_batch(g.handle(), batch.data, batch.dim2, batch.dim1, ids, m, out.data()));
    return out;
}

} // namespace Ohnsw

namespace Ba {

// Hnsw.Ba.knn hgraph point ~num_neighbours_search ~num_neighbours (lib/hnsw.ml:763-767)
inline std::vector<value_distance> knn(const Hgraph &g, const float *point, int num_neighbours_search, int num_neighbours) {
    std::vector<int32_t> ids; std::vector<float> dist;
    detail::search(g, Mat{point, 1, g.dim()}, num_neighbours_search, num_neighbours, HNSW_FILL_BA, ids, dist, HNSW_SEM_FUNCTOR);
    std::vector<value_distance> out;
    for (int i = 0; i < num_neighbours && ids[(size_t)i] >= g.id_base(); ++i) out.push_back({ids[(size_t)i], dist[(size_t)i]});
    return out;
}

// Hnsw.Ba.knn_batch hgraph batch ~num_neighbours_search ~num_neighbours -> distances
// (lib/hnsw.ml:769-777): k x nq, +inf filled.
inline std::vector<float> knn_batch(const Hgraph &g, const Mat &batch, int num_neighbours_search, int num_neighbours) {
    std::vector<int32_t> ids; std::vector<float> dist;
    detail::search(g, batch, num_neighbours_search, num_neighbours, HNSW_FILL_BA, ids, dist, HNSW_SEM_FUNCTOR);
    return dist;
}

// Hnsw_algo.Search.search hgraph layer visited ~start_nodes target ~size_nearest (lib/hnsw_algo.ml:350-391)
inline std::vector<value_distance> search(const Hgraph &g, int layer, const std::vector<int64_t> &start_nodes,
                                          const float *target, int size_nearest) {
    return Ohnsw::search_k(g, layer, start_nodes, target, size_nearest, HNSW_SEM_FUNCTOR);
}

} // namespace Ba

// A batch in flight (hnsw_search_submit / hnsw_search_wait): wait() returns what knn_batch_bigarray would have.
class Pending {
public:
    Pending(const Hgraph &g, const Mat &batch, int ef, int k, int fill = HNSW_FILL_OHNSW, int sem = HNSW_SEM_OHNSW) : nq_(batch.dim2), k_(k) {
        hnsw_search_params p{ef, k, fill, sem};
        check(hnsw_search_submit(g.handle(), batch.data, batch.dim2, batch.dim1, &p, &r_));
    }
    Pending(const Pending &) = delete;
    Pending &operator=(const Pending &) = delete;
    Pending(Pending &&o) noexcept : r_(o.r_), nq_(o.nq_), k_(o.k_) { o.r_ = nullptr; }
    std::pair<std::vector<int32_t>, std::vector<float>> wait() {
        std::vector<int32_t> ids((size_t)nq_ * k_); std::vector<float> dist((size_t)nq_ * k_);
        hnsw_request *r = r_; r_ = nullptr;
        check(hnsw_search_wait(r, ids.data(), dist.data(), nullptr, nullptr));
        return {std::move(ids), std::move(dist)};
    }
private:
    hnsw_request *r_ = nullptr;
    int64_t nq_ = 0;
    int k_ = 0;
};

// One host process, several GPUs: the flattened graph replicated on every listed device, batches split
// into contiguous shards (hnsw_multi_*).
class MultiHgraph {
public:
    MultiHgraph(const hnsw_index_desc &desc, const std::vector<int32_t> &devices) : d_(desc.d) {
        check(hnsw_multi_create(&desc, devices.data(), (int32_t)devices.size(), &m_));
    }
    MultiHgraph(const MultiHgraph &) = delete;
    MultiHgraph &operator=(const MultiHgraph &) = delete;
    ~MultiHgraph() { if (m_) hnsw_multi_destroy(m_); }
    int num_replicas() const { int32_t c = 0; check(hnsw_multi_num_replicas(m_, &c)); return c; }

    // Ohnsw.knn_batch_bigarray over all replicas: ids and distances [nq][k]
    std::pair<std::vector<int32_t>, std::vector<float>> knn_batch_bigarray(int k, const Mat &batch, int ef = 0) const {
        std::vector<int32_t> ids((size_t)batch.dim2 * k, -1); std::vector<float> dist((size_t)batch.dim2 * k, 0.f);
        hnsw_search_params p{ef > 0 ? ef : k, k, HNSW_FILL_OHNSW, HNSW_SEM_OHNSW};
        check(hnsw_multi_search_batch(m_, batch.data, batch.dim2, batch.dim1, &p, ids.data(), dist.data(), nullptr, nullptr));
        return {std::move(ids), std::move(dist)};
    }

private:
    hnsw_multi *m_ = nullptr;
    int d_ = 0;
};

} // namespace Hnsw
