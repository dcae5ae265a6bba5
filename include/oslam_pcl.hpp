/*
 * oslam_pcl.hpp -- header-only C++ adaptor that gives the C-ABI of oslam.h the
 * reference's own C++ signatures (pcl/alignment/include/ppf.h:9-15,
 * scene.h:10-52, model.h:14-115).  It is a template over the cloud / point /
 * matrix types so that it compiles with or without PCL and Eigen:
 *
 *   #include <pcl/point_cloud.h>
 *   #include <pcl/point_types.h>
 *   #include <Eigen/Core>
 *   #include "oslam_pcl.hpp"
 *   using Cloud = pcl::PointCloud<pcl::PointNormal>;
 *   std::vector<std::vector<Eigen::Matrix4f>> results =
 *       oslam::ppf_registration<Eigen::Matrix4f>(scene_clouds, model_clouds, model_d_dists,
 *           ref_point_df, vote_count_threshold, cpu_clustering, use_l1_norm,
 *           use_averaged_clusters, devUse, model_weights);
 *
 * Requirements on the types: a cloud pointer type with ->size() and operator[]
 * returning a point with float members x, y, z, normal_x, normal_y, normal_z laid
 * out at a fixed stride (pcl::PointNormal: 48 bytes); a matrix type with
 * float& operator()(int row, int col).  PCL and Eigen are not in this image, so
 * this header is exercised in tests/ with plain structs of the same shape.
 */
#ifndef OSLAM_PCL_HPP
#define OSLAM_PCL_HPP

#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "oslam.h"

namespace oslam {

inline void check(int rc)
{
    if (rc != OSLAM_OK && rc != OSLAM_E_NO_VOTES)
        throw std::runtime_error(std::string("oslam: ") + oslam_last_error());
}

/* Scene(cloud*, d_dist, ref_point_downsample_factor = 1)  -- scene.h:15-16 */
template <class CloudT>
class Scene {
  public:
    Scene(CloudT *cloud, float d_dist, unsigned int ref_point_downsample_factor = 1,
          const oslam_params *params = nullptr)
        : cloud_ptr(cloud), h_(nullptr)
    {
        const auto &p0 = (*cloud)[0];
        const std::size_t stride = cloud->size() > 1 ? (std::size_t)((const char *)&(*cloud)[1] - (const char *)&p0) : sizeof(p0);
        check(oslam_scene_create(&p0.x, &p0.normal_x, cloud->size(), stride, d_dist,
                                 ref_point_downsample_factor, params, &h_));
    }
    ~Scene() { oslam_scene_destroy(h_); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    int numPoints() const { return (int)cloud_ptr->size(); }
    oslam_scene *handle() const { return h_; }
    CloudT *cloud_ptr;

  private:
    oslam_scene *h_;
};

/* Model(cloud*, d_dist, vote_count_threshold, cpu_clustering, use_l1_norm,
 *       use_averaged_clusters)  -- model.h:17-19; ppf_lookup(Scene*) -- model.h:35 */
template <class CloudT>
class Model {
  public:
    Model(CloudT *cloud, float d_dist, float vote_count_threshold, bool cpu_clustering, bool use_l1_norm,
          bool use_averaged_clusters, int dev = 0)
        : cloud_ptr(cloud), h_(nullptr), cpu_clustering_(cpu_clustering)
    {
        oslam_params p;
        oslam_params_default(&p);
        p.vote_count_threshold = vote_count_threshold;
        p.cpu_clustering = cpu_clustering;
        p.use_l1_norm = use_l1_norm;
        p.use_averaged_clusters = use_averaged_clusters;
        p.dev = dev;
        const auto &p0 = (*cloud)[0];
        const std::size_t stride = cloud->size() > 1 ? (std::size_t)((const char *)&(*cloud)[1] - (const char *)&p0) : sizeof(p0);
        check(oslam_model_create(&p0.x, &p0.normal_x, cloud->size(), stride, d_dist, &p, &h_));
    }
    ~Model() { oslam_model_destroy(h_); }
    Model(const Model &) = delete;
    Model &operator=(const Model &) = delete;

    void SetModelPointVoteWeights(const std::vector<float> &w)
    {
        check(oslam_model_set_point_weights(h_, w.data(), w.size()));
    }
    /* Model::ppf_lookup (model.cu:269-306).  Afterwards the result fields below hold what the reference's
     * hold (model.h:92-113), for the cells above the threshold in (count descending, code ascending) order;
     * best_T (row-major) is what ppf.cu:74-93 extracts from them; counters in stats. */
    template <class SceneT>
    void ppf_lookup(SceneT *scene)
    {
        check(oslam_align(h_, scene->handle(), best_T, &stats));
        std::size_t n = 0;
        check(oslam_last_cells(h_, nullptr, nullptr, 0, &n));
        std::vector<oslam_cell> cells(n ? n : 1);
        transformations.assign(16 * n, 0.0f);
        transformation_trans.assign(n, float3_t{0, 0, 0});
        transformation_rots.assign(n, float4_t{0, 0, 0, 0});
        vote_counts_out.assign(n, 0.0f);
        voteCodes.resize(n);
        voteCounts.resize(n);
        cpu_transformations.clear();
        max_idx = 0;
        if (n) {
            check(oslam_last_cells(h_, cells.data(), transformations.data(), n, &n));
            check(oslam_last_result(h_, scene->handle(), &transformation_trans[0].x, &transformation_rots[0].x,
                                    vote_counts_out.data(), n, &n, &max_idx));
            for (std::size_t i = 0; i < n; i++) {
                voteCodes[i] = cells[i].code;
                voteCounts[i] = cells[i].count;
            }
        }
        if (cpu_clustering_) {                       /* ClusterTransformationsCPU: the winning cluster first (ppf.cu:75-77) */
            PoseWithVotes pv;
            for (int k = 0; k < 16; k++) pv.pose[k] = best_T[k];
            pv.votes = n ? (unsigned int)vote_counts_out[0] : 0u;
            cpu_transformations.push_back(pv);
        }
    }
    const std::vector<float> &getTransformations() const { return transformations; }
    oslam_model *handle() const { return h_; }
    CloudT *cloud_ptr;

    /* ---- the reference's public result fields (model.h:92-113), on the host ---- */
    struct float3_t { float x, y, z; };
    struct float4_t { float x, y, z, w; };                 /* a quaternion as (w, x, y, z) in (.x, .y, .z, .w): kernel.cu:124-144 */
    struct PoseWithVotes { float pose[16]; unsigned int votes; };   /* transformation_clustering.h */
    std::vector<unsigned long long> voteCodes;             /* [scene ref | model point << 6 | angle], kernel.cu:549 */
    std::vector<unsigned int> voteCounts;
    std::vector<float> transformations;                    /* 4x4 row-major per cell, linear indexing as the reference's */
    std::vector<float3_t> transformation_trans;            /* after the clustering stage (kernel.cu:758) */
    std::vector<float4_t> transformation_rots;
    std::vector<float> vote_counts_out;                    /* clustered scores (kernel.cu:702-763) */
    std::vector<PoseWithVotes> cpu_transformations;        /* with cpu_clustering: [0] = the winning cluster's pose */
    unsigned int max_idx = 0;
    float best_T[16] = {0};
    oslam_stats stats{};

  private:
    oslam_model *h_;
    bool cpu_clustering_;
};

/* ppf_registration -- ppf.h:9-15.  CloudPtr is e.g. pcl::PointCloud<pcl::PointNormal>::Ptr;
 * Matrix4 e.g. Eigen::Matrix4f.  Models are built once and stay resident; model_weights is
 * accepted and ignored, as in the reference (ppf.cu:35). */
template <class Matrix4, class CloudPtr>
std::vector<std::vector<Matrix4>> ppf_registration(std::vector<CloudPtr> scene_clouds,
                                                   std::vector<CloudPtr> model_clouds,
                                                   std::vector<float> model_d_dists,
                                                   unsigned int ref_point_downsample_factor,
                                                   float vote_count_threshold, bool cpu_clustering,
                                                   bool use_l1_norm, bool use_averaged_clusters, int devUse,
                                                   float *model_weights)
{
    (void)model_weights;
    using CloudT = typename std::remove_reference<decltype(*scene_clouds[0])>::type;
    std::vector<std::vector<Matrix4>> results;
    std::vector<std::unique_ptr<Model<CloudT>>> models;      /* released also when a later step throws */
    for (std::size_t j = 0; j < model_clouds.size(); j++)
        models.emplace_back(new Model<CloudT>(&*model_clouds[j], model_d_dists[j], vote_count_threshold,
                                              cpu_clustering, use_l1_norm, use_averaged_clusters, devUse));
    /* The scenes x models loop of ppf.cu:57-100 as a resident database (oslam_db): models that share a d_dist
     * share the scene pass of every frame, and ONE scene object (d_dist 0: its pair keys are made per group of
     * models inside the align, where the reference prepares the scene per model, ppf.cu:64-67) serves them all.
     * oslam_db_align returns what ppf.cu:74-93 extracts (the pose of max_idx with the clustered translation, or
     * the winning cluster's pose with cpu_clustering); the per-model result fields stay available through
     * Model::ppf_lookup for callers that want them. */
    struct DbGuard {
        oslam_db *h = nullptr;
        ~DbGuard() { oslam_db_destroy_with_models(h); }      /* the models are destroyed right after */
    } db;
    if (!models.empty()) {
        std::vector<oslam_model *> handles;
        for (auto &m : models) handles.push_back(m->handle());
        check(oslam_db_create(handles.data(), handles.size(), &db.h));
    }
    oslam_params sp;
    oslam_params_default(&sp);
    sp.dev = devUse;
    std::vector<float> T(16 * (models.empty() ? 1 : models.size()));
    for (std::size_t i = 0; i < scene_clouds.size(); i++) {
        results.push_back(std::vector<Matrix4>());
        if (models.empty()) continue;
        Scene<CloudT> scene(&*scene_clouds[i], 0.0f, ref_point_downsample_factor, &sp);
        check(oslam_db_align(db.h, scene.handle(), T.data(), nullptr));
        for (std::size_t j = 0; j < models.size(); j++) {
            Matrix4 M;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) M(r, c) = T[16 * j + 4 * r + c];
            results.back().push_back(M);
        }
    }
    return results;
}

} /* namespace oslam */
#endif
