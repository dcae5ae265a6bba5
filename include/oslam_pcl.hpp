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
#include <stdexcept>
#include <string>
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
        : cloud_ptr(cloud), h_(nullptr)
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
    /* best pose lands in best_T (row-major), counters in stats */
    template <class SceneT>
    void ppf_lookup(SceneT *scene)
    {
        check(oslam_align(h_, scene->handle(), best_T, &stats));
    }
    oslam_model *handle() const { return h_; }
    CloudT *cloud_ptr;
    float best_T[16] = {0};
    oslam_stats stats{};

  private:
    oslam_model *h_;
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
    std::vector<Model<CloudT> *> models;
    for (std::size_t j = 0; j < model_clouds.size(); j++)
        models.push_back(new Model<CloudT>(&*model_clouds[j], model_d_dists[j], vote_count_threshold,
                                           cpu_clustering, use_l1_norm, use_averaged_clusters, devUse));
    oslam_params sp;
    oslam_params_default(&sp);
    sp.dev = devUse;
    for (std::size_t i = 0; i < scene_clouds.size(); i++) {
        results.push_back(std::vector<Matrix4>());
        for (std::size_t j = 0; j < model_clouds.size(); j++) {
            /* the scene is prepared per model: its keys depend on the model's d_dist (ppf.cu:64-67) */
            Scene<CloudT> scene(&*scene_clouds[i], model_d_dists[j], ref_point_downsample_factor, &sp);
            models[j]->ppf_lookup(&scene);
            Matrix4 T;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) T(r, c) = models[j]->best_T[4 * r + c];
            results.back().push_back(T);
        }
    }
    for (auto *m : models) delete m;
    return results;
}

} /* namespace oslam */
#endif
