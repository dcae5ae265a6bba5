// Compile-and-link check of include/oslam_pcl.hpp with plain structs shaped like
// pcl::PointNormal / pcl::PointCloud / Eigen::Matrix4f (none of which are in this image).
// With a GPU it also runs one registration and prints the pose.
#include <cstdio>
#include <memory>
#include <type_traits>
#include <vector>

#include "oslam_pcl.hpp"

struct alignas(16) PointNormal {          // 48 bytes, like pcl::PointNormal
    float x, y, z, pad0;
    float normal_x, normal_y, normal_z, pad1;
    float curvature, pad2[3];
};
struct Cloud {
    std::vector<PointNormal> pts;
    std::size_t size() const { return pts.size(); }
    const PointNormal &operator[](std::size_t i) const { return pts[i]; }
    PointNormal &operator[](std::size_t i) { return pts[i]; }
};
struct Mat4 {
    float m[4][4];
    float &operator()(int r, int c) { return m[r][c]; }
};

int main(int argc, char **argv)
{
    static_assert(sizeof(PointNormal) == 48, "layout");
    auto model = std::make_shared<Cloud>(), scene = std::make_shared<Cloud>();
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (int i = 0; i < 80; i++) {
        PointNormal p{};
        p.x = rnd(); p.y = rnd(); p.z = 0.3f * rnd();
        p.normal_x = 0.1f * rnd(); p.normal_y = 0.1f * rnd(); p.normal_z = 1.0f;
        model->pts.push_back(p);
        p.x += 2.0f; p.y -= 1.0f;           // the scene is the model, translated
        scene->pts.push_back(p);
    }
    if (argc > 1) {                         // "run": needs a GPU
        auto res = oslam::ppf_registration<Mat4>(std::vector<std::shared_ptr<Cloud>>{scene},
                                                 std::vector<std::shared_ptr<Cloud>>{model},
                                                 std::vector<float>{0.05f}, 1, 0.4f, false, false, false, 0, nullptr);
        std::printf("t = %.3f %.3f %.3f\n", res[0][0](0, 3), res[0][0](1, 3), res[0][0](2, 3));
        // the reference's result fields (model.h:92-113) after a lookup, read the way ppf.cu:74-93 reads them:
        // the pose they give is the pose oslam_align returned, with and without cpu_clustering
        for (int cpu = 0; cpu < 2; cpu++) {
            oslam::Scene<Cloud> sc(scene.get(), 0.05f, 1);
            oslam::Model<Cloud> mo(model.get(), 0.05f, 0.4f, cpu != 0, false, false);
            mo.ppf_lookup(&sc);
            const std::size_t n = mo.voteCodes.size();
            if (n < 2 || mo.transformations.size() != 16 * n || mo.transformation_trans.size() != n ||
                mo.transformation_rots.size() != n || mo.vote_counts_out.size() != n || mo.max_idx >= n)
                return 3;
            float T[16];
            if (cpu) {
                if (mo.cpu_transformations.size() != 1 || mo.cpu_transformations[0].votes == 0) return 4;
                for (int k = 0; k < 16; k++) T[k] = mo.cpu_transformations[0].pose[k];
            } else {
                for (int k = 0; k < 16; k++) T[k] = mo.getTransformations()[mo.max_idx * 16 + k];
                T[3] = mo.transformation_trans[mo.max_idx].x;
                T[7] = mo.transformation_trans[mo.max_idx].y;
                T[11] = mo.transformation_trans[mo.max_idx].z;
                for (std::size_t i = 0; i < n; i++)
                    if (mo.vote_counts_out[i] > mo.vote_counts_out[mo.max_idx]) return 5;   // max_idx is the argmax
            }
            for (int k = 0; k < 16; k++)
                if (T[k] != mo.best_T[k]) return 6;
        }
        std::printf("fields ok\n");
    }
    return 0;
}
